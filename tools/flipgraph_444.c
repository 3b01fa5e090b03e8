// tools/flipgraph_444.c -- search for a bilinear scheme of rank < 49 for the 4 x 4 x 4 matrix product over GF(2) by a random walk
// in the flip graph of matrix-multiplication schemes (the method of Kauers & Moosbauer, "Flip graphs for matrix multiplication", 2022;
// that rank 47 exists over GF(2) was first found by AlphaTensor, Fawzi et al., Nature 2022).  Own code; developer tool: it produced the
// coefficient table in m4ri_amd/csrc/scheme47.h, which tests/test_host_logic.py re-verifies against the definition of the product.
//
//   A scheme is a list of rank-one tensors (a, b, c), a over the 16 entries of A, b over those of B, c over those of C (16-bit masks), with
//       sum_r a_r[i,j] b_r[j',k] c_r[i',k'] = [i = i'][j = j'][k = k']        (mod 2).
//   flip:    two tensors that share a factor, (a, b, c) + (a, b', c') = (a, b + b', c) + (a, b', c + c')   (and the same with the roles permuted)
//   reduce:  two tensors that share TWO factors merge: (a, b, c) + (a, b, c') = (a, b, c + c'); a zero factor deletes its tensor.
// The walk starts from Strassen's algorithm applied twice (rank 49), flips at random, reduces whenever it can, restarts from the best
// scheme after a path limit, and prints every scheme that beats the best rank so far.
//
//   gcc -O2 -pthread tools/flipgraph_444.c -o build/flipgraph_444
//   build/flipgraph_444 [threads] [seconds] [target rank] [s = start from Strassen squared] [checkpoint out] [checkpoint in] [path limit] [n > 0: a plus transition when stuck or after n flips without a reduction] [0 = without the general reduction] [margin: ranks above its start a walk may climb by plus transitions] [linked plus 0/1] [quality pools 0/1] [span]
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#define MAXR 64
#ifndef N
#define N 4   // -DN=3: the 3 x 3 x 3 product (rank 27 -> 23 is known to be reachable by flips: a check of the moves themselves)
#endif
typedef struct { uint16_t f[3]; } Tri;
typedef struct { Tri t[MAXR]; int r; } Scheme;

static uint64_t rng_next(uint64_t *s) {  // splitmix64
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// entry (i, j) of a 4 x 4 matrix <-> bit 4 i + j
static int verify(const Scheme *s) {
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) for (int j2 = 0; j2 < N; ++j2) for (int k = 0; k < N; ++k)
    for (int i2 = 0; i2 < N; ++i2) for (int k2 = 0; k2 < N; ++k2) {
      int v = 0;
      for (int r = 0; r < s->r; ++r)
        v ^= ((s->t[r].f[0] >> (N * i + j)) & 1) & ((s->t[r].f[1] >> (N * j2 + k)) & 1) & ((s->t[r].f[2] >> (N * i2 + k2)) & 1);
      if (v != ((i == i2) && (j == j2) && (k == k2))) return 0;
    }
  return 1;
}

static void strassen_squared(Scheme *s) {
  // Strassen over GF(2), entries of a 2 x 2 matrix as bits 2 i + j:  M1 = (A11+A22)(B11+B22) -> C11, C22;  M2 = (A21+A22) B11 -> C21, C22;
  // M3 = A11 (B12+B22) -> C12, C22;  M4 = A22 (B21+B11) -> C11, C21;  M5 = (A11+A12) B22 -> C11, C12;  M6 = (A21+A11)(B11+B12) -> C22;
  // M7 = (A12+A22)(B21+B22) -> C11
  static const uint8_t S[7][3] = {{0x9, 0x9, 0x9}, {0xC, 0x1, 0xC}, {0x1, 0xA, 0xA}, {0x8, 0x5, 0x5}, {0x3, 0x8, 0x3}, {0x5, 0x3, 0x8}, {0xA, 0xC, 0x1}};
  s->r = 0;
  for (int u = 0; u < 7; ++u)
    for (int v = 0; v < 7; ++v) {
      Tri t;
      for (int f = 0; f < 3; ++f) {
        uint16_t m = 0;
        for (int e1 = 0; e1 < 4; ++e1)
          for (int e2 = 0; e2 < 4; ++e2)
            if (((S[u][f] >> e1) & 1) && ((S[v][f] >> e2) & 1)) {
              const int i = 2 * (e1 >> 1) + (e2 >> 1), j = 2 * (e1 & 1) + (e2 & 1);
              m |= (uint16_t)1 << (4 * i + j);
            }
        t.f[f] = m;
      }
      s->t[s->r++] = t;
    }
}

// remove tensors with a zero factor and merge tensors that share two factors, until nothing changes
static void reduce(Scheme *s) {
  for (int again = 1; again;) {
    again = 0;
    for (int i = 0; i < s->r; ++i)
      if (!s->t[i].f[0] || !s->t[i].f[1] || !s->t[i].f[2]) { s->t[i] = s->t[--s->r]; again = 1; --i; }
    for (int i = 0; i < s->r && !again; ++i)
      for (int j = i + 1; j < s->r && !again; ++j)
        for (int f = 0; f < 3; ++f) {
          const int g = (f + 1) % 3, h = (f + 2) % 3;
          if (s->t[i].f[g] == s->t[j].f[g] && s->t[i].f[h] == s->t[j].f[h]) {
            s->t[i].f[f] ^= s->t[j].f[f];
            s->t[j] = s->t[--s->r];
            again = 1;
            break;
          }
        }
  }
}

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static Scheme g_best;
static volatile int g_stop = 0;
static int g_target = 47;
static double g_t0;
static volatile uint64_t g_steps = 0;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void report(const Scheme *s, int thread, uint64_t steps) {
  pthread_mutex_lock(&g_mu);
  if (s->r < g_best.r && verify(s)) {
    g_best = *s;
    printf("# rank %d after %.1f s (thread %d, %llu flips)\n", s->r, now() - g_t0, thread, (unsigned long long)steps);
    for (int r = 0; r < s->r; ++r) printf("{0x%04x, 0x%04x, 0x%04x},\n", s->t[r].f[0], s->t[r].f[1], s->t[r].f[2]);
    fflush(stdout);
    if (s->r <= g_target) g_stop = 1;
  }
  pthread_mutex_unlock(&g_mu);
}

static Scheme g_start;
static uint64_t g_path_limit = 20000000;
static int g_plus = 0, g_general = 1, g_linked = 1;
static volatile uint64_t g_moved = 0, g_end_hist[12];   // walks by (rank at the end - rank at the start + 2)

// The general reduction of the flip-graph paper: if among the tensors that share the factor `val` at position f the factors at another
// position g are linearly DEPENDENT, t_k.g = sum_{i in S} t_i.g, then  val x t_k.g x t_k.h = sum_i val x t_i.g x t_k.h  folds into the others
// (t_i.h += t_k.h) and t_k disappears: one rank less.  (Two tensors equal in two factors are the case |S| = 1.)  Returns 1 if it reduced.
static int reduce_group(Scheme *s, int f, uint16_t val) {
  int idx[MAXR], n = 0;
  for (int i = 0; i < s->r; ++i)
    if (s->t[i].f[f] == val) idx[n++] = i;
  if (n < 2) return 0;
  for (int gg = 1; gg <= 2; ++gg) {
    const int g = (f + gg) % 3, h = 3 - f - g;
    // Gaussian elimination over GF(2) on the vectors t.g, remembering which tensors each reduced vector is a sum of (bit masks over idx)
    uint16_t basis[16];
    uint64_t comb[16];
    int nb = 0;
    for (int k = 0; k < n; ++k) {
      uint16_t v = s->t[idx[k]].f[g];
      uint64_t c = (uint64_t)1 << k;
      for (int b = 0; b < nb; ++b)
        if ((v ^ basis[b]) < v) { v ^= basis[b]; c ^= comb[b]; }
      if (v) {   // a new basis vector: keep the basis reduced from the top (each basis vector has a leading bit the later ones lack)
        basis[nb] = v; comb[nb] = c; ++nb;
        continue;
      }
      // dependent: the tensors in c (k among them) sum to zero at position g.  Drop tensor k: every other tensor of c takes its h factor
      for (int q = 0; q < n; ++q)
        if (q != k && ((c >> q) & 1)) s->t[idx[q]].f[h] ^= s->t[idx[k]].f[h];
      s->t[idx[k]] = s->t[--s->r];
      return 1;
    }
  }
  return 0;
}

// one flip at random + the reductions it makes possible; returns 1 when something was flipped
static int step(Scheme *cur, uint64_t *rng) {
  const int i = (int)(rng_next(rng) % (uint64_t)cur->r), f = (int)(rng_next(rng) % 3);
  int cand[MAXR], nc = 0;
  for (int j = 0; j < cur->r; ++j)
    if (j != i && cur->t[j].f[f] == cur->t[i].f[f]) cand[nc++] = j;
  if (!nc) return 0;
  const int j = cand[rng_next(rng) % (uint64_t)nc];
  // (a, b, c) + (a, b', c') -> (a, b + b', c) + (a, b', c + c'), the two other factors in a random order
  const int g = (rng_next(rng) & 1) ? (f + 1) % 3 : (f + 2) % 3, h = 3 - f - g;
  cur->t[i].f[g] ^= cur->t[j].f[g];
  cur->t[j].f[h] ^= cur->t[i].f[h];
  int hit = !cur->t[i].f[g] || !cur->t[j].f[h];
  for (int k = 0; k < cur->r && !hit; ++k) {   // did the flip create a pair that shares two factors?  (only pairs with i or j can be new)
    if (k != i && (cur->t[k].f[0] == cur->t[i].f[0]) + (cur->t[k].f[1] == cur->t[i].f[1]) + (cur->t[k].f[2] == cur->t[i].f[2]) >= 2) hit = 1;
    if (k != j && (cur->t[k].f[0] == cur->t[j].f[0]) + (cur->t[k].f[1] == cur->t[j].f[1]) + (cur->t[k].f[2] == cur->t[j].f[2]) >= 2) hit = 1;
  }
  static int fullcheck = -1;
  if (fullcheck < 0) fullcheck = getenv("FLIP_FULLCHECK") ? 1 : 0;
  if (fullcheck) {   // diagnostic: every group of the scheme after every flip
    reduce(cur);
    for (int again = 1; again;) {
      again = 0;
      for (int k = 0; k < cur->r && !again; ++k)
        for (int ff = 0; ff < 3 && !again; ++ff)
          if (reduce_group(cur, ff, cur->t[k].f[ff])) { reduce(cur); again = 1; }
    }
    return 1;
  }
  if (hit) reduce(cur);
  else if (g_general) {   // the general reduction on the three groups the flip touched
    // five groups can have become dependent: the one the two tensors share (both changed a vector in it), the two they JOINED with their
    // new factors, and the two they stayed in with a changed vector -- tensor i in the group of its h factor (its g vector changed),
    // tensor j in the group of its g factor (its h vector changed).  (Leaving the last two out -- as this tool did at first -- misses most of the
    // reductions: the 3 x 3 x 3 walk then stops at rank 26 instead of reaching 23 in a second.)
    const uint16_t a = cur->t[i].f[f], bi = cur->t[i].f[g], cj = cur->t[j].f[h], ci = cur->t[i].f[h], bj = cur->t[j].f[g];
    if (reduce_group(cur, f, a) || reduce_group(cur, g, bi) || reduce_group(cur, h, cj) || reduce_group(cur, h, ci) || reduce_group(cur, g, bj)) reduce(cur);
  }
  return 1;
}

// Pools by rank (the search strategy of the flip-graph paper): walks start from a random member of the pool of the working level L --
// the lowest rank whose pool is full -- and run until they lose a rank (the reduced scheme joins the pool of its rank) or reach the path
// limit; a quarter of the walks start from the best rank reached so far, however few schemes it has.  Breadth at every level is what
// gets below the plateaus a single greedy path sticks on.
#ifndef POOL
#define POOL 256
#endif
static int MARGIN = 4;
static Scheme g_pool[65][POOL];
static int g_count[65];

// "Quality" of a scheme: the number of pairs of tensors that share a factor, i.e. the flips it offers.  Most schemes of rank 49 and 50 a
// plain descent reaches offer none or a handful (245 of 256 pooled rank-49 schemes had zero): dead ends.  With g_quality the pools prefer
// schemes that offer many (tournament replacement), walks start from the better of two random members of one of the SPAN + 1 lowest
// ranks, and a walk that ends at the rank it began at leaves the richest scheme it passed through in the pool.
static int g_quality = 0, SPAN = 3;
static int g_q[65][POOL];
static int quality(const Scheme *s) {
  int q = 0;
  for (int i = 1; i < s->r; ++i)
    for (int j = 0; j < i; ++j)
      q += (s->t[i].f[0] == s->t[j].f[0]) + (s->t[i].f[1] == s->t[j].f[1]) + (s->t[i].f[2] == s->t[j].f[2]);
  return q;
}
static void pool_insert(const Scheme *s, uint64_t *rng) {   // g_mu held
  const int r = s->r, c = g_count[r] < POOL ? g_count[r] : POOL, q = quality(s);
  if (c < POOL) { g_pool[r][c] = *s; g_q[r][c] = q; g_count[r] = c + 1; return; }
  int a = (int)(rng_next(rng) % POOL), b = (int)(rng_next(rng) % POOL);
  if (g_q[r][b] < g_q[r][a]) a = b;
  if (!g_quality || q >= g_q[r][a] || (rng_next(rng) & 15) == 0) { g_pool[r][a] = *s; g_q[r][a] = q; }
  g_count[r] = g_count[r] + 1;
}

static int working_level(void) {   // the lowest rank whose pool is full; before any is (the start, a resumed checkpoint): the highest rank that has schemes
  int lvl = 0;
  for (int r = 64; r >= 1; --r)
    if (g_count[r] >= POOL) lvl = r;
  for (int r = 64; r >= 1 && !lvl; --r)
    if (g_count[r]) lvl = r;
  return lvl;
}

static void *walk(void *arg) {
  const int id = (int)(intptr_t)arg;
  uint64_t rng = 0x1234567ull * (uint64_t)(id + 1) + (uint64_t)time(NULL);
  uint64_t steps = 0;
  const uint64_t path_limit = g_path_limit;
  while (!g_stop) {
    Scheme cur;
    pthread_mutex_lock(&g_mu);
    int lvl = working_level();
    // (report() publishes a new best rank before the walk that found it has put it into its pool: until then that pool may be empty)
    if ((rng_next(&rng) & 3) == 0 && g_count[g_best.r]) lvl = g_best.r;
    int q0 = 0;
    if (g_quality) {
      for (int tries = 0; tries < 16; ++tries) {
        lvl = g_best.r + (int)(rng_next(&rng) % (uint64_t)(SPAN + 1));
        if (lvl > 64 || !g_count[lvl]) { lvl = g_best.r; continue; }
        const uint64_t n = (uint64_t)(g_count[lvl] < POOL ? g_count[lvl] : POOL);
        int a = (int)(rng_next(&rng) % n), b = (int)(rng_next(&rng) % n);
        if (g_q[lvl][b] > g_q[lvl][a]) a = b;
        cur = g_pool[lvl][a];
        q0 = g_q[lvl][a];
        if (q0 || g_plus) break;
      }
    } else
    cur = g_pool[lvl][rng_next(&rng) % (uint64_t)(g_count[lvl] < POOL ? g_count[lvl] : POOL)];
    pthread_mutex_unlock(&g_mu);
    const int start_rank = cur.r;
    Scheme snap = cur;
    int snap_q = q0;
    // "Plus transitions" (Kauers & Moosbauer 2023; Arai, Ichikawa & Hukushima 2024): two tensors become three,
    //   (a, b, c) + (a', b', c') = (a + a', b, c) + (a', b + b', c) + (a', b', c + c'),
    // one rank up and into a part of the graph plain flips do not reach.  Schemes of low rank tend to be DEAD ENDS -- no two tensors share
    // a factor (Strassen applied twice is one: 49 distinct factors in every position) -- so a walk that finds nothing to flip, or has not
    // lost a rank for g_plus flips, takes a plus transition as long as it stays within MARGIN ranks of where it started.
    uint64_t p = 0, since = 0;
    int fails = 0;
    for (; p < path_limit && !g_stop && cur.r >= start_rank; ++p) {
      const int before = cur.r;
      if (step(&cur, &rng)) fails = 0; else ++fails;
      ++since;
      if (!g_plus && fails > 4096) break;   // nothing to flip and no plus transitions: a dead end
      if (g_quality && (p & 2047) == 2047 && cur.r == start_rank) {
        const int q = quality(&cur);
        if (q > snap_q) { snap = cur; snap_q = q; }
      }
      if (cur.r < before) since = 0;
      if (g_plus && (fails > 64 || since > (uint64_t)g_plus) && cur.r < start_rank + MARGIN && cur.r < MAXR - 1 && cur.r >= 2) {
        // the pair and the orientation (f, g, h) of the transition: preferably one whose new factor t_i.f + t_j.f is ALREADY the f-th factor
        // of a third tensor -- the new tensor then has somebody to flip with, and the walk leaves the three tensors of the transition
        // (a plus transition between tensors that have nothing in common with anybody mostly just undoes itself)
        int i = (int)(rng_next(&rng) % (uint64_t)cur.r), j = -1, f = (int)(rng_next(&rng) % 3);
        if (g_linked) {
          const int j0 = (int)(rng_next(&rng) % (uint64_t)cur.r);
          for (int q = 0; q < cur.r && j < 0; ++q) {
            const int jj = (j0 + q) % cur.r;
            if (jj == i) continue;
            const uint16_t v = cur.t[i].f[f] ^ cur.t[jj].f[f];
            for (int k = 0; k < cur.r; ++k)
              if (k != i && k != jj && cur.t[k].f[f] == v) { j = jj; break; }
          }
        }
        if (j < 0) {
          j = (int)(rng_next(&rng) % (uint64_t)(cur.r - 1));
          if (j >= i) ++j;
        }
        const int g = (f + 1 + (int)(rng_next(&rng) & 1)) % 3, h = 3 - f - g;
        const Tri a = cur.t[i], b = cur.t[j];
        if (a.f[0] != b.f[0] && a.f[1] != b.f[1] && a.f[2] != b.f[2]) {
          // (a, b, c) + (a', b', c') = (a + a', b, c) + (a', b + b', c) + (a', b', c + c')   with (f, g, h) in the roles of the three positions
          cur.t[i].f[f] = a.f[f] ^ b.f[f];
          cur.t[j].f[h] = a.f[h] ^ b.f[h];
          Tri n;
          n.f[f] = b.f[f]; n.f[g] = (uint16_t)(a.f[g] ^ b.f[g]); n.f[h] = a.f[h];
          cur.t[cur.r++] = n;
          fails = 0; since = 0;
        }
      }
    }
    steps += p;
    __sync_fetch_and_add(&g_steps, p);
    if (cur.r - start_rank + 2 >= 0 && cur.r - start_rank + 2 < 12) __sync_fetch_and_add(&g_end_hist[cur.r - start_rank + 2], 1);
    if (cur.r < start_rank) {
      if (cur.r < g_best.r) report(&cur, id, steps);
      pthread_mutex_lock(&g_mu);
      pool_insert(&cur, &rng);
      pthread_mutex_unlock(&g_mu);
    } else if (g_quality) {
      if (snap_q > q0 && verify(&snap)) {
        pthread_mutex_lock(&g_mu);
        pool_insert(&snap, &rng);
        pthread_mutex_unlock(&g_mu);
        __sync_fetch_and_add(&g_moved, 1);
      }
    } else if (g_plus && cur.r == start_rank && verify(&cur)) {
      // back at the rank it started from, a whole path later (and, where the walk began with a plus transition, possibly in another
      // component of the graph): it replaces a random member of its pool, so that the pool drifts instead of sitting in one basin
      pthread_mutex_lock(&g_mu);
      if (g_count[cur.r] >= POOL) g_pool[cur.r][rng_next(&rng) % POOL] = cur;
      else g_pool[cur.r][g_count[cur.r]++] = cur;
      pthread_mutex_unlock(&g_mu);
      __sync_fetch_and_add(&g_moved, 1);
    }
  }
  return NULL;
}

static void standard(Scheme *s) {  // the definition: a_ij b_jk -> c_ik, rank 64
  s->r = 0;
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) for (int k = 0; k < N; ++k) {
    Tri t = {{(uint16_t)(1u << (N * i + j)), (uint16_t)(1u << (N * j + k)), (uint16_t)(1u << (N * i + k))}};
    s->t[s->r++] = t;
  }
}

// A checkpoint path that must not be used: absent, "none" / "-" (as the GPU tool takes them), or something that exists and is not a
// regular file -- checkpoint() renames a temporary over its path, which as root would REPLACE a device node such as /dev/null by a file
static int usable_path(const char *path) {
  if (!path || !path[0] || !strcmp(path, "none") || !strcmp(path, "-")) return 0;
  struct stat sb;
  if (stat(path, &sb) == 0 && !S_ISREG(sb.st_mode)) return 0;
  return 1;
}

static void checkpoint(const char *path) {   // the pools of the three lowest ranks reached, one scheme per line (rank, then its tensors)
  if (!usable_path(path)) return;
  char tmp[512];
  snprintf(tmp, sizeof tmp, "%s.tmp", path);
  pthread_mutex_lock(&g_mu);
  FILE *fo = fopen(tmp, "w");
  if (fo) {
    for (int r = g_best.r; r <= g_best.r + 2 && r <= 64; ++r)
      for (int q = 0; q < (g_count[r] < POOL ? g_count[r] : POOL); ++q) {
        fprintf(fo, "%d", r);
        for (int t = 0; t < r; ++t) fprintf(fo, " %x %x %x", g_pool[r][q].t[t].f[0], g_pool[r][q].t[t].f[1], g_pool[r][q].t[t].f[2]);
        fprintf(fo, "\n");
      }
    fclose(fo);
    rename(tmp, path);
  }
  pthread_mutex_unlock(&g_mu);
}

int main(int argc, char **argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 8;
  const double seconds = argc > 2 ? atof(argv[2]) : 600.0;
  g_target = argc > 3 ? atoi(argv[3]) : 47;
  if (argc > 7) g_path_limit = strtoull(argv[7], NULL, 10);
  if (argc > 8) g_plus = atoi(argv[8]);
  if (argc > 9) g_general = atoi(argv[9]);
  if (argc > 10) MARGIN = atoi(argv[10]);
  if (argc > 11) g_linked = atoi(argv[11]);
  if (argc > 12) g_quality = atoi(argv[12]);
  if (argc > 13) SPAN = atoi(argv[13]);
  if (argc > 4 && argv[4][0] == 's') strassen_squared(&g_best); else standard(&g_best);
  memset(g_count, 0, sizeof g_count);
  if (!verify(&g_best)) { fprintf(stderr, "the start scheme does not verify\n"); return 2; }
  g_start = g_best;
  g_pool[g_best.r][0] = g_best;
  g_q[g_best.r][0] = quality(&g_best);
  g_count[g_best.r] = 1;
  if (argc > 6 && usable_path(argv[6])) {   // resume from a checkpoint
    FILE *fi = fopen(argv[6], "r");
    int r;
    while (fi && fscanf(fi, "%d", &r) == 1 && r >= 1 && r <= 64) {
      Scheme x;
      x.r = r;
      for (int t = 0; t < r; ++t) { unsigned a, b, c; if (fscanf(fi, "%x %x %x", &a, &b, &c) != 3) { r = 0; break; } x.t[t].f[0] = (uint16_t)a; x.t[t].f[1] = (uint16_t)b; x.t[t].f[2] = (uint16_t)c; }
      if (!r || !verify(&x)) continue;
      if (g_count[r] < POOL) { g_q[r][g_count[r]] = quality(&x); g_pool[r][g_count[r]++] = x; }
      if (r < g_best.r) g_best = x;
    }
    if (fi) fclose(fi);
    for (int r2 = g_best.r; r2 <= 64; ++r2) if (g_count[r2] && g_count[r2] < POOL && r2 > g_best.r + 2) g_count[r2] = 0;
    printf("# resumed: best rank %d, pool of it %d\n", g_best.r, g_count[g_best.r]);
  }
  printf("# start: %s, rank %d, verified\n", (argc > 4 && argv[4][0] == 's') ? "Strassen squared" : "the standard algorithm", g_best.r);
  fflush(stdout);
  g_t0 = now();
  pthread_t th[256];
  for (int i = 0; i < threads && i < 256; ++i) pthread_create(&th[i], NULL, walk, (void *)(intptr_t)i);
  for (double last = now(); !g_stop && now() - g_t0 < seconds;) {
    struct timespec ts = {0, 200000000};
    nanosleep(&ts, NULL);
    if (now() - last > 120.0) {   // a checkpoint every two minutes, and where the pools stand
      last = now();
      checkpoint(argc > 5 ? argv[5] : NULL);
      printf("# %.0f s: %llu pool members replaced; best %d, pools", now() - g_t0, (unsigned long long)g_moved, g_best.r);
      for (int r = g_best.r + 3; r >= g_best.r; --r) if (r <= 64) {
        long sum = 0; int mx = 0; const int c = g_count[r] < POOL ? g_count[r] : POOL;
        for (int q = 0; q < c; ++q) { sum += g_q[r][q]; if (g_q[r][q] > mx) mx = g_q[r][q]; }
        printf(" %d:%d(q %.1f max %d)", r, g_count[r], c ? (double)sum / c : 0.0, mx);
      }
      printf("; walks ended at start%+d..:", -2);
      for (int k = 0; k < 9; ++k) printf(" %llu", (unsigned long long)g_end_hist[k]);
      printf("\n");
      fflush(stdout);
    }
  }
  g_stop = 1;
  for (int i = 0; i < threads && i < 256; ++i) pthread_join(th[i], NULL);
  checkpoint(argc > 5 ? argv[5] : NULL);
  printf("# best rank %d, %.3g flips in all; pool sizes:", g_best.r, (double)g_steps);
  for (int r = 64; r >= 40; --r) if (g_count[r]) printf(" %d:%d", r, g_count[r]);
  printf("\n");
  return g_best.r <= g_target ? 0 : 1;
}
