// tools/flipgraph_444_gpu.hip -- the flip-graph search of tools/flipgraph_444.c (same moves: flips, the general reduction, linked plus
// transitions, pools by rank) with the WALKS on the MI355X: one wavefront per walk, one tensor of the scheme per lane (rank <= 64 = the
// width of a wave), so that a flip is a handful of ballots and readlanes and needs no memory at all:
//     who shares factor f with tensor i?      one v_cmp + ballot
//     pick the k-th of them                   mbcnt + ballot + ff1
//     did the flip create a reducible pair?   three compares per lane against tensors i and j, one ballot
//     is one of the three touched groups linearly dependent?   members by ballot, their vectors by readlane, elimination in SGPRs
// Only the rare moves (a real reduction, a plus transition) go through LDS, where lane 0 runs the serial code of the CPU tool.
// 8192+ walks are resident at once; the host keeps the pools, restarts walks that outlive the path limit, verifies and prints every scheme
// below the best rank.  Developer tool (own code; Kauers & Moosbauer 2022, Arai, Ichikawa & Hukushima 2024 for the method).
//
//   hipcc -O3 --offload-arch=gfx950 tools/flipgraph_444_gpu.hip -o build/flipgraph_444_gpu
//   build/flipgraph_444_gpu [seconds] [pool file in] [pool file out] [path limit] [plus interval] [margin] [walks] [flips per launch] [x = start from the standard algorithm] [span] [thresholds rank:flips,...] [lazy mask] [target rank]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define MAXR 64
#ifndef NDIM
#define NDIM 4   // -DNDIM=3: the 3 x 3 x 3 product, where rank 27 -> 23 is known to be reachable in seconds: a check of the moves
#endif
struct Scheme { uint16_t t[MAXR][3]; int32_t r; };
struct Walker {
  Scheme s;
  int32_t base;        // the lowest rank this walk has had since it was (re)started
  uint32_t since;      // flips since the last reduction or plus transition
  uint64_t rng;
  uint64_t epoch;      // flips since the restart
  uint64_t slow;       // visits of the serial path (statistics)
  uint64_t failed;     // reductions declined (selective reduction)
  uint32_t done, pad;  // flips of the last launch
};
struct Found { Scheme s; };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(3); } } while (0)

__host__ __device__ static inline uint32_t pcg(uint64_t &s) {
  const uint64_t o = s;
  s = o * 6364136223846793005ull + 1442695040888963407ull;
  const uint32_t x = (uint32_t)(((o >> 18) ^ o) >> 27), rot = (uint32_t)(o >> 59);
  return (x >> rot) | (x << ((32 - rot) & 31));
}

// ---------------------------------------------------------------- the serial moves (lane 0 on a scheme in LDS; the host uses them too)
__host__ __device__ static void reduce_plain(Scheme *s) {   // zero factors delete; two tensors equal in two factors merge
  for (int again = 1; again;) {
    again = 0;
    for (int i = 0; i < s->r; ++i)
      if (!s->t[i][0] || !s->t[i][1] || !s->t[i][2]) {
        --s->r;
        for (int f = 0; f < 3; ++f) s->t[i][f] = s->t[s->r][f];
        again = 1; --i;
      }
    for (int i = 0; i < s->r && !again; ++i)
      for (int j = i + 1; j < s->r && !again; ++j)
        for (int f = 0; f < 3; ++f) {
          const int g = (f + 1) % 3, h = (f + 2) % 3;
          if (s->t[i][g] == s->t[j][g] && s->t[i][h] == s->t[j][h]) {
            s->t[i][f] ^= s->t[j][f];
            --s->r;
            for (int q = 0; q < 3; ++q) s->t[j][q] = s->t[s->r][q];
            again = 1;
            break;
          }
        }
  }
}

// the general reduction on the group of tensors whose factor f is val: a linear dependence among their factors at another position
__host__ __device__ static int reduce_group(Scheme *s, int f, uint16_t val) {
  int idx[MAXR], n = 0;
  for (int i = 0; i < s->r; ++i)
    if (s->t[i][f] == val) idx[n++] = i;
  if (n < 2) return 0;
  for (int gg = 1; gg <= 2; ++gg) {
    const int g = (f + gg) % 3, h = 3 - f - g;
    uint16_t basis[16];
    uint64_t comb[16];
    int nb = 0;
    for (int k = 0; k < n; ++k) {
      uint16_t v = s->t[idx[k]][g];
      uint64_t c = (uint64_t)1 << k;
      for (int b = 0; b < nb; ++b)
        if ((uint16_t)(v ^ basis[b]) < v) { v ^= basis[b]; c ^= comb[b]; }
      if (v) { basis[nb] = v; comb[nb] = c; ++nb; continue; }
      for (int q = 0; q < n; ++q)
        if (q != k && ((c >> q) & 1)) s->t[idx[q]][h] ^= s->t[idx[k]][h];
      --s->r;
      for (int q = 0; q < 3; ++q) s->t[idx[k]][q] = s->t[s->r][q];
      return 1;
    }
  }
  return 0;
}

__host__ __device__ static void reduce_all(Scheme *s) {   // everything the two reductions can do, until nothing changes
  for (int again = 1; again;) {
    again = 0;
    reduce_plain(s);
    for (int i = 0; i < s->r && !again; ++i)
      for (int f = 0; f < 3 && !again; ++f)
        if (reduce_group(s, f, s->t[i][f])) again = 1;
  }
}

__host__ __device__ static int quality_of(const Scheme *s) {   // pairs of tensors that share a factor = the flips the scheme offers
  int q = 0;
  for (int i = 1; i < s->r; ++i)
    for (int j = 0; j < i; ++j) q += (s->t[i][0] == s->t[j][0]) + (s->t[i][1] == s->t[j][1]) + (s->t[i][2] == s->t[j][2]);
  return q;
}

// Selective reduction.  Left to itself the walk reduces whenever it can, and from the standard algorithm that funnel ends, 99 times in 100,
// on the orbit of Strassen applied twice: rank 49, factor ranks (36 x 1, 12 x 2, 1 x 4) in all three positions, no two tensors share a
// factor, nothing to flip (72 157 arrivals at rank 49 in nine minutes of the GPU: 191 with a flip to offer, each with a component of a handful
// of schemes).  On the way down the flips a scheme offers fall by about four per rank (52: 13, 51: 8.5, 50: 4.8, 49: 0).  So a reduction is
// TAKEN only if the reduced scheme still offers at least thr[its rank] flips; otherwise the flip that made it possible is undone and the
// walk goes on at its rank -- it then explores the plateau until it finds a way down that keeps it alive.
struct Thresholds { int8_t t[65]; };

// (a, b, c) + (a', b', c') = (a + a', b, c) + (a', b + b', c) + (a', b', c + c'), preferably where a + a' already is somebody's factor
__host__ __device__ static void plus_transition(Scheme *s, uint64_t &rng) {
  if (s->r >= MAXR - 1 || s->r < 2) return;
  for (int attempt = 0; attempt < 8; ++attempt) {
    const uint32_t rnd = pcg(rng);
    const int r = s->r, i = (int)((rnd & 0xffff) * (uint32_t)r >> 16), f = (int)(((rnd >> 16) & 0xff) * 3 >> 8);
    int j = -1;
    const int j0 = (int)(((rnd >> 24) & 0x7f) * (uint32_t)r >> 7);
    for (int q = 0; q < r && j < 0; ++q) {
      const int jj = (j0 + q) % r;
      if (jj == i) continue;
      const uint16_t v = s->t[i][f] ^ s->t[jj][f];
      for (int k = 0; k < r; ++k)
        if (k != i && k != jj && s->t[k][f] == v) { j = jj; break; }
    }
    if (j < 0) { j = (j0 == i) ? (i + 1) % r : j0; }
    const int g = (f + 1 + (int)(rnd >> 31)) % 3, h = 3 - f - g;
    uint16_t a[3], b[3];
    for (int q = 0; q < 3; ++q) { a[q] = s->t[i][q]; b[q] = s->t[j][q]; }
    if (a[0] == b[0] || a[1] == b[1] || a[2] == b[2]) continue;
    s->t[i][f] = a[f] ^ b[f];
    s->t[j][h] = a[h] ^ b[h];
    s->t[r][f] = b[f]; s->t[r][g] = (uint16_t)(a[g] ^ b[g]); s->t[r][h] = a[h];
    s->r = r + 1;
    return;
  }
}

// ---------------------------------------------------------------- the wave-level walk
__device__ static inline uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ static inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// is the set of up to 6 vectors (the lanes of mask m, read from V) linearly dependent?  More than 6: say yes (the serial path decides).
__device__ static inline bool dependent(uint64_t m, uint32_t V) {
  const int n = __popcll(m);
  if (n < 3) return false;     // two members: equal vectors = a pair that shares two factors, which the pair test sees
  if (n > 6) return true;
  uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
  int nb = 0;
  bool dep = false;
  for (int k = 0; k < n; ++k) {
    const int l = __ffsll((unsigned long long)m) - 1;
    m &= m - 1;
    uint32_t v = rdlane(V, l);
    v = min(v, v ^ b0); v = min(v, v ^ b1); v = min(v, v ^ b2); v = min(v, v ^ b3); v = min(v, v ^ b4);
    if (!v) { dep = true; break; }
    // keep the basis reduced from the top: b0..b4 in order of insertion, each new one already minimal against the earlier ones
    if (nb == 0) b0 = v; else if (nb == 1) b1 = v; else if (nb == 2) b2 = v; else if (nb == 3) b3 = v; else if (nb == 4) b4 = v;
    ++nb;   // (a sixth independent vector is the last one: nothing after it needs the basis)
  }
  return dep;
}

// SA, SB, SC (wave-uniform 64-bit masks): lane l is set in SA when factor A of tensor l is also the factor A of another tensor, i.e.
// (l, position A) has somebody to flip with.  A step draws uniformly among the set bits of the three masks, so that no step is spent on a
// tensor that has no partner (with 5 - 10 such pairs among 50 tensors, 7 draws of 8 would be); the walk is the same Markov chain as that
// of the CPU tool, minus its idle steps.  The masks are maintained incrementally: a flip changes one factor of tensor i and one of tensor j.
__device__ static inline uint64_t shared_mask(uint32_t V, int r, bool valid) {
  uint64_t m = 0;
  for (int l = 0; l < r; ++l) {
    const uint64_t g = __ballot(V == rdlane(V, l) && valid);
    if (__popcll(g) >= 2) m |= g;
  }
  return m;
}

__device__ static inline int nth_bit(uint64_t m, int k, int lane) {
  const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  const uint64_t sel = __ballot(((m >> lane) & 1) && (int)pre == k);
  return uni(__ffsll((unsigned long long)sel) - 1);
}

// one flip with tensor i at the position held in X; i's factor Y and the partner's factor Z change.  res: 0 = nothing to flip (the masks
// were stale: cannot happen), 1 = flipped, 2 = flipped and something is reducible (serial path).
#define FLIP(X, Y, Z, SX, SY, SZ)                                                                                            \
  {                                                                                                                          \
    const uint32_t xi = rdlane(X, i);                                                                                        \
    const uint64_t grp = __ballot(X == xi && valid);                                                                         \
    const uint64_t m = grp & ~(1ull << i);                                                                                   \
    if (!m) { res = 0; }                                                                                                     \
    else {                                                                                                                   \
      const int nc = __popcll(m);                                                                                            \
      const int j = nth_bit(m, (int)(((rnd >> 25) * (uint32_t)nc) >> 7), lane);                                              \
      const uint32_t yi = rdlane(Y, i), yj = rdlane(Y, j), zi = rdlane(Z, i), zj = rdlane(Z, j);                             \
      const uint32_t nyi = yi ^ yj, nzj = zj ^ zi;                                                                           \
      if (lane == i) Y = nyi;                                                                                                \
      if (lane == j) Z = nzj;                                                                                                \
      const int ci = (int)(X == xi) + (int)(Y == nyi) + (int)(Z == zi);                                                      \
      const int cj = (int)(X == xi) + (int)(Y == yj) + (int)(Z == nzj);                                                      \
      const uint64_t pair = __ballot(valid && ((lane != i && ci >= 2) || (lane != j && cj >= 2)));                           \
      const uint64_t oy = __ballot(Y == yi && valid), gy = __ballot(Y == nyi && valid);                                      \
      const uint64_t oz = __ballot(Z == zj && valid), gz = __ballot(Z == nzj && valid);                                      \
      if (__popcll(oy) == 1) SY &= ~oy;                                                                                      \
      if (__popcll(gy) >= 2) SY |= gy; else SY &= ~(1ull << i);                                                              \
      if (__popcll(oz) == 1) SZ &= ~oz;                                                                                      \
      if (__popcll(gz) >= 2) SZ |= gz; else SZ &= ~(1ull << j);                                                              \
      bool red = !nyi || !nzj || pair != 0;                                                                                  \
      if (!red) red = dependent(grp, Y) || dependent(grp, Z) || dependent(gy, X) || dependent(gy, Z) || dependent(gz, X) || dependent(gz, Y); \
      /* ... and the two groups the tensors STAYED in with a changed vector: i in the group of its Z (its Y changed), j in that of its Y */ \
      if (!red) { const uint64_t hz = __ballot(Z == zi && valid); red = dependent(hz, Y) || dependent(hz, X); }               \
      if (!red) { const uint64_t hy = __ballot(Y == yj && valid); red = dependent(hy, Z) || dependent(hy, X); }               \
      res = red ? 2 : 1;                                                                                                     \
    }                                                                                                                        \
  }

__global__ __launch_bounds__(256) void walk_kernel(Walker *walkers, int nwalk, uint32_t flips, uint32_t plus_interval, int margin, Found *found,
                                                   int *nfound, int maxfound, int *best, Thresholds thr, unsigned long long *qhist, uint32_t lazy_mask) {
  __shared__ Scheme sh[4], sh2[4];
  __shared__ int verdict[4];
  const int lane = (int)(threadIdx.x & 63), wv = (int)(threadIdx.x >> 6);
  const int w = (int)blockIdx.x * 4 + wv;
  if (w >= nwalk) return;
  Walker *W = walkers + w;
  Scheme *S = &sh[wv];
  uint32_t A = W->s.t[lane][0], B = W->s.t[lane][1], C = W->s.t[lane][2];
  int r = uni(W->s.r), base = uni(W->base);
  uint32_t since = (uint32_t)uni((int)W->since);
  uint64_t rng = W->rng;
  rng = ((uint64_t)(uint32_t)uni((int)(rng >> 32)) << 32) | (uint32_t)uni((int)rng);
  uint64_t slow = 0, failed = 0;
  bool dead = false;
  uint64_t SA = shared_mask(A, r, lane < r), SB = shared_mask(B, r, lane < r), SC = shared_mask(C, r, lane < r);
  uint32_t p = 0;
  for (; p < flips; ++p) {
    const uint32_t rnd = (uint32_t)uni((int)pcg(rng));
    const bool valid = lane < r;
    const int nA = __popcll(SA), nB = __popcll(SB), nC = __popcll(SC), total = nA + nB + nC;
    int res = 0;
    const uint32_t pA = A, pB = B, pC = C;
    const uint64_t pSA = SA, pSB = SB, pSC = SC;
    if (total) {
      int t = (int)(((rnd & 0xffff) * (uint32_t)total) >> 16);
      const int pos = t < nA ? 0 : (t < nA + nB ? 1 : 2);
      t -= pos == 0 ? 0 : (pos == 1 ? nA : nA + nB);
      const int i = nth_bit(pos == 0 ? SA : (pos == 1 ? SB : SC), t, lane);
      switch (pos * 2 + (int)((rnd >> 24) & 1)) {
        case 0: FLIP(A, B, C, SA, SB, SC) break;
        case 1: FLIP(A, C, B, SA, SC, SB) break;
        case 2: FLIP(B, A, C, SB, SA, SC) break;
        case 3: FLIP(B, C, A, SB, SC, SA) break;
        case 4: FLIP(C, A, B, SC, SA, SB) break;
        default: FLIP(C, B, A, SC, SB, SA) break;
      }
      ++since;
      if (res == 0) ++failed;
    }
    // lazy reduction: of the reducible states the walk comes upon only one in lazy_mask + 1 is entered (the others: the flip is undone), so
    // that the walk mixes on its plateau before it steps down instead of taking the first way down it sees
    if (res == 2 && lazy_mask && ((rnd >> 8) * 2654435761u >> 12 & lazy_mask) != 0) {
      A = pA; B = pB; C = pC; SA = pSA; SB = pSB; SC = pSC;
      continue;
    }
    const bool can_plus = plus_interval && r < base + margin && r < MAXR - 1;
    const bool want_plus = can_plus && (!total || since > plus_interval);
    if (!total && !want_plus) { dead = true; break; }   // a dead end and no way up: the host restarts this walk
    if (res == 2 || want_plus || res == 0) {
      ++slow;
      if (valid) { S->t[lane][0] = (uint16_t)A; S->t[lane][1] = (uint16_t)B; S->t[lane][2] = (uint16_t)C; }
      if (lane == 0) S->r = r;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        verdict[wv] = 1;
        if (res == 2) {
          Scheme *S2 = &sh2[wv];
          *S2 = *S;
          reduce_all(S2);
          if (S2->r < r) {
            const int q = quality_of(S2);
            atomicAdd(&qhist[S2->r * 32 + (q < 31 ? q : 31)], 1ull);
            if (q >= thr.t[S2->r]) *S = *S2; else verdict[wv] = 0;
          }
        } else if (want_plus) { uint64_t g = rng; plus_transition(S, g); }
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (!uni(verdict[wv])) {   // a way down that would leave the scheme (nearly) dead: undo the flip and walk on
        A = pA; B = pB; C = pC; SA = pSA; SB = pSB; SC = pSC;
        ++failed;
        if (failed >= 48) { ++p; break; }   // the serial path is slow: a walk that keeps coming upon declined reductions rests until the next launch
        continue;
      }
      if (res != 2) { pcg(rng); pcg(rng); pcg(rng); pcg(rng); pcg(rng); pcg(rng); pcg(rng); pcg(rng); }   // the draws the transition used
      const int nr = uni(S->r);
      A = S->t[lane][0]; B = S->t[lane][1]; C = S->t[lane][2];
      if (nr != r) since = 0;
      if (nr < base) {   // a rank this walk has not had: into the list the host reads
        base = nr;
        if (lane == 0) {
          atomicMin(best, nr);
          const int slot = atomicAdd(nfound, 1);
          if (slot < maxfound) found[slot].s = *S;
        }
      }
      r = nr;
      SA = shared_mask(A, r, lane < r); SB = shared_mask(B, r, lane < r); SC = shared_mask(C, r, lane < r);
    }
  }
  if (lane < MAXR) { W->s.t[lane][0] = (uint16_t)A; W->s.t[lane][1] = (uint16_t)B; W->s.t[lane][2] = (uint16_t)C; }
  if (lane == 0) {
    W->s.r = r; W->base = base; W->since = since; W->rng = rng; W->epoch = dead ? ~0ull : W->epoch + p; W->slow += slow; W->failed += failed;
    W->done = p;
  }
}

// ---------------------------------------------------------------- host
static int verify(const Scheme *s) {
  const int N = NDIM;
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) for (int j2 = 0; j2 < N; ++j2) for (int k = 0; k < N; ++k)
    for (int i2 = 0; i2 < N; ++i2) for (int k2 = 0; k2 < N; ++k2) {
      int v = 0;
      for (int r = 0; r < s->r; ++r)
        v ^= ((s->t[r][0] >> (N * i + j)) & 1) & ((s->t[r][1] >> (N * j2 + k)) & 1) & ((s->t[r][2] >> (N * i2 + k2)) & 1);
      if (v != ((i == i2) && (j == j2) && (k == k2))) return 0;
    }
  return 1;
}

static int quality(const Scheme &s) {   // pairs of tensors that share a factor = the flips the scheme offers
  int q = 0;
  for (int i = 1; i < s.r; ++i)
    for (int j = 0; j < i; ++j) q += (s.t[i][0] == s.t[j][0]) + (s.t[i][1] == s.t[j][1]) + (s.t[i][2] == s.t[j][2]);
  return q;
}

static int rank4(uint16_t m) {   // rank over GF(2) of the NDIM x NDIM matrix whose entry (i, j) is bit NDIM i + j
  uint16_t rows[NDIM];
  for (int i = 0; i < NDIM; ++i) rows[i] = (uint16_t)((m >> (NDIM * i)) & ((1u << NDIM) - 1));
  int r = 0;
  for (int b = 0; b < NDIM; ++b) {
    int p = -1;
    for (int i = r; i < NDIM; ++i) if ((rows[i] >> b) & 1) { p = i; break; }
    if (p < 0) continue;
    std::swap(rows[r], rows[p]);
    for (int i = 0; i < NDIM; ++i) if (i != r && ((rows[i] >> b) & 1)) rows[i] ^= rows[r];
    ++r;
  }
  return r;
}
// Strassen applied twice has, in each of the three positions, 36 factors of rank 1, 12 of rank 2 and one of rank 4
static bool strassen_squared_signature(const Scheme &s) {
  if (s.r != 49) return false;
  for (int f = 0; f < 3; ++f) {
    int c[5] = {0, 0, 0, 0, 0};
    for (int t = 0; t < s.r; ++t) c[rank4(s.t[t][f])]++;
    if (c[1] != 36 || c[2] != 12 || c[4] != 1) return false;
  }
  return true;
}
static uint64_t g_sig_strassen = 0, g_sig_other = 0;

static const size_t POOLCAP = 4096;
static std::vector<Scheme> g_pool[65];
static std::mt19937_64 g_rng(12345);

static std::vector<Scheme> g_live[65];   // those of the pool that offer at least one flip
static uint64_t g_arrived[65], g_arrived_live[65];
static int quality(const Scheme &s);
static void pool_add(const Scheme &s) {
  auto &p = g_pool[s.r];
  if (p.size() < POOLCAP) p.push_back(s); else p[g_rng() % POOLCAP] = s;
  ++g_arrived[s.r];
  if (quality(s) > 0) {
    ++g_arrived_live[s.r];
    auto &l = g_live[s.r];
    if (l.size() < POOLCAP) l.push_back(s); else l[g_rng() % POOLCAP] = s;
  }
}

static void write_pools(const char *path, int best) {
  if (!path) return;
  std::string tmp = std::string(path) + ".tmp";
  FILE *fo = fopen(tmp.c_str(), "w");
  if (!fo) return;
  for (int r = best; r <= best + 2 && r <= 64; ++r)
    for (size_t q = 0; q < g_pool[r].size() && q < 1024; ++q) {
      fprintf(fo, "%d", r);
      for (int t = 0; t < r; ++t) fprintf(fo, " %x %x %x", g_pool[r][q].t[t][0], g_pool[r][q].t[t][1], g_pool[r][q].t[t][2]);
      fprintf(fo, "\n");
    }
  fclose(fo);
  rename(tmp.c_str(), path);
  fo = fopen((std::string(path) + ".live").c_str(), "w");   // the schemes of the two lowest ranks that offer a flip
  if (!fo) return;
  for (int r = best; r <= best + 1 && r <= 64; ++r)
    for (auto &s : g_live[r]) {
      fprintf(fo, "%d", r);
      for (int t = 0; t < r; ++t) fprintf(fo, " %x %x %x", s.t[t][0], s.t[t][1], s.t[t][2]);
      fprintf(fo, "\n");
    }
  fclose(fo);
}

int main(int argc, char **argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 60.0;
  const char *pool_in = argc > 2 ? argv[2] : nullptr, *pool_out = argc > 3 ? argv[3] : nullptr;
  const uint64_t path_limit = argc > 4 ? strtoull(argv[4], nullptr, 10) : 300000000ull;
  const uint32_t plus_interval = argc > 5 ? (uint32_t)strtoul(argv[5], nullptr, 10) : 1000000u;
  const int margin = argc > 6 ? atoi(argv[6]) : 2;
  const int nwalk = argc > 7 ? atoi(argv[7]) : 16384;
  const uint32_t flips = argc > 8 ? (uint32_t)strtoul(argv[8], nullptr, 10) : 1000000u;
  const bool from_standard = argc > 9 && (argv[9][0] == 'x' || argv[9][0] == 'k');
  const int span = argc > 10 ? atoi(argv[10]) : 3;   // walks start from the pools of the span + 1 lowest ranks
  Thresholds thr;
  memset(&thr, 0, sizeof thr);
  // "51:10,50:8,49:6": a reduction to rank 51 is taken only if the result offers >= 10 flips, ...
  if (argc > 11)
    for (const char *q = argv[11]; q && *q;) {
      int r = 0, t = 0;
      if (sscanf(q, "%d:%d", &r, &t) == 2 && r >= 1 && r <= 64) thr.t[r] = (int8_t)t;
      q = strchr(q, ',');
      if (q) ++q;
    }
  int best = 64;
  {
    Scheme s;
    memset(&s, 0, sizeof s);
    s.r = 0;
    for (int i = 0; i < NDIM; ++i) for (int j = 0; j < NDIM; ++j) for (int k = 0; k < NDIM; ++k) {
      s.t[s.r][0] = (uint16_t)(1u << (NDIM * i + j)); s.t[s.r][1] = (uint16_t)(1u << (NDIM * j + k)); s.t[s.r][2] = (uint16_t)(1u << (NDIM * i + k));
      ++s.r;
    }
    if (!verify(&s)) { fprintf(stderr, "the standard algorithm does not verify\n"); return 2; }
    if (from_standard || !pool_in) pool_add(s);
    // 'k': the Kronecker products of Strassen's algorithm (rank 7) and the definition of the 2 x 2 product (rank 8), in both orders:
    // rank 56, half way between the standard algorithm and Strassen applied twice
    if (argc > 9 && argv[9][0] == 'k') {
      static const uint8_t S7[7][3] = {{0x9, 0x9, 0x9}, {0xC, 0x1, 0xC}, {0x1, 0xA, 0xA}, {0x8, 0x5, 0x5}, {0x3, 0x8, 0x3}, {0x5, 0x3, 0x8}, {0xA, 0xC, 0x1}};
      uint8_t S8[8][3];
      int n8 = 0;
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
        S8[n8][0] = (uint8_t)(1u << (2 * i + j)); S8[n8][1] = (uint8_t)(1u << (2 * j + k)); S8[n8][2] = (uint8_t)(1u << (2 * i + k)); ++n8;
      }
      for (int order = 0; order < 2; ++order) {
        Scheme x;
        memset(&x, 0, sizeof x);
        for (int u = 0; u < 7; ++u) for (int v = 0; v < 8; ++v) {
          for (int f = 0; f < 3; ++f) {
            const uint8_t outer = order ? S8[v][f] : S7[u][f], inner = order ? S7[u][f] : S8[v][f];
            uint16_t m = 0;
            for (int e1 = 0; e1 < 4; ++e1) for (int e2 = 0; e2 < 4; ++e2)
              if (((outer >> e1) & 1) && ((inner >> e2) & 1)) m |= (uint16_t)(1u << (4 * (2 * (e1 >> 1) + (e2 >> 1)) + 2 * (e1 & 1) + (e2 & 1)));
            x.t[x.r][f] = m;
          }
          ++x.r;
        }
        if (!verify(&x)) { fprintf(stderr, "Strassen (x) standard does not verify\n"); return 2; }
        pool_add(x);
      }
    }
  }
  if (pool_in && !from_standard) {
    FILE *fi = fopen(pool_in, "r");
    int r;
    while (fi && fscanf(fi, "%d", &r) == 1 && r >= 1 && r <= 64) {
      Scheme x;
      memset(&x, 0, sizeof x);
      x.r = r;
      for (int t = 0; t < r; ++t) { unsigned a, b, c; if (fscanf(fi, "%x %x %x", &a, &b, &c) != 3) { r = 0; break; } x.t[t][0] = (uint16_t)a; x.t[t][1] = (uint16_t)b; x.t[t][2] = (uint16_t)c; }
      if (r && verify(&x)) pool_add(x);
    }
    if (fi) fclose(fi);
  }
  for (int r = 64; r >= 1; --r) if (!g_pool[r].empty()) best = r;
  printf("# start: best rank %d, pools", best);
  for (int r = best; r <= 64; ++r) if (!g_pool[r].empty()) printf(" %d:%zu", r, g_pool[r].size());
  printf("; %d walks, %u flips per launch, path limit %llu, plus every %u, margin %d, span %d\n", nwalk, flips, (unsigned long long)path_limit, plus_interval, margin, span);
  fflush(stdout);

  // a walk starts from a random member of the pool of one of the span + 1 lowest ranks; without plus transitions only from members that
  // offer a flip (most schemes of rank 49 offer none), and half of the walks from the lowest rank that has such members
  auto restart = [&](Walker &w) {
    auto *pools = plus_interval ? g_pool : g_live;
    int lvl = -1;
    if (g_rng() & 1)
      for (int r = best; r <= best + span && r <= 64 && lvl < 0; ++r) if (!pools[r].empty()) lvl = r;
    for (int tries = 0; tries < 64 && lvl < 0; ++tries) { const int r = best + (int)(g_rng() % (uint64_t)(span + 1)); if (r <= 64 && !pools[r].empty()) lvl = r; }
    for (int r = best; r <= 64 && lvl < 0; ++r) if (!pools[r].empty()) lvl = r;
    if (lvl < 0) { fprintf(stderr, "no scheme to start a walk from\n"); exit(5); }
    const uint64_t keep_slow = w.slow, keep_failed = w.failed;
    memset(&w.s, 0, sizeof w.s);
    w.s = pools[lvl][g_rng() % pools[lvl].size()];
    w.base = w.s.r; w.since = 0; w.epoch = 0; w.rng = g_rng(); w.slow = keep_slow; w.failed = keep_failed;
  };
  std::vector<Walker> host((size_t)nwalk);
  memset(host.data(), 0, host.size() * sizeof(Walker));
  for (auto &w : host) restart(w);
  const uint32_t lazy_mask = argc > 12 ? (uint32_t)strtoul(argv[12], nullptr, 0) : 0u;   // e.g. 1023: one reducible state in 1024 is entered
  Walker *dw; Found *df; int *dn, *db;
  unsigned long long *dq;
  CHECK(hipMalloc(&dq, 65 * 32 * sizeof(unsigned long long)));
  CHECK(hipMemset(dq, 0, 65 * 32 * sizeof(unsigned long long)));
  std::vector<unsigned long long> hq(65 * 32);
  const int maxfound = 65536;
  CHECK(hipMalloc(&dw, host.size() * sizeof(Walker)));
  CHECK(hipMalloc(&df, (size_t)maxfound * sizeof(Found)));
  CHECK(hipMalloc(&dn, sizeof(int)));
  CHECK(hipMalloc(&db, sizeof(int)));
  CHECK(hipMemcpy(dw, host.data(), host.size() * sizeof(Walker), hipMemcpyHostToDevice));
  std::vector<Found> hf((size_t)maxfound);
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  double last_report = 0;
  uint64_t total = 0, restarts = 0, descents = 0;
  int launches = 0;
  const int target = argc > 13 ? atoi(argv[13]) : (NDIM == 4 ? 47 : 23);   // stop at this rank
  while (elapsed() < seconds && best > target) {
    int zero = 0;
    CHECK(hipMemcpy(dn, &zero, sizeof zero, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, &best, sizeof best, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(walk_kernel, dim3((unsigned)((nwalk + 3) / 4)), dim3(256), 0, 0, dw, nwalk, flips, plus_interval, margin, df, dn, maxfound, db, thr, dq, lazy_mask);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    ++launches;
    int nf = 0;
    CHECK(hipMemcpy(&nf, dn, sizeof nf, hipMemcpyDeviceToHost));
    nf = std::min(nf, maxfound);
    if (nf) CHECK(hipMemcpy(hf.data(), df, (size_t)nf * sizeof(Found), hipMemcpyDeviceToHost));
    for (int q = 0; q < nf; ++q) {
      const Scheme &s = hf[q].s;
      if (s.r < 1 || s.r > 64 || !verify(&s)) { fprintf(stderr, "a scheme of rank %d from the device does not verify\n", s.r); return 4; }
      ++descents;
      pool_add(s);
      if (s.r == 49) { if (strassen_squared_signature(s)) ++g_sig_strassen; else ++g_sig_other; }
      if (s.r < best) {
        best = s.r;
        printf("# rank %d after %.1f s (%.3g flips), quality %d\n", s.r, elapsed(), (double)total, quality(s));
        for (int r = 0; r < s.r; ++r) printf("{0x%04x, 0x%04x, 0x%04x},\n", s.t[r][0], s.t[r][1], s.t[r][2]);
        fflush(stdout);
      }
    }
    CHECK(hipMemcpy(host.data(), dw, host.size() * sizeof(Walker), hipMemcpyDeviceToHost));
    int hist[65] = {0};
    uint64_t slow = 0, failed = 0;
    for (auto &w : host) {
      hist[w.s.r]++;
      total += w.done;
      slow += w.slow; failed += w.failed;
      // a walk is over at the path limit, or when the front has moved below where walks start from
      if (w.epoch >= path_limit || w.s.r > best + margin + span) { restart(w); ++restarts; }
    }
    CHECK(hipMemcpy(dw, host.data(), host.size() * sizeof(Walker), hipMemcpyHostToDevice));
    if (elapsed() - last_report > 60.0 || launches <= 2) {
      last_report = elapsed();
      printf("# %.0f s: %.3g flips (%.3g /s), best %d; walks by rank", elapsed(), (double)total, (double)total / elapsed(), best);
      for (int r = best; r <= best + 6 && r <= 64; ++r) printf(" %d:%d", r, hist[r]);
      printf("; pools");
      for (int r = best; r <= best + 3 && r <= 64; ++r) {
        long qs = 0; int live = 0;
        for (auto &s : g_pool[r]) { const int q = quality(s); qs += q; live += q > 0; }
        printf(" %d:%zu(live %d, q %.1f)", r, g_pool[r].size(), live, g_pool[r].empty() ? 0.0 : (double)qs / (double)g_pool[r].size());
      }
      printf("; arrived (with a flip to offer)");
      for (int r = best; r <= best + 2 && r <= 64; ++r) printf(" %d:%llu(%llu)", r, (unsigned long long)g_arrived[r], (unsigned long long)g_arrived_live[r]);
      printf("; rank 49 with the factor ranks of Strassen squared %llu, other %llu", (unsigned long long)g_sig_strassen, (unsigned long long)g_sig_other);
      printf("; %llu descents, %llu restarts, serial path %.3g of flips, declined reductions %.3g\n", (unsigned long long)descents, (unsigned long long)restarts,
             (double)slow / (double)total, (double)failed / (double)total);
      CHECK(hipMemcpy(hq.data(), dq, hq.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      for (int r = best + 3; r >= best - 1 && r >= 1; --r) {   // flips offered by every reduced scheme the walks came upon (taken or declined)
        if (r > 64) continue;
        printf("#   reductions to rank %d (threshold %d) by flips offered 0..31+:", r, (int)thr.t[r]);
        for (int q = 0; q < 32; ++q) printf(" %llu", hq[(size_t)r * 32 + q]);
        printf("\n");
      }
      fflush(stdout);
      write_pools(pool_out, best);
    }
  }
  write_pools(pool_out, best);
  printf("# best rank %d, %.3g flips in %.0f s\n", best, (double)total, elapsed());
  return best <= target ? 0 : 1;
}
