// tools/tsan_threads.cpp -- the threading contract of the host entry points under ThreadSanitizer (developer tool; the
// bit-exactness of the same pattern is tests/test_gpu_threads.py).  Four host threads call mzd_mul / mzd_addmul / mzd_mul_m4rm on
// their own matrices, pin / chain / unpin, two of them go through mzd_mul_mp on virtual ranks, all at once; every thread compares
// with the product the main thread computed alone beforehand.  Built by tools/build_tsan.sh against a -fsanitize=thread build of
// the library's HOST code (the device code is not instrumented; races between kernels are the determinism stress tests' business).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../include/m4ri_amd.h"

static uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static mzd_t *random_matrix(rci_t r, rci_t c, uint64_t seed) {
  mzd_t *M = m4ri_amd_mzd_init(r, c);
  for (rci_t i = 0; i < r; ++i) {
    for (wi_t k = 0; k < M->width; ++k) M->data[(int64_t)i * M->rowstride + k] = splitmix(seed);
    M->data[(int64_t)i * M->rowstride + M->width - 1] &= M->high_bitmask;
  }
  return M;
}

static bool same(const mzd_t *X, const mzd_t *Y) {
  for (rci_t i = 0; i < X->nrows; ++i)
    if (memcmp(X->data + (int64_t)i * X->rowstride, Y->data + (int64_t)i * Y->rowstride, (size_t)X->width * 8)) return false;
  return true;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 3;
  // round 6: argv[2] = g > 1 makes every rank of the distributed Strassen schedule multiply its sub-products g at a time as batched
  // products (m4ri_amd_mul_batch_dev from the ranks' threads; multi.hip: M4RI_AMD_MULTI_GROUP)
  if (argc > 2 && atoi(argv[2]) > 1) setenv("M4RI_AMD_MULTI_GROUP", argv[2], 1);
  const int shapes[4][3] = {{1100, 1290, 1411}, {2048, 2048, 4096}, {513, 700, 65}, {1536, 1536, 1536}};
  if (m4ri_amd_init(0)) { fprintf(stderr, "no device\n"); return 2; }
  const int ids[3] = {0, 0, 0};
  m4ri_amd_set_devices(3, ids);
  m4ri_amd_set_multi_threshold(512);
  mzd_t *A[4], *B[4], *want[4];
  for (int i = 0; i < 4; ++i) {
    A[i] = random_matrix(shapes[i][0], shapes[i][1], 100 + i);
    B[i] = random_matrix(shapes[i][1], shapes[i][2], 200 + i);
    want[i] = mzd_mul(NULL, A[i], B[i], 0);
  }
  int bad[4] = {0, 0, 0, 0};
  std::vector<std::thread> th;
  for (int i = 0; i < 4; ++i)
    th.emplace_back([&, i] {
      if (m4ri_amd_init(0)) { bad[i] = 100; return; }
      for (int r = 0; r < reps; ++r) {
        mzd_t *C = mzd_mul(NULL, A[i], B[i], r % 2 ? 0 : 256);
        bad[i] += !same(C, want[i]);
        mzd_addmul(C, A[i], B[i], 0);  // C ^= A*B: zero
        mzd_t *Z = m4ri_amd_mzd_init(C->nrows, C->ncols);
        bad[i] += !same(C, Z);
        mzd_t *D = mzd_mul_m4rm(NULL, A[i], B[i], 0);
        bad[i] += !same(D, want[i]);
        if (i < 2) {  // the multi-device entry point (virtual ranks), two threads at once
          mzd_t *E = mzd_mul_mp(NULL, A[i], B[i], 0);
          bad[i] += !same(E, want[i]);
          m4ri_amd_result_free(E);
        } else {      // residency: pin, chain on the device, unpin
          m4ri_amd_pin(A[i]); m4ri_amd_pin(B[i]); m4ri_amd_pin(Z);
          mzd_addmul(Z, A[i], B[i], 0);
          m4ri_amd_unpin(Z); m4ri_amd_unpin(B[i]); m4ri_amd_unpin(A[i]);
          bad[i] += !same(Z, want[i]);
        }
        m4ri_amd_result_free(C); m4ri_amd_result_free(D); m4ri_amd_mzd_free(Z);
      }
    });
  // round 5: a fifth thread keeps two products in flight on the two lanes of the distributed path while the others run (own distributed
  // matrices: the lanes' rule), and a sixth asks about / syncs the pins of thread 2 and 3 from outside (the pin table's entries are
  // guarded by the lock of the device the pin lives on, not by the caller's)
  int bad_lane = 0;
  th.emplace_back([&] {
    if (m4ri_amd_init(0)) { bad_lane = 100; return; }
    m4ri_amd_dmat *dA = m4ri_amd_dmat_create(shapes[1][0], shapes[1][1], M4RI_AMD_LAYOUT_CYCLIC1);
    m4ri_amd_dmat *dB = m4ri_amd_dmat_create(shapes[1][1], shapes[1][2], M4RI_AMD_LAYOUT_CYCLIC1);
    m4ri_amd_dmat *dC[2] = {m4ri_amd_dmat_create(shapes[1][0], shapes[1][2], M4RI_AMD_LAYOUT_CYCLIC1), m4ri_amd_dmat_create(shapes[1][0], shapes[1][2], M4RI_AMD_LAYOUT_CYCLIC1)};
    if (!dA || !dB || !dC[0] || !dC[1] || m4ri_amd_dmat_upload(dA, A[1]) || m4ri_amd_dmat_upload(dB, B[1])) { bad_lane = 100; return; }
    for (int r = 0; r < 2 * reps; ++r) bad_lane += m4ri_amd_dmat_mul_lane(dC[r & 1], dA, dB, 0, 0, 0, r & 1) != 0;
    for (int k = 0; k < 2; ++k) {
      mzd_t *H = m4ri_amd_mzd_init(shapes[1][0], shapes[1][2]);
      bad_lane += m4ri_amd_dmat_download(dC[k], H) != 0 || !same(H, want[1]);
      m4ri_amd_mzd_free(H);
    }
    m4ri_amd_dmat_free(dA); m4ri_amd_dmat_free(dB); m4ri_amd_dmat_free(dC[0]); m4ri_amd_dmat_free(dC[1]);
  });
  th.emplace_back([&] {
    for (int r = 0; r < 200 * reps; ++r)
      for (int i = 2; i < 4; ++i) { (void)m4ri_amd_is_pinned(A[i]); (void)m4ri_amd_is_pinned(B[i]); if (r % 16 == 0) (void)m4ri_amd_sync(B[i]); }
  });
  // round 6, third session: two more threads stream SMALL products through the entry points -- the library's host routine (thread-local
  // scratch, the device lock released while it runs) and, above its bound, the one-launch small leaf -- while the others run
  int bad_small[2] = {0, 0};
  for (int k = 0; k < 2; ++k)
    th.emplace_back([&, k] {
      if (m4ri_amd_init(0)) { bad_small[k] = 100; return; }
      const int sh[4][3] = {{64, 64, 64}, {200, 300, 100}, {448, 448, 448}, {600, 520, 700}};
      for (int q = 0; q < 4; ++q) {
        mzd_t *X = random_matrix(sh[q][0], sh[q][1], 300 + 10 * k + q), *Y = random_matrix(sh[q][1], sh[q][2], 400 + 10 * k + q);
        mzd_t *first = mzd_mul(NULL, X, Y, 0);
        for (int r = 0; r < 20 * reps; ++r) {
          mzd_t *C = mzd_mul(NULL, X, Y, 0);
          bad_small[k] += !same(C, first);
          mzd_addmul(C, X, Y, 0);
          for (rci_t i = 0; i < C->nrows; ++i)
            for (wi_t w = 0; w < C->width; ++w) bad_small[k] += C->data[(int64_t)i * C->rowstride + w] != 0;
          m4ri_amd_result_free(C);
        }
        m4ri_amd_result_free(first); m4ri_amd_mzd_free(X); m4ri_amd_mzd_free(Y);
      }
    });
  for (auto &t : th) t.join();
  printf("small-product threads: %d + %d mismatches (%lld products took the host routine)\n", bad_small[0], bad_small[1], (long long)m4ri_amd_small_product_count());
  printf("lanes thread: %d mismatches\n", bad_lane);
  int total = bad_lane + bad_small[0] + bad_small[1];
  for (int i = 0; i < 4; ++i) { printf("thread %d: %d mismatches\n", i, bad[i]); total += bad[i]; }
  m4ri_amd_release_workspace();
  printf("%s\n", total ? "TSAN_THREADS FAILED" : "TSAN_THREADS results ok");
  return total != 0;
}
