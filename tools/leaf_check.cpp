// tools/leaf_check.cpp -- developer harness for the M4RM leaf kernel (not part of the product path,
// not a test the driver runs): checks gf2_launch_m4rm_leaf against a definitional CPU multiply on
// ragged/batched/strided shapes, then times the bench-sized launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I m4ri_amd/csrc tools/leaf_check.cpp \
//         m4ri_amd/csrc/m4rm_leaf.hip m4ri_amd/csrc/a4_pack.hip m4ri_amd/csrc/m4rm8q_leaf.hip -o build/leaf_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gf2_common.h"

extern "C" hipError_t gf2_launch_m4rm_leaf(hipStream_t stream, LeafArgs a, int rg);
extern "C" hipError_t gf2_launch_m4rm_leaf_variant(hipStream_t stream, LeafArgs a, int rg, int ug, int pipe);
extern "C" int64_t gf2_m4rm8_a4_words(int64_t m, int64_t l, int64_t batch);
extern "C" hipError_t gf2_launch_a4_pack_rot(hipStream_t stream, LeafArgs a, word *a4_ws, int rot);
extern "C" hipError_t gf2_launch_m4rm8q(hipStream_t stream, LeafArgs a, word *a4_ws);
static word *g_a7 = nullptr; static int64_t g_a7_words = 0;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
// pipe: variant bits of the two-phase kernel (1 = software-pipelined use phase, 2 = B rows staged
// through LDS); 11 = generation 4.  (The generation-2, generation-3, generation-5 and half-builder
// experiments were removed with their kernels in round 2; their measurements are in DESIGN.md 3.1.)
static hipError_t launch(LeafArgs a, int rg, int ug, int pipe) {
  if (pipe == 11) {  // generation 4: 8-bit tables, 64-byte entries, 4096 x 512 tiles
    const int64_t need = gf2_m4rm8_a4_words(a.m, a.l, a.batch);
    if (need > g_a7_words) { if (g_a7) (void)hipFree(g_a7); CK(hipMalloc(&g_a7, need * 8)); g_a7_words = need; }
    CK(gf2_launch_a4_pack_rot(0, a, g_a7, 1));
    return gf2_launch_m4rm8q(0, a, g_a7);
  }
  return gf2_launch_m4rm_leaf_variant(0, a, rg, ug, pipe);
}


static uint64_t sm_state;
static uint64_t splitmix() {
  uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// definitional product: C[i,:] ^= B[j,:] for every set bit (i,j) of A
static void cpu_mul(std::vector<word> &C, int64_t cs, const std::vector<word> &A, int64_t as,
                    const std::vector<word> &B, int64_t bs, int m, int l, int n, bool add) {
  const int wn = (n + 63) / 64;
  for (int i = 0; i < m; ++i) {
    word *c = &C[(size_t)i * cs];
    if (!add) for (int w = 0; w < wn; ++w) c[w] = 0;
    for (int j = 0; j < l; ++j)
      if ((A[(size_t)i * as + j / 64] >> (j % 64)) & 1) {
        const word *b = &B[(size_t)j * bs];
        for (int w = 0; w < wn; ++w) c[w] ^= b[w];
      }
  }
}

static void fill(std::vector<word> &M, int64_t stride, int rows, int cols, bool junk_outside) {
  const int w = (cols + 63) / 64;
  for (int r = 0; r < rows; ++r)
    for (int64_t k = 0; k < stride; ++k) {
      word v = splitmix();
      if (k >= w) v = junk_outside ? v : 0;
      else if (k == w - 1 && cols % 64) v &= (~0ull) >> (64 - cols % 64);
      M[(size_t)r * stride + k] = v;
    }
}

static int check(int m, int l, int n, int batch, int ksplit, int mode, int rg, int pad, int ug = 0, int pipe = 0) {
  const int wa = (l + 63) / 64, wn = (n + 63) / 64;
  const int64_t as = wa + pad, bs = wn + pad, cs = wn + pad;
  const int64_t abs_ = (int64_t)m * as + 3, bbs = (int64_t)l * bs + 5, cbs = (int64_t)m * cs + 7;
  std::vector<word> A(abs_ * batch), B(bbs * batch), C(cbs * batch), Cref;
  for (int b = 0; b < batch; ++b) {
    std::vector<word> a((size_t)m * as), bb((size_t)l * bs), c((size_t)m * cs);
    fill(a, as, m, l, false);  // A: zero excess (engine invariant), padding words zero
    fill(bb, bs, l, n, false);
    fill(c, cs, m, n, true);   // C: junk beyond the row's words must survive
    memcpy(&A[b * abs_], a.data(), a.size() * 8);
    memcpy(&B[b * bbs], bb.data(), bb.size() * 8);
    memcpy(&C[b * cbs], c.data(), c.size() * 8);
  }
  Cref = C;
  for (int b = 0; b < batch; ++b) {
    std::vector<word> a(A.begin() + b * abs_, A.begin() + b * abs_ + (size_t)m * as);
    std::vector<word> bb(B.begin() + b * bbs, B.begin() + b * bbs + (size_t)l * bs);
    std::vector<word> c(Cref.begin() + b * cbs, Cref.begin() + b * cbs + (size_t)m * cs);
    cpu_mul(c, cs, a, as, bb, bs, m, l, n, mode != 0);
    memcpy(&Cref[b * cbs], c.data(), c.size() * 8);
  }
  word *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dB, B.size() * 8)); CK(hipMalloc(&dC, C.size() * 8));
  CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
  LeafArgs a{};
  a.A = dA; a.B = dB; a.C = dC;
  a.a_stride = as; a.b_stride = bs; a.c_stride = cs;
  a.a_bs = abs_; a.b_bs = bbs; a.c_bs = cbs;
  a.m = m; a.l = l; a.n = n; a.batch = batch; a.ksplit = ksplit; a.mode = mode;
  CK(launch(a, rg, ug, pipe));
  CK(hipDeviceSynchronize());
  std::vector<word> Cg(C.size());
  CK(hipMemcpy(Cg.data(), dC, C.size() * 8, hipMemcpyDeviceToHost));
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  size_t bad = 0;
  for (size_t i = 0; i < Cg.size(); ++i) if (Cg[i] != Cref[i]) { if (!bad) fprintf(stderr, "  first mismatch at word %zu: got %016llx want %016llx\n", i, (unsigned long long)Cg[i], (unsigned long long)Cref[i]); ++bad; }
  printf("check m=%d l=%d n=%d batch=%d ksplit=%d mode=%d rg=%d ug=%d pipe=%d pad=%d : %s (%zu bad words)\n", m, l, n, batch, ksplit, mode, rg, ug, pipe, pad, bad ? "FAIL" : "ok", bad);
  return bad != 0;
}

static bool g_share_b = false;  // --traffic ... 1: every product of the batch reads the SAME B (its HBM traffic vanishes)
static void timeit(int m, int l, int n, int batch, int ksplit, int rg, int reps, int ug = 0, int pipe = 0) {
  const int wa = (l + 63) / 64, wn = (n + 63) / 64;
  const size_t asz = (size_t)m * wa * batch, bsz = (size_t)l * wn * batch, csz = (size_t)m * wn * batch;
  word *dA, *dB, *dC;
  CK(hipMalloc(&dA, asz * 8)); CK(hipMalloc(&dB, bsz * 8)); CK(hipMalloc(&dC, csz * 8));
  std::vector<word> h(asz > bsz ? asz : bsz);
  for (auto &x : h) x = splitmix();
  CK(hipMemcpy(dA, h.data(), asz * 8, hipMemcpyHostToDevice));
  for (auto &x : h) x = splitmix();
  CK(hipMemcpy(dB, h.data(), bsz * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dC, 0, csz * 8));
  LeafArgs a{};
  a.A = dA; a.B = dB; a.C = dC;
  a.a_stride = wa; a.b_stride = wn; a.c_stride = wn;
  a.a_bs = (int64_t)m * wa; a.b_bs = g_share_b ? 0 : (int64_t)l * wn; a.c_bs = (int64_t)m * wn;
  a.m = m; a.l = l; a.n = n; a.batch = batch; a.ksplit = ksplit; a.mode = ksplit > 1 ? 1 : 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(launch(a, rg, ug, pipe)); CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, 0));
    CK(launch(a, rg, ug, pipe));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms; sum += ms;
  }
  const double ops = (double)m * l * n * batch;
  printf("time m=%d l=%d n=%d batch=%d ksplit=%d rg=%d ug=%d pipe=%d : best %.3f ms avg %.3f ms  -> %.3e bit-MAC/s (best)\n",
         m, l, n, batch, ksplit, rg, ug, pipe, best, sum / reps, ops / (best * 1e-3));
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main(int argc, char **argv) {
  sm_state = 12345;
  int fails = 0;
  const int rgs[3] = {32, 24, 16};
  if (!(argc > 1 && (!strcmp(argv[1], "--one") || !strcmp(argv[1], "--v4") || !strcmp(argv[1], "--shape") || !strcmp(argv[1], "--traffic"))))
  for (int rg : rgs) {
    fails += check(1024, 1024, 2048, 1, 1, 0, rg, 0, 4, 0);
    fails += check(1000, 777, 1234, 1, 1, 0, rg, 1, 4, 0);
    fails += check(1000, 777, 1234, 2, 1, 1, rg, 3, 4, 0);
    fails += check(1, 1, 1, 1, 1, 0, rg, 0, 4, 0);
    fails += check(3, 131, 257, 1, 1, 0, rg, 0, 4, 0);
    fails += check(2100, 300, 4100, 2, 3, 1, rg, 2, 4, 0);
    fails += check(64, 64, 64, 5, 1, 0, rg, 0, 4, 0);
    fails += check(193, 65, 65, 1, 2, 1, rg, 0, 4, 0);
  }
  if (argc > 1 && !strcmp(argv[1], "--check-only")) return fails != 0;
  if (argc > 6 && !strcmp(argv[1], "--traffic")) {  // --traffic m l n batch shareB : generation 4 once, for rocprofv3 --pmc
    g_share_b = atoi(argv[6]) != 0;
    timeit(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), 1, 32, 1, 1, 11);
    return 0;
  }
  if (argc > 6 && !strcmp(argv[1], "--shape")) {  // --shape m l n batch pipe : time one variant on one shape
    timeit(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), 1, 32, 3, 1, atoi(argv[6]));
    return 0;
  }
  if (argc > 5 && !strcmp(argv[1], "--one")) {  // --one rg ug pipe batch [reps] : profile a single variant (many reps: a power trace)
    timeit(8192, 8192, 8192, atoi(argv[5]), 1, atoi(argv[2]), argc > 6 ? atoi(argv[6]) : 3, atoi(argv[3]), atoi(argv[4]));
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--v4")) {  // generation 4 alone (build with -DK8Q_BUILDER_HALF=1 for the control experiment)
    const int v[][3] = {{32, 1, 11}};
    for (auto &x : v) {
      fails += check(1000, 777, 1234, 2, 1, 1, x[0], 3, x[1], x[2]);
      fails += check(2100, 300, 4100, 2, 3, 1, x[0], 2, x[1], x[2]);
      fails += check(5000, 1111, 700, 1, 1, 0, x[0], 1, x[1], x[2]);
      fails += check(4096, 96, 512, 2, 1, 1, x[0], 0, x[1], x[2]);
      timeit(8192, 8192, 8192, 64, 1, x[0], 3, x[1], x[2]);
      timeit(8192, 8192, 8192, 343, 1, x[0], 2, x[1], x[2]);
    }
    printf("%s\n", fails ? "LEAF_CHECK FAILED" : "LEAF_CHECK ALL OK");
    return fails != 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--variants")) {
    const int v[][3] = {{32, 4, 0}, {24, 4, 0}, {16, 4, 0}, {32, 1, 11}};
    for (auto &x : v) {
      fails += check(1000, 777, 1234, 2, 1, 1, x[0], 3, x[1], x[2]);
      fails += check(2100, 300, 4100, 2, 3, 1, x[0], 2, x[1], x[2]);
      fails += check(193, 65, 65, 1, 1, 0, x[0], 0, x[1], x[2]);
      fails += check(1024, 1024, 2048, 1, 1, 0, x[0], 0, x[1], x[2]);
      fails += check(2048, 512, 1024, 1, 1, 0, x[0], 0, x[1], x[2]);
      fails += check(3000, 1000, 3000, 1, 1, 0, x[0], 1, x[1], x[2]);
      fails += check(1, 1, 1, 1, 1, 0, x[0], 0, x[1], x[2]);
      fails += check(64, 64, 64, 5, 1, 0, x[0], 0, x[1], x[2]);
      fails += check(5000, 1111, 700, 1, 1, 0, x[0], 1, x[1], x[2]);
      fails += check(4096, 96, 512, 2, 1, 1, x[0], 0, x[1], x[2]);
      timeit(8192, 8192, 8192, 64, 1, x[0], 3, x[1], x[2]);
      if (x[0] == 40) timeit(10240, 8192, 8192, 64, 1, x[0], 3, x[1], x[2]);
    }
    printf("%s\n", fails ? "LEAF_CHECK FAILED" : "LEAF_CHECK ALL OK");
    return fails != 0;
  }
  for (int rg : rgs) {
    timeit(8192, 8192, 8192, 8, 1, rg, 5, 4, 0);
    timeit(16384, 16384, 16384, 1, 4, rg, 5, 4, 0);
  }
  timeit(4096, 4096, 4096, 64, 1, 32, 5, 4, 0);
  printf("%s\n", fails ? "LEAF_CHECK FAILED" : "LEAF_CHECK ALL OK");
  return fails != 0;
}
