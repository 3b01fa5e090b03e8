// small_host_timing.c -- the library's host routine for small products (m4ri_amd/csrc/small_host.cpp) next to the reference's
// mzd_mul / mzd_addmul on the same host matrices, both called from C (no ctypes overhead in the microseconds that are compared).
// Needs no GPU: m4ri_amd_small_mul_host is pure host code.  Lives under tests/ because it executes the reference checker
// (oracle/_ref/libm4ri_ref.so); a measurement program, not a pytest module.
//
//   cc -O2 -o /tmp/small_host_timing tests/small_host_timing.c -ldl && /tmp/small_host_timing [repo root]
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct mzd_t mzd_t;  // only handled through the reference's own constructors
typedef mzd_t *(*init_fn)(int, int);
typedef void (*free_fn)(mzd_t *);
typedef void (*rand_fn)(mzd_t *);
typedef mzd_t *(*mul_fn)(mzd_t *, const mzd_t *, const mzd_t *, int);
typedef int (*equal_fn)(const mzd_t *, const mzd_t *);
typedef int (*small_fn)(mzd_t *, const mzd_t *, const mzd_t *, int);
typedef mzd_t *(*window_fn)(const mzd_t *, int, int, int, int);

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, char **argv) {
  const char *root = argc > 1 ? argv[1] : ".";
  char path[4096];
  snprintf(path, sizeof path, "%s/oracle/_ref/libm4ri_ref.so", root);
  void *ref = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!ref) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  snprintf(path, sizeof path, "%s/m4ri_amd/libm4ri_amd.so", root);
  void *amd = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!amd) { fprintf(stderr, "%s\n", dlerror()); return 1; }
  init_fn mzd_init     = (init_fn)dlsym(ref, "mzd_init");
  free_fn mzd_free     = (free_fn)dlsym(ref, "mzd_free");
  rand_fn mzd_rand     = (rand_fn)dlsym(ref, "mzd_randomize");
  mul_fn ref_mul       = (mul_fn)dlsym(ref, "mzd_mul");
  mul_fn ref_addmul    = (mul_fn)dlsym(ref, "mzd_addmul");
  equal_fn mzd_equal   = (equal_fn)dlsym(ref, "mzd_equal");
  window_fn mzd_window = (window_fn)dlsym(ref, "mzd_init_window");
  small_fn small       = (small_fn)dlsym(amd, "m4ri_amd_small_mul_host");
  if (!mzd_init || !ref_mul || !small || !mzd_window) { fprintf(stderr, "missing symbol\n"); return 1; }

  static const int shapes[][3] = {
      {16, 16, 16},   {32, 32, 32},    {64, 64, 64},    {96, 96, 96},    {128, 128, 128}, {192, 192, 192}, {256, 256, 256},
      {320, 320, 320}, {384, 384, 384}, {400, 400, 400}, {448, 448, 448}, {512, 512, 512}, {576, 576, 576}, {640, 640, 640}, {768, 768, 768},
      {1024, 256, 256}, {256, 1024, 256}, {256, 256, 1024}, {2048, 2048, 16}, {16, 2048, 2048}, {2048, 16, 2048}, {1000, 10, 20},  {16, 4096, 16},  {4096, 16, 64},  {64, 64, 4096},
      {2048, 64, 64}, {8, 512, 512},   {512, 8, 512},   {512, 512, 8},   {100, 1000, 100}, {1, 64, 64},    {64, 1, 64},
      {200, 200, 1000}, {33, 777, 129}, {1024, 64, 1024}, {64, 16384, 64}};
  printf("%-18s | %9s %9s | %9s %9s | ratio mul, addmul (host routine / reference)\n", "m x l x n", "ref mul", "ref addmul", "host mul", "host addm");
  double worst = 0;
  for (size_t s = 0; s < sizeof shapes / sizeof shapes[0]; ++s) {
    const int m = shapes[s][0], l = shapes[s][1], n = shapes[s][2];
    mzd_t *A = mzd_init(m, l), *B = mzd_init(l, n), *C = mzd_init(m, n), *D = mzd_init(m, n);
    mzd_rand(A);
    mzd_rand(B);
    double t[4] = {1e30, 1e30, 1e30, 1e30};
    const int inner = (int64_t)m * l * n < (1 << 20) ? 200 : 20;
    for (int rep = 0; rep < 7; ++rep) {
      double t0 = now();
      for (int i = 0; i < inner; ++i) ref_mul(C, A, B, 0);
      double t1 = now();
      for (int i = 0; i < inner; ++i) ref_addmul(C, A, B, 0);
      double t2 = now();
      for (int i = 0; i < inner; ++i) small(D, A, B, 0);
      double t3 = now();
      for (int i = 0; i < inner; ++i) small(D, A, B, 1);
      double t4 = now();
      const double d[4] = {t1 - t0, t2 - t1, t3 - t2, t4 - t3};
      for (int k = 0; k < 4; ++k)
        if (d[k] / inner < t[k]) t[k] = d[k] / inner;
    }
    ref_mul(C, A, B, 0);
    small(D, A, B, 0);
    const int ok = mzd_equal(C, D);
    char name[64];
    snprintf(name, sizeof name, "%d x %d x %d", m, l, n);
    printf("%-18s | %9.2f %9.2f | %9.2f %9.2f | %5.2f %5.2f %s\n", name, t[0] * 1e6, t[1] * 1e6, t[2] * 1e6, t[3] * 1e6, t[2] / t[0], t[3] / t[1],
           ok ? "" : "MISMATCH");
    if ((int64_t)m * l * n <= (1 << 26) && t[2] / t[0] > worst) worst = t[2] / t[0];
    mzd_free(A); mzd_free(B); mzd_free(C); mzd_free(D);
  }
  printf("worst mul ratio among shapes of at most 2^26 bit operations: %.2f\n", worst);
  return 0;
}
