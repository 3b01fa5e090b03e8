/* tools/l4_profile_shim.c -- LD_PRELOAD shim that TIMES (does not replace) the non-recursive base cases of
 * M4RI's L4 routines, to see what is left on the CPU once the products run on the GPU:
 *
 *   gcc -O2 -fPIC -shared -o build/l4_profile_shim.so tools/l4_profile_shim.c -ldl
 *   LD_PRELOAD="build/l4_profile_shim.so m4ri_amd/libm4ri_amd.so" oracle/_ref/l4_timing_driver 32768
 *
 * Every wrapped function forwards to the next definition in link order (dlsym RTLD_NEXT); matrices are
 * opaque here.  Only leaf-like functions are wrapped, so the inclusive times do not double count. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <time.h>

static double now(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

enum { N = 8 };
static const char *names[N] = {"_mzd_ple_russian", "_mzd_trsm_upper_left_russian", "_mzd_trsm_lower_left_russian",
                               "mzd_apply_p_right_trans_even_capped", "mzd_apply_p_left", "mzd_copy", "_mzd_trsm_pack", "mzd_apply_p_right_trans_tri"};
static double secs[N];
static long calls[N];

#define WRAP(idx, ret, name, params, args)                                \
  ret name params {                                                       \
    static ret(*next) params;                                             \
    if (!next) next = (ret(*) params)dlsym(RTLD_NEXT, #name);             \
    const double t = now();                                               \
    ret r = next args;                                                    \
    secs[idx] += now() - t;                                               \
    calls[idx] += 1;                                                      \
    return r;                                                             \
  }
#define WRAPV(idx, name, params, args)                                    \
  void name params {                                                      \
    static void (*next) params;                                           \
    if (!next) next = (void (*) params)dlsym(RTLD_NEXT, #name);           \
    const double t = now();                                               \
    next args;                                                            \
    secs[idx] += now() - t;                                               \
    calls[idx] += 1;                                                      \
  }

WRAP(0, int, _mzd_ple_russian, (void *A, void *P, void *Q, int k), (A, P, Q, k))
WRAPV(1, _mzd_trsm_upper_left_russian, (void const *U, void *B, int k), (U, B, k))
WRAPV(2, _mzd_trsm_lower_left_russian, (void const *L, void *B, int k), (L, B, k))
WRAPV(3, mzd_apply_p_right_trans_even_capped, (void *A, void const *P, int start_row, int start_col), (A, P, start_row, start_col))
WRAPV(4, mzd_apply_p_left, (void *A, void const *P), (A, P))
WRAP(5, void *, mzd_copy, (void *DST, void const *A), (DST, A))
WRAPV(7, mzd_apply_p_right_trans_tri, (void *A, void const *Q), (A, Q))

static void __attribute__((destructor)) report(void) {
  fprintf(stderr, "l4_profile_shim: inclusive seconds in the wrapped base cases\n");
  for (int i = 0; i < N; ++i)
    if (calls[i]) fprintf(stderr, "  %-40s %8ld calls %9.3f s\n", names[i], calls[i], secs[i]);
}
