/* tests/integration_stub.c -- the binding hunk of INTEGRATION.md section 2, as a compilable unit: what a
 * maintainer would put into m4ri/strassen.c behind a configure switch.  tests/test_cabi.py compiles it
 * (syntax only) against M4RI's own headers and include/m4ri_amd.h, so the documented stub cannot rot. */
#include <m4ri/m4ri.h>

#define __M4RI_HAVE_M4RI_AMD 1

#if __M4RI_HAVE_M4RI_AMD
#define M4RI_AMD_NO_MZD_T          /* use M4RI's own mzd_t: same 64 bytes */
#include <m4ri_amd.h>
#include <dlfcn.h>

/* the accelerator's entry points carry M4RI's names, so bind them under private aliases */
static mzd_t *(*amd_mul)(mzd_t *, mzd_t const *, mzd_t const *, int);
static mzd_t *(*amd_addmul)(mzd_t *, mzd_t const *, mzd_t const *, int);

static void __attribute__((constructor)) m4ri_amd_bind(void) {
  void *h = dlopen("libm4ri_amd.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) m4ri_die("m4ri: built --with-m4ri-amd but libm4ri_amd.so is missing: %s\n", dlerror());
  amd_mul    = (mzd_t *(*)(mzd_t *, mzd_t const *, mzd_t const *, int))dlsym(h, "mzd_mul");
  amd_addmul = (mzd_t *(*)(mzd_t *, mzd_t const *, mzd_t const *, int))dlsym(h, "mzd_addmul");
}

mzd_t *mzd_mul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return amd_mul(C, A, B, cutoff); }
mzd_t *mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return amd_addmul(C, A, B, cutoff); }
#endif
