// tools/xcc_map.hip -- developer probe: which XCD does workgroup b of a 1-D grid land on, and when?  The leaf's tile
// order assumes b % 8 (m4rm8q_leaf.hip: "XCD remap"); this records XCC_ID, the CU and the start time of every workgroup
// of a grid shaped like the bench's leaf launch (10976 workgroups x 512 threads, 128 KiB of LDS each, ~`spin` us long).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcc_map.hip -o build/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(512) void probe(uint32_t *xcc, uint32_t *hwid, uint64_t *t0, uint64_t *t1, int spin_ticks) {
  __shared__ unsigned char big[128 * 1024];
  big[threadIdx.x] = (unsigned char)threadIdx.x;
  __syncthreads();
  const uint64_t start = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    xcc[blockIdx.x]  = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID, bits 0..3
    hwid[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    t0[blockIdx.x]   = start;
  }
  while (__builtin_amdgcn_s_memrealtime() - start < (uint64_t)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) t1[blockIdx.x] = __builtin_amdgcn_s_memrealtime() + big[17];
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10976, spin = argc > 2 ? atoi(argv[2]) : 6500;  // 100 MHz ticks: 65 us
  uint32_t *dx, *dh; uint64_t *d0, *d1;
  CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dh, n * 4)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(n), dim3(512), 0, 0, dx, dh, d0, d1, spin); CK(hipDeviceSynchronize()); }
  std::vector<uint32_t> x(n), h(n); std::vector<uint64_t> a(n), b(n);
  CK(hipMemcpy(x.data(), dx, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data(), dh, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost));
  int same = 0; int hist[8][16] = {};
  for (int i = 0; i < n; ++i) { same += (int)(x[i] & 15) == (i & 7); hist[i & 7][x[i] & 15]++; }
  printf("grid %d: XCC_ID == blockIdx %% 8 for %d workgroups (%.1f %%)\n", n, same, 100.0 * same / n);
  for (int r = 0; r < 8; ++r) { printf("  b%%8=%d:", r); for (int c = 0; c < 8; ++c) printf(" %5d", hist[r][c]); printf("\n"); }
  // start-time skew inside the groups of 32 consecutive "lids" that the leaf expects to be co-resident on one XCD
  uint64_t base = a[0]; for (int i = 0; i < n; ++i) if (a[i] < base) base = a[i];
  printf("first 24 workgroups: b, xcc, hw_id, start (us), end (us)\n");
  for (int i = 0; i < 24; ++i) printf("  %5d %2u %08x %9.2f %9.2f\n", i, x[i] & 15, h[i], (a[i] - base) / 100.0, (b[i] - base) / 100.0);
  // for each XCD: sort its workgroups by start time and report how far apart (in us) the starts of workgroups
  // whose remapped lids fall into the same block of 32 are
  double worst = 0, sum = 0; int groups = 0;
  const int per = n / 8;
  for (int xc = 0; xc < 8 && n % 8 == 0; ++xc)
    for (int g0 = 0; g0 + 32 <= per; g0 += 32) {
      uint64_t lo = ~0ull, hi = 0;
      for (int j = 0; j < 32; ++j) { const int bidx = (g0 + j) * 8 + xc; if (a[bidx] < lo) lo = a[bidx]; if (a[bidx] > hi) hi = a[bidx]; }
      const double d = (hi - lo) / 100.0; sum += d; ++groups; if (d > worst) worst = d;
    }
  if (groups) printf("start skew inside blocks of 32 consecutive lids (one product per XCD): mean %.2f us, worst %.2f us, workgroup length %.1f us\n", sum / groups, worst, spin / 100.0);
  return 0;
}
