// tools/l2_share_probe.hip -- developer probe: do two workgroups on ONE XCD that read the same lines share one fabric
// fetch?  Workgroups 0 and 8 (both XCD 0) stream the same buffer with 16-byte loads; the second starts `delay_us` later.
// Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum: requests x 128 B vs the buffer size tells whether the second reader hit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_share_probe.hip -o build/l2_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int MODE>  // 0: both readers on XCD 0 (blocks 0 and 8); 1: readers on XCD 0 and XCD 1 (blocks 0 and 1)
__global__ __launch_bounds__(256) void probe(const uint4 *buf, size_t n16, int delay_ticks, int chunk_ticks, uint32_t *out) {
  const bool second = MODE == 0 ? blockIdx.x == 8 : blockIdx.x == 1;
  if (!(blockIdx.x == 0 || second)) return;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  if (second) while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)delay_ticks) __builtin_amdgcn_s_sleep(4);
  uint32_t acc = 0;
  // paced like the leaf: 16 KiB (1024 x 16 B) per `chunk_ticks`
  for (size_t base = 0; base < n16; base += 1024) {
    const uint64_t c0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < 4; ++k) { const uint4 v = buf[base + k * 256 + threadIdx.x]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    while (__builtin_amdgcn_s_memrealtime() - c0 < (uint64_t)chunk_ticks) __builtin_amdgcn_s_sleep(2);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, delay_us = argc > 2 ? atoi(argv[2]) : 0, chunk_ns = argc > 3 ? atoi(argv[3]) : 1000;
  const size_t bytes = 32u << 20;
  uint4 *buf; uint32_t *out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 16 * 256 * 4));
  CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
  if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(16), dim3(256), 0, 0, buf, bytes / 16, delay_us * 100, chunk_ns / 10, out);
  else hipLaunchKernelGGL(probe<1>, dim3(16), dim3(256), 0, 0, buf, bytes / 16, delay_us * 100, chunk_ns / 10, out);
  CK(hipDeviceSynchronize());
  printf("mode %d delay %d us chunk %d ns: buffer %zu MiB = %zu lines of 128 B\n", mode, delay_us, chunk_ns, bytes >> 20, bytes / 128);
  return 0;
}
