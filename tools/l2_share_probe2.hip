// tools/l2_share_probe2.hip -- developer probe, second part: `readers` workgroups on XCD 0 (blocks 0, 8, 16, ...; 128 KiB of
// LDS each, so one per CU) read the SAME buffer the way the leaf reads its packed A: per step a 16 KiB chunk = 128 lines,
// every lane group takes 16 bytes of "its" line, eight passes (g = 0..7) cover the line, `gap_ns` apart; steps `step_ns`
// apart.  Under rocprofv3 --pmc TCC_EA0_RDREQ_sum: requests / lines = how often a line is fetched from the fabric.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_share_probe2.hip -o build/l2_share_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// BUFLOAD: the loads go through a raw buffer descriptor, as in the leaf
template <bool BUFLOAD>
__global__ __launch_bounds__(512) void probe(const unsigned char *buf, int chunks, int chunk_stride, int readers, int gap_ticks, int step_ticks,
                                             uint32_t *out) {
  __shared__ unsigned char big[128 * 1024];
  big[threadIdx.x] = 1;
  if ((blockIdx.x & 7) != 0 || (int)(blockIdx.x >> 3) >= readers) return;
  const int rgrp = threadIdx.x >> 2;  // 0..127: the line of the chunk this lane group reads
  uint32_t acc = 0;
  for (int q = 0; q < chunks; ++q) {
    const uint64_t s0 = __builtin_amdgcn_s_memrealtime();
    for (int g = 0; g < 8; ++g) {
      const uint64_t g0 = __builtin_amdgcn_s_memrealtime();
      uint4 v;
      if (BUFLOAD) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(buf), (short)0, chunks * chunk_stride, 0x00020000);
        v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((size_t)q * chunk_stride + rgrp * 128 + g * 16), 0, 0));
      } else {
        v = *reinterpret_cast<const uint4 *>(buf + (size_t)q * chunk_stride + rgrp * 128 + g * 16);
      }
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
      while (__builtin_amdgcn_s_memrealtime() - g0 < (uint64_t)gap_ticks) __builtin_amdgcn_s_sleep(1);
    }
    while (__builtin_amdgcn_s_memrealtime() - s0 < (uint64_t)step_ticks) __builtin_amdgcn_s_sleep(2);
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc + big[17];
}

int main(int argc, char **argv) {
  const int readers = argc > 1 ? atoi(argv[1]) : 16, gap_ns = argc > 2 ? atoi(argv[2]) : 800, step_ns = argc > 3 ? atoi(argv[3]) : 6400;
  const int chunk_stride = argc > 4 ? atoi(argv[4]) : 32768;
  const int bufload = argc > 5 ? atoi(argv[5]) : 0;
  const int chunks = 2048;
  unsigned char *buf; uint32_t *out;
  CK(hipMalloc(&buf, (size_t)chunks * chunk_stride)); CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMemset(buf, 1, (size_t)chunks * chunk_stride)); CK(hipDeviceSynchronize());
  if (bufload) hipLaunchKernelGGL(probe<true>, dim3(256), dim3(512), 0, 0, buf, chunks, chunk_stride, readers, gap_ns / 10, step_ns / 10, out);
  else hipLaunchKernelGGL(probe<false>, dim3(256), dim3(512), 0, 0, buf, chunks, chunk_stride, readers, gap_ns / 10, step_ns / 10, out);
  CK(hipDeviceSynchronize());
  printf("readers %d gap %d ns step %d ns chunk stride %d %s: %d lines of 128 B read\n", readers, gap_ns, step_ns, chunk_stride,
         bufload ? "buffer loads" : "global loads", chunks * 128);
  return 0;
}
