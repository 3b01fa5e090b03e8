// tools/half_line_probe.hip -- developer probe: does a read of one 64-byte half of a 128-byte L2 line cost the fabric
// 64 or 128 bytes?  (The leaf reads its packed A in half lines at different times; the fabric request counter shows two
// requests per A line, and FETCH_SIZE, which tallies every request at 64 B and is to be doubled for full-line streams,
// cannot tell.)  Times three streams over a buffer far beyond the 256 MiB Infinity Cache:
//   (a) N/2 bytes read contiguously, (b) the first half of every line of N bytes, (c) N bytes read contiguously.
// (b) ~ (a): half-line requests move 64 B;  (b) ~ (c): they move the whole line.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/half_line_probe.hip -o build/half_line_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int HALF>  // 0: contiguous 16-byte pieces; 1: pieces 0..3 of every 128-byte line
__global__ __launch_bounds__(256) void stream(const uint4 *buf, size_t pieces, uint32_t *out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (size_t)gridDim.x * 256) {
    const size_t idx = HALF ? (i >> 2) * 8 + (i & 3) : i;
    const uint4 v    = buf[idx];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t N = (size_t)8 << 30;
  uint4 *buf; uint32_t *out;
  CK(hipMalloc(&buf, N)); CK(hipMalloc(&out, 64));
  CK(hipMemset(buf, 1, N)); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, int half, size_t pieces, double useful) {
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0, 0));
      if (half) hipLaunchKernelGGL(stream<1>, dim3(256 * 16), dim3(256), 0, 0, buf, pieces, out);
      else hipLaunchKernelGGL(stream<0>, dim3(256 * 16), dim3(256), 0, 0, buf, pieces, out);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r && ms < best) best = ms;
    }
    printf("%-46s %8.3f ms  %7.2f TB/s of useful bytes\n", name, best, useful / (best * 1e-3) / 1e12);
  };
  timeit("(a) N/2 = 4 GiB contiguous", 0, N / 2 / 16, (double)N / 2);
  timeit("(b) first half of every line of N = 8 GiB", 1, N / 2 / 16, (double)N / 2);
  timeit("(c) N = 8 GiB contiguous", 0, N / 16, (double)N);
  return 0;
}
