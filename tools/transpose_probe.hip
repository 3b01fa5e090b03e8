// tools/transpose_probe.hip -- the timeline of the transpose kernel's workgroups (m4ri_amd/csrc/transpose.hip built with
// -DTR_TIMING): when a block started, had its loads issued, saw its first group arrive, was half way, finished its groups,
// had its stores issued.  build: hipcc --offload-arch=gfx950 -O3 -DTR_TIMING -o build/transpose_probe tools/transpose_probe.hip
// (without -DTR_TIMING: the product kernel, timed only -- what tools/prof_transpose.sh runs under rocprofv3)
#pragma clang diagnostic ignored "-Wunused-value"
#include "../m4ri_amd/csrc/transpose.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
  const int64_t s = n / 64;
  word *A, *D;
  hipMalloc((void **)&A, n * s * 8); hipMalloc((void **)&D, n * s * 8);
  hipMemset(A, 0x5a, n * s * 8);
  for (int it = 0; it < 3; ++it) m4ri_amd_transpose_dev(D, s, A, s, n, n, nullptr);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, nullptr);
  m4ri_amd_transpose_dev(D, s, A, s, n, n, nullptr);
  hipEventRecord(e1, nullptr); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
#ifndef TR_TIMING
  printf("n = %lld: %.3f ms  %.3f TB/s\n", (long long)n, ms, 2.0 * n * s * 8 / ms / 1e9);
  return 0;
#else
  static unsigned long long t[8192][6];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(tr_times), sizeof(t));
  const int64_t tiles = (n / 1024) * (n / 1024), nb = tiles < 8192 ? tiles : 8192;
  unsigned long long t0 = ~0ull, tend = 0;
  for (int64_t b = 0; b < nb; ++b) { t0 = std::min(t0, t[b][0]); tend = std::max(tend, t[b][5]); }
  printf("n = %lld: %.3f ms; first start .. last end: %.1f us\n", (long long)n, ms, (tend - t0) / 100.0);
  const char *names[5] = {"loads issued", "first group arrived", "group 8", "groups done", "stores issued"};
  for (int k = 0; k < 5; ++k) {
    std::vector<double> d;
    for (int64_t b = 0; b < nb; ++b) d.push_back((t[b][k + 1] - t[b][k]) / 100.0);
    std::sort(d.begin(), d.end());
    printf("  %-20s median %7.2f us   p10 %7.2f   p90 %7.2f\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
  }
  std::vector<double> st, life;
  for (int64_t b = 0; b < nb; ++b) { st.push_back((t[b][0] - t0) / 100.0); life.push_back((t[b][5] - t[b][0]) / 100.0); }
  std::sort(st.begin(), st.end()); std::sort(life.begin(), life.end());
  printf("  block start times: p10 %.1f  median %.1f  p90 %.1f us; lifetime median %.2f us\n", st[st.size() / 10], st[st.size() / 2], st[st.size() * 9 / 10], life[life.size() / 2]);
  return 0;
#endif
}
