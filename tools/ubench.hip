// tools/ubench.hip -- micro-benchmarks that price the instructions the M4RM leaf is made of on
// MI355X (developer tool; results are quoted in DESIGN.md).  Every kernel runs 256 blocks x 512
// threads (1 block per CU, 2 waves per SIMD, the leaf's own geometry) unless noted and reports
// cycles from s_memtime of wave 0 / block 0 plus wall time of the launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o build/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int ITERS = 2000;

__device__ __forceinline__ uint64_t now() { return __builtin_amdgcn_s_memtime(); }

// ---- VALU: 16 independent chains, OP selects the instruction ---------------------------------
template <int OP>
__global__ __launch_bounds__(512) void valu_kernel(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  uint32_t a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = seed * (i + 1) + threadIdx.x;
  uint32_t b = seed ^ 0x9e3779b9u, c = seed + threadIdx.x * 77u;
  __syncthreads();
  const uint64_t t0 = now();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
      if (OP == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(0x0c010500u));
      if (OP == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  const uint64_t t1 = now();
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) x ^= a[i];
  out[blockIdx.x * 512 + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- LDS reads: each wave issues NF ds_read_b128 at pseudo-random 256-byte-row addresses in the
// leaf's conflict-free layout (16 lanes x 16 B per row, 4 rows per instruction), then consumes them
template <int NF, int WIDTH>  // WIDTH: 16 = b128, 8 = b64
__global__ __launch_bounds__(512) void lds_read_kernel(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
  for (int i = threadIdx.x; i < 131072 / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * seed;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t idx   = (threadIdx.x >> 4) * 2654435761u + seed;
  uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  const uint32_t col = (WIDTH == 16) ? (lane & 15) * 16 : (lane & 31) * 8;
  const uint64_t t0 = now();
  for (int it = 0; it < ITERS; ++it) {
    uint4 v[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      idx                 = idx * 1664525u + 1013904223u;
      const uint32_t addr = ((idx >> 8) & 0x1ff00u) | col;  // row = 9 random bits, 256-B rows
      if (WIDTH == 16) v[f] = *reinterpret_cast<const uint4 *>(lds + addr);
      else { const uint2 t = *reinterpret_cast<const uint2 *>(lds + addr); v[f] = make_uint4(t.x, t.y, 0, 0); }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) { acc0 ^= v[f].x; acc1 ^= v[f].y; acc2 ^= v[f].z; acc3 ^= v[f].w; }
    asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
  }
  const uint64_t t1 = now();
  out[blockIdx.x * 512 + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- LDS reads, clean: NF precomputed row addresses per lane (leaf layout), per iteration NF
// ds_read_b128 then the folds.  MODE 0: 4 v_xor per read; 1: one v_bitop3 x4 per PAIR of reads;
// 2: MODE 0 + one v_perm_b32 per read (address generation as in the leaf)
template <int NF, int MODE>
__global__ __launch_bounds__(512) void lds_read2_kernel(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
  for (int i = threadIdx.x; i < 131072 / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * seed;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t idx   = (threadIdx.x >> 4) * 2654435761u + seed;
  const uint32_t col = (lane & 15) * 16;
  uint32_t addr[NF], areg[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    idx     = idx * 1664525u + 1013904223u;
    addr[f] = ((idx >> 8) & 0x1ff00u) | col;
    areg[f] = idx;
  }
  uint32_t acc[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f][0] = acc[f][1] = acc[f][2] = acc[f][3] = 0;
  const uint32_t coloff = col | 0x0100u;
  const uint64_t t0 = now();
  for (int it = 0; it < ITERS; ++it) {
    uint4 v[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      uint32_t a = addr[f];
      if (MODE == 2) a = __builtin_amdgcn_perm(areg[f], coloff, 0x0c010500u | ((it & 1) << 16));
      v[f] = *reinterpret_cast<const uint4 *>(lds + a);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 1) {
#pragma unroll
      for (int f = 0; f < NF; f += 2) {
        acc[f][0] = __builtin_amdgcn_bitop3_b32(acc[f][0], v[f].x, v[f + 1].x, 0x96);
        acc[f][1] = __builtin_amdgcn_bitop3_b32(acc[f][1], v[f].y, v[f + 1].y, 0x96);
        acc[f][2] = __builtin_amdgcn_bitop3_b32(acc[f][2], v[f].z, v[f + 1].z, 0x96);
        acc[f][3] = __builtin_amdgcn_bitop3_b32(acc[f][3], v[f].w, v[f + 1].w, 0x96);
        asm volatile("" : "+v"(acc[f][0]), "+v"(acc[f][1]), "+v"(acc[f][2]), "+v"(acc[f][3]));
      }
    } else {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        acc[f][0] ^= v[f].x; acc[f][1] ^= v[f].y; acc[f][2] ^= v[f].z; acc[f][3] ^= v[f].w;
        asm volatile("" : "+v"(acc[f][0]), "+v"(acc[f][1]), "+v"(acc[f][2]), "+v"(acc[f][3]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const uint64_t t1 = now();
  uint32_t x = 0;
#pragma unroll
  for (int f = 0; f < NF; ++f) x ^= acc[f][0] ^ acc[f][1] ^ acc[f][2] ^ acc[f][3];
  out[blockIdx.x * 512 + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- use-phase replica of the 7-bit kernel: per group 2*UG perm + 2*UG ds_read_b128 + (WR: 4 xor +
// one ds_write_b128) + 4*UG bitop3; barrier every 8 groups.  PIPE: next group's reads issued before the fold.
template <int UG, bool WR, bool PIPE, bool BAR>
__global__ __launch_bounds__(512) void group_replica_kernel(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
  for (int i = threadIdx.x; i < 131072 / 4; i += 512) reinterpret_cast<uint32_t *>(lds)[i] = i * seed;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t col = (lane & 15) * 16, coloff = col | 0x0100u;
  uint32_t areg[8 * UG];
  uint32_t idx = (threadIdx.x >> 4) * 2654435761u + seed;
#pragma unroll
  for (int t = 0; t < 8 * UG; ++t) { idx = idx * 1664525u + 1013904223u; areg[t] = idx; }
  uint32_t acc[8 * UG][4];
#pragma unroll
  for (int t = 0; t < 8 * UG; ++t) acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0;
  uint32_t cur[4] = {seed, seed * 3, seed * 5, seed * 7};
  unsigned char *wr = lds + (threadIdx.x >> 4) * 2048 + col;
  const uint64_t t0 = now();
  for (int it = 0; it < ITERS / 8; ++it) {
    uint4 t0v[PIPE ? 2 : 1][UG], t1v[PIPE ? 2 : 1][UG];
    auto issue = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const uint32_t a0 = __builtin_amdgcn_perm(areg[g * UG + u], coloff, 0x0c0c0400u);
        const uint32_t a1 = __builtin_amdgcn_perm(areg[g * UG + u], coloff, 0x0c0c0500u);
        t0v[slot][u] = *reinterpret_cast<const uint4 *>(lds + a0);
        t1v[slot][u] = *reinterpret_cast<const uint4 *>(lds + a1);
      }
    };
    auto fold = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        uint32_t *a = acc[g * UG + u];
        a[0] = __builtin_amdgcn_bitop3_b32(a[0], t0v[slot][u].x, t1v[slot][u].x, 0x96);
        a[1] = __builtin_amdgcn_bitop3_b32(a[1], t0v[slot][u].y, t1v[slot][u].y, 0x96);
        a[2] = __builtin_amdgcn_bitop3_b32(a[2], t0v[slot][u].z, t1v[slot][u].z, 0x96);
        a[3] = __builtin_amdgcn_bitop3_b32(a[3], t0v[slot][u].w, t1v[slot][u].w, 0x96);
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      }
    };
    if (PIPE) { issue(0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (PIPE) { if (g + 1 < 8) issue(g + 1, (g + 1) & 1); } else issue(g, 0);
      if (WR) {
        cur[0] ^= areg[g]; cur[1] ^= areg[g]; cur[2] ^= areg[g]; cur[3] ^= areg[g];
        asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
        *reinterpret_cast<uint4 *>(wr + 65536 + (g ^ (g >> 1)) * 256) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
      fold(g, PIPE ? (g & 1) : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __syncthreads();
  }
  const uint64_t t1 = now();
  uint32_t x = cur[0];
#pragma unroll
  for (int t = 0; t < 8 * UG; ++t) x ^= acc[t][0] ^ acc[t][1] ^ acc[t][2] ^ acc[t][3];
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = x ^ reinterpret_cast<uint32_t *>(lds)[threadIdx.x + 20000];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- table-build replica: per iteration every thread does the leaf's Gray chain (16 x {4 dependent
// XORs + ds_write_b128 at Gray-ordered rows}) followed by a workgroup barrier
template <int MODE>  // 0: chain + writes + barrier, 1: same without the barrier, 2: writes of constant data (no chain)
__global__ __launch_bounds__(512) void build_replica_kernel(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  uint32_t r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j][0] = seed * (j + 3) + threadIdx.x; r[j][1] = r[j][0] * 3; r[j][2] = r[j][0] * 5; r[j][3] = r[j][0] * 7; }
  uint32_t cur[4] = {seed, seed + 1, seed + 2, seed + 3};
  unsigned char *base = lds + g * 4096 + c * 16;
  __syncthreads();
  const uint64_t t0 = now();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE != 2 && i > 0) {
        const int j = __builtin_ctz(i);
        cur[0] ^= r[j][0]; cur[1] ^= r[j][1]; cur[2] ^= r[j][2]; cur[3] ^= r[j][3];
      }
      asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
      const int gc = i ^ (i >> 1);
      *reinterpret_cast<uint4 *>(base + gc * 256) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
    }
    if (MODE != 1) __syncthreads();
  }
  const uint64_t t1 = now();
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<uint32_t *>(lds)[threadIdx.x] ^ cur[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- LDS writes: NF ds_write per iteration, rows as in the table build ---------------------------
template <int WIDTH>
__global__ __launch_bounds__(512) void lds_write_kernel(uint32_t *out, uint64_t *cyc, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[131072];
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  uint32_t x0 = seed + threadIdx.x, x1 = seed * 3, x2 = seed * 5, x3 = seed * 7;
  unsigned char *base = lds + g * 4096 + c * 16;
  __syncthreads();
  const uint64_t t0 = now();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x0 ^= x1 + i;
      if (WIDTH == 16) *reinterpret_cast<uint4 *>(base + i * 256) = make_uint4(x0, x1, x2, x3);
      if (WIDTH == 8) { *reinterpret_cast<uint2 *>(base + i * 256) = make_uint2(x0, x1); *reinterpret_cast<uint2 *>(base + i * 256 + 8) = make_uint2(x2, x3); }
      asm volatile("" : "+v"(x0));
    }
  }
  const uint64_t t1 = now();
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<uint32_t *>(lds)[threadIdx.x];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// ---- register-table lookup: acc ^= T[idx] with a wave-uniform idx through VGPR index mode --------
// 16-entry table in v[TBASE..TBASE+15]; M0[7:0] selects the entry (s_set_gpr_idx_on ..., SRC0).
template <int NACC>
__global__ __launch_bounds__(512) void gpridx_kernel(uint32_t *out, uint64_t *cyc, const uint32_t *idxs, uint32_t seed) {
  uint32_t r = 0;
  const uint32_t lane_seed = seed + threadIdx.x * 2654435761u;
  uint64_t t0, t1;
  // hand-allocated registers: v[32..47] = table, v[48..48+NACC) = accumulators, s20 = index word
  asm volatile(
      "v_mov_b32 v32, %2\n v_add_u32 v33, 0x11111111, v32\n v_add_u32 v34, 0x11111111, v33\n v_add_u32 v35, 0x11111111, v34\n"
      "v_add_u32 v36, 0x11111111, v35\n v_add_u32 v37, 0x11111111, v36\n v_add_u32 v38, 0x11111111, v37\n v_add_u32 v39, 0x11111111, v38\n"
      "v_add_u32 v40, 0x11111111, v39\n v_add_u32 v41, 0x11111111, v40\n v_add_u32 v42, 0x11111111, v41\n v_add_u32 v43, 0x11111111, v42\n"
      "v_add_u32 v44, 0x11111111, v43\n v_add_u32 v45, 0x11111111, v44\n v_add_u32 v46, 0x11111111, v45\n v_add_u32 v47, 0x11111111, v46\n"
      "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"
      "s_memtime %0\n s_waitcnt lgkmcnt(0)\n"
      "s_mov_b32 s21, %4\n"
      "s_mov_b32 s20, %3\n"
      "s_set_gpr_idx_on s20, gpr_idx(SRC0)\n"
      "1:\n"
      // 8 lookups per loop trip, index = successive nibbles of s20
      "s_bfe_u32 s22, s20, 0x40000\n s_set_gpr_idx_idx s22\n v_xor_b32 v48, v32, v48\n v_xor_b32 v49, v32, v49\n v_xor_b32 v50, v32, v50\n v_xor_b32 v51, v32, v51\n"
      "s_bfe_u32 s22, s20, 0x40004\n s_set_gpr_idx_idx s22\n v_xor_b32 v52, v32, v52\n v_xor_b32 v53, v32, v53\n v_xor_b32 v54, v32, v54\n v_xor_b32 v55, v32, v55\n"
      "s_bfe_u32 s22, s20, 0x40008\n s_set_gpr_idx_idx s22\n v_xor_b32 v48, v32, v48\n v_xor_b32 v49, v32, v49\n v_xor_b32 v50, v32, v50\n v_xor_b32 v51, v32, v51\n"
      "s_bfe_u32 s22, s20, 0x4000c\n s_set_gpr_idx_idx s22\n v_xor_b32 v52, v32, v52\n v_xor_b32 v53, v32, v53\n v_xor_b32 v54, v32, v54\n v_xor_b32 v55, v32, v55\n"
      "s_bfe_u32 s22, s20, 0x40010\n s_set_gpr_idx_idx s22\n v_xor_b32 v48, v32, v48\n v_xor_b32 v49, v32, v49\n v_xor_b32 v50, v32, v50\n v_xor_b32 v51, v32, v51\n"
      "s_bfe_u32 s22, s20, 0x40014\n s_set_gpr_idx_idx s22\n v_xor_b32 v52, v32, v52\n v_xor_b32 v53, v32, v53\n v_xor_b32 v54, v32, v54\n v_xor_b32 v55, v32, v55\n"
      "s_bfe_u32 s22, s20, 0x40018\n s_set_gpr_idx_idx s22\n v_xor_b32 v48, v32, v48\n v_xor_b32 v49, v32, v49\n v_xor_b32 v50, v32, v50\n v_xor_b32 v51, v32, v51\n"
      "s_bfe_u32 s22, s20, 0x4001c\n s_set_gpr_idx_idx s22\n v_xor_b32 v52, v32, v52\n v_xor_b32 v53, v32, v53\n v_xor_b32 v54, v32, v54\n v_xor_b32 v55, v32, v55\n"
      "s_mul_i32 s20, s20, 0x19660d\n s_add_u32 s20, s20, 0x3c6ef35f\n"
      "s_sub_u32 s21, s21, 1\n s_cmp_lg_u32 s21, 0\n s_cbranch_scc1 1b\n"
      "s_set_gpr_idx_off\n"
      "s_memtime %1\n s_waitcnt lgkmcnt(0)\n"
      "v_xor_b32 v48, v48, v52\n"
      : "=&s"(t0), "=&s"(t1)
      : "v"(lane_seed), "s"(seed), "s"(ITERS)
      : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
        "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s20", "s21", "s22", "scc", "memory");
  asm volatile("v_mov_b32 %0, v48" : "=v"(r));
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
static void run(const char *name, F launch, double work_per_block_iter, const char *unit, uint32_t *dout, uint64_t *dcyc) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  launch();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  uint64_t cyc; CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
  // s_memtime ticks at a constant 100 MHz on this part, so derive rates from wall time of the launch
  const double per_cu_per_us = work_per_block_iter * ITERS / (ms * 1e3);
  printf("%-44s %8.3f ms  memtime %8llu  -> %10.1f %s/CU/us ; %8.2f %s per memtime-tick per CU\n", name, ms,
         (unsigned long long)cyc, per_cu_per_us, unit, work_per_block_iter * ITERS / (double)cyc, unit);
}

int main(int argc, char **argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int which = argc > 1 ? atoi(argv[1]) : -1;  // -1: everything except the gpr-idx kernel
  uint32_t *dout; uint64_t *dcyc;
  CK(hipMalloc(&dout, 256 * 512 * 4)); CK(hipMalloc(&dcyc, 64));
  const dim3 g(256), b(512);
  // VALU: 8 waves x 16 instr per iteration per block
  run("v_xor_b32", [&] { hipLaunchKernelGGL(valu_kernel<0>, g, b, 0, 0, dout, dcyc, 12345u); }, 8 * 16, "wave-instr", dout, dcyc);
  run("v_bitop3_b32", [&] { hipLaunchKernelGGL(valu_kernel<1>, g, b, 0, 0, dout, dcyc, 12345u); }, 8 * 16, "wave-instr", dout, dcyc);
  run("v_perm_b32", [&] { hipLaunchKernelGGL(valu_kernel<2>, g, b, 0, 0, dout, dcyc, 12345u); }, 8 * 16, "wave-instr", dout, dcyc);
  run("v_and_or_b32", [&] { hipLaunchKernelGGL(valu_kernel<3>, g, b, 0, 0, dout, dcyc, 12345u); }, 8 * 16, "wave-instr", dout, dcyc);
  run("v_lshl_add_u32", [&] { hipLaunchKernelGGL(valu_kernel<4>, g, b, 0, 0, dout, dcyc, 12345u); }, 8 * 16, "wave-instr", dout, dcyc);
  // LDS reads: bytes per block-iteration = 8 waves x NF x 64 lanes x WIDTH
  run("ds_read_b128 x4 in flight", [&] { hipLaunchKernelGGL((lds_read_kernel<4, 16>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 4 * 1024, "bytes", dout, dcyc);
  run("ds_read_b128 x8 in flight", [&] { hipLaunchKernelGGL((lds_read_kernel<8, 16>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 8 * 1024, "bytes", dout, dcyc);
  run("ds_read_b128 x16 in flight", [&] { hipLaunchKernelGGL((lds_read_kernel<16, 16>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("ds_read_b64 x8 in flight", [&] { hipLaunchKernelGGL((lds_read_kernel<8, 8>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 8 * 512, "bytes", dout, dcyc);
  run("ds_read_b64 x16 in flight", [&] { hipLaunchKernelGGL((lds_read_kernel<16, 8>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 512, "bytes", dout, dcyc);
  run("clean ds_read_b128 x2  + 4 xor each", [&] { hipLaunchKernelGGL((lds_read2_kernel<2, 0>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 2 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x4  + 4 xor each", [&] { hipLaunchKernelGGL((lds_read2_kernel<4, 0>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 4 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x8  + 4 xor each", [&] { hipLaunchKernelGGL((lds_read2_kernel<8, 0>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 8 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x16 + 4 xor each", [&] { hipLaunchKernelGGL((lds_read2_kernel<16, 0>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x4  + bitop3 pairs", [&] { hipLaunchKernelGGL((lds_read2_kernel<4, 1>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 4 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x8  + bitop3 pairs", [&] { hipLaunchKernelGGL((lds_read2_kernel<8, 1>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 8 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x16 + bitop3 pairs", [&] { hipLaunchKernelGGL((lds_read2_kernel<16, 1>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x4  + perm + 4 xor", [&] { hipLaunchKernelGGL((lds_read2_kernel<4, 2>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 4 * 1024, "bytes", dout, dcyc);
  run("clean ds_read_b128 x8  + perm + 4 xor", [&] { hipLaunchKernelGGL((lds_read2_kernel<8, 2>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 8 * 1024, "bytes", dout, dcyc);
  // bytes read per block-iteration (ITERS/8 outer trips x 8 groups): 8 waves x 2*UG reads x 1 KiB per group
#define GR(UGV, WRV, PV, BV, label) run(label, [&] { hipLaunchKernelGGL((group_replica_kernel<UGV, WRV, PV, BV>), g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 2 * UGV * 1024, "read-bytes", dout, dcyc)
  GR(4, false, false, false, "group replica UG4 reads only");
  GR(4, true, false, false, "group replica UG4 +write");
  GR(4, true, false, true, "group replica UG4 +write +barrier/8");
  GR(2, true, false, true, "group replica UG2 +write +barrier/8");
  GR(2, true, true, true, "group replica UG2 +write +barrier PIPE");
  GR(4, true, true, true, "group replica UG4 +write +barrier PIPE");
  GR(4, false, true, false, "group replica UG4 reads only PIPE");
  run("build replica: chain+write+barrier", [&] { hipLaunchKernelGGL(build_replica_kernel<0>, g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("build replica: chain+write, no barrier", [&] { hipLaunchKernelGGL(build_replica_kernel<1>, g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("build replica: const data + barrier", [&] { hipLaunchKernelGGL(build_replica_kernel<2>, g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("ds_write_b128 (16 per wave per iter)", [&] { hipLaunchKernelGGL(lds_write_kernel<16>, g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  run("ds_write_b64 x2 (16 pairs per wave per iter)", [&] { hipLaunchKernelGGL(lds_write_kernel<8>, g, b, 0, 0, dout, dcyc, 7u); }, 8.0 * 16 * 1024, "bytes", dout, dcyc);
  // register-table lookups: 8 lookups x 4 v_xor per loop trip per wave
  if (which != 99) return 0;
  run("gpr-idx table lookup (4 xor per idx)", [&] { hipLaunchKernelGGL(gpridx_kernel<8>, g, b, 0, 0, dout, dcyc, nullptr, 99u); }, 8.0 * 32, "wave-xor", dout, dcyc);
  // correctness of the gpr-idx lookup: recompute lane 0 of block 0 on the host
  {
    std::vector<uint32_t> h(512);
    CK(hipMemcpy(h.data(), dout, 512 * 4, hipMemcpyDeviceToHost));
    uint32_t bad = 0;
    for (int t = 0; t < 512; t += 37) {
      uint32_t T[16]; T[0] = 99u + t * 2654435761u; for (int i = 1; i < 16; ++i) T[i] = T[i - 1] + 0x11111111u;
      uint32_t acc[8] = {0}; uint32_t s = 99u;
      for (int it = 0; it < ITERS; ++it) {
        for (int q = 0; q < 8; ++q) { const uint32_t v = T[(s >> (4 * q)) & 15]; for (int j = 0; j < 4; ++j) acc[(q & 1) * 4 + j] ^= v; }
        s = s * 0x19660du + 0x3c6ef35fu;
      }
      uint32_t x = acc[0] ^ acc[4];
      if (x != h[t]) ++bad;
    }
    printf("gpr-idx lookup correctness: %s\n", bad ? "MISMATCH" : "ok");
  }
  return 0;
}
