// m4rm_leaf_db.hip -- M4RM leaf, double-buffered tables (the product's leaf kernel).
//
// Same result as m4rm_leaf.hip's two-phase kernel and the same lane layout (C-stationary tile of
// 32*RG rows x 2048 columns in VGPRs; 256-byte table entries; four rows per ds_read_b128, bank-
// conflict free; v_perm_b32 address generation), but the inner loop never drains the LDS pipe:
//
//   * a stage is 8 inner bits = ONE 256-entry table (64 KiB); two table buffers live in LDS;
//   * while all 8 waves gather from table s (buffer s&1), four of them also build table s+1 into the
//     other buffer -- the two halves of the workgroup take turns, so every wave runs
//     "build, use, use" over two stages and ONE barrier per stage replaces two drain+barrier pairs;
//   * the measured cost of the two-phase kernel was 4790 clk per 16 bits against 3072 clk of LDS-
//     array work (SQ_LDS_IDX_ACTIVE): the difference was pipe drain/refill around the barriers.
//
// Replaces (result-identical) _mzd_mul_m4rm, mzd_make_table and _mzd_combine_N of the reference
// (/root/reference m4ri/brilliantrussian.c:1032-1190, :163-211, m4ri/xor_template.h:12-227).
#include <hip/hip_runtime.h>
#include "gf2_common.h"

namespace {

__device__ __forceinline__ int64_t words_of_dev(int64_t ncols) { return (ncols + 63) >> 6; }

// Raw buffer descriptor from wave-uniform inputs (readfirstlane makes the uniformity provable to
// hipcc; otherwise it may wrap every buffer_load in a waterfall loop, cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
  const uint64_t b  = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane(bytes);
  void *p           = reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)nb, 0x00020000);
}

// v_perm_b32(a, coloff, sel): (byte j of a) << 8 | coloff.byte0 | buffer << 16, coloff.byte1 == 0x01
__device__ __forceinline__ constexpr uint32_t perm_sel(int j, int buf) {
  return 0x0c000000u | ((buf ? 0x01u : 0x0cu) << 16) | ((uint32_t)(4 + j) << 8) | 0x00u;
}

template <int RG, int UG, bool XOR_OUT>
struct LeafDB {
  static constexpr int R  = 32 * RG;
  static constexpr int NG = RG / UG;
  static_assert(RG % UG == 0, "RG must be a multiple of UG");

  // per-thread state shared by the two role variants of the main loop
  uint32_t acc[RG][4];
  uint32_t areg[RG];
  uint4 brow[8];
  unsigned char *lds;
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
  uint32_t a_rs, b_rs, a_lane, b_lane, coloff;
  int bhi, c;

  __device__ __forceinline__ void load_b(int stage) {
    // 8 rows x 16 B of B for the table this thread helps to build; one running offset VGPR (the
    // empty asm keeps hipcc from materialising 8 hoisted offsets).  Rows >= l read as 0 through the
    // descriptor's range check.
    uint32_t off = b_lane + (uint32_t)stage * 8u * b_rs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      brow[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
      off += b_rs;
      asm volatile("" : "+v"(off));
    }
  }

  __device__ __forceinline__ void load_a_all(int q) {
    uint32_t off = a_lane + (uint32_t)q * 4u;
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      areg[t] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, (int)off, 0, 0);
      off += a_rs;
      asm volatile("" : "+v"(off));
    }
  }

  // 16 entries (high nibble bhi, Gray order over the low nibble) of the table in buffer `buf`
  __device__ __forceinline__ void build(int buf) {
    // the B rows become visible to the optimiser only HERE (volatile asm stays behind the previous
    // barrier): otherwise hipcc hoists the first XORs up to where the rows were requested and
    // waits out the whole load latency there.
#pragma unroll
    for (int j = 0; j < 8; ++j)
      asm volatile("" : "+v"(brow[j].x), "+v"(brow[j].y), "+v"(brow[j].z), "+v"(brow[j].w));
    unsigned char *wr = lds + buf * 65536 + bhi * 16 * 256 + c * 16;
    uint32_t cur[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool on = (bhi >> j) & 1;
      cur[0] ^= on ? brow[4 + j].x : 0u;
      cur[1] ^= on ? brow[4 + j].y : 0u;
      cur[2] ^= on ? brow[4 + j].z : 0u;
      cur[3] ^= on ? brow[4 + j].w : 0u;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i > 0) {
        const int j = __builtin_ctz(i);
        cur[0] ^= brow[j].x;
        cur[1] ^= brow[j].y;
        cur[2] ^= brow[j].z;
        cur[3] ^= brow[j].w;
      }
      // keep the Gray chain a chain (one XOR + one ds_write_b128 per entry)
      asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
      const int gcode = i ^ (i >> 1);
      *reinterpret_cast<uint4 *>(wr + gcode * 256) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
    }
  }

  // gather from the table of stage byte J (buffer J&1) for all RG rows; RELOAD: refill areg with
  // the next 32-bit chunk of A as soon as a row's last index has been extracted
  template <int J, bool RELOAD>
  __device__ __forceinline__ void use(int q_next) {
    uint32_t aoff = a_lane + (uint32_t)q_next * 4u;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      uint4 t[UG];
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const uint32_t ad = __builtin_amdgcn_perm(areg[g * UG + u], coloff, perm_sel(J, J & 1));
        t[u]              = *reinterpret_cast<const uint4 *>(lds + ad);
        if constexpr (RELOAD) {
          areg[g * UG + u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, (int)aoff, 0, 0);
          aoff += a_rs;
          asm volatile("" : "+v"(aoff));
        }
      }
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        uint32_t *a = acc[g * UG + u];
        a[0] ^= t[u].x;
        a[1] ^= t[u].y;
        a[2] ^= t[u].z;
        a[3] ^= t[u].w;
        // pin the accumulation here (XOR is associative: un-pinned, hipcc re-associates the whole
        // loop into one late XOR tree and keeps every loaded table row live)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  __device__ __forceinline__ void stage_end() {
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }

  // main loop for the waves of builder group GRP (0: builds the even-numbered tables, 1: the odd)
  template <int GRP>
  __device__ __forceinline__ void run(int q_begin, int q_end) {
    // prologue: table 4*q_begin (even) is built by group 0 before anyone gathers
    load_a_all(q_begin);
    load_b(4 * q_begin + GRP);
    if constexpr (GRP == 0) {
      build(0);
      load_b(4 * q_begin + 2);
    }
    stage_end();
    for (int q = q_begin; q < q_end; ++q) {
      const int s = 4 * q;
      // stage s (table in buffer 0); group 1 builds table s+1 into buffer 1
      if constexpr (GRP == 1) { build(1); load_b(s + 3); }
      use<0, false>(0);
      stage_end();
      // stage s+1 (buffer 1); group 0 builds table s+2 into buffer 0
      if constexpr (GRP == 0) { build(0); load_b(s + 4); }
      use<1, false>(0);
      stage_end();
      // stage s+2 (buffer 0); group 1 builds table s+3
      if constexpr (GRP == 1) { build(1); load_b(s + 5); }
      use<2, false>(0);
      stage_end();
      // stage s+3 (buffer 1); group 0 builds table s+4 = first table of the next chunk
      if constexpr (GRP == 0) { build(0); load_b(s + 6); }
      use<3, true>(q + 1);
      stage_end();
    }
  }
};

template <int RG, int UG, bool XOR_OUT>
__global__ __launch_bounds__(LEAF_THREADS) void m4rm_leaf_db_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 65536];
  using K = LeafDB<RG, UG, XOR_OUT>;
  constexpr int R = K::R;

  const int tid  = threadIdx.x;
  const int c    = tid & 15;  // 16-byte column slot of the 256-byte table entry
  const int rgrp = tid >> 4;  // row group 0..31 (use) ; tid>>8 = builder group, (tid>>4)&15 = high nibble

  // block -> (batch, tile_n, ksplit, tile_m); consecutive logical ids share a B panel, and the XCD
  // remap keeps them on one XCD's L2 (blocks are dispatched round-robin over 8 XCDs)
  uint32_t lid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x;
    if ((nwg & 7u) == 0u) lid = (lid & 7u) * (nwg >> 3) + (lid >> 3);
  }
  const int tile_m = lid % p.tiles_m; lid /= p.tiles_m;
  const int ks     = lid % p.ksplit;  lid /= p.ksplit;
  const int tile_n = lid % p.tiles_n; lid /= p.tiles_n;
  const int64_t bat = lid;

  const word *Ab = p.A + bat * p.a_bs;
  const word *Bb = p.B + bat * p.b_bs;
  word *__restrict__ Cb = p.C + bat * p.c_bs;

  const int w0   = tile_n * LEAF_TW + c * 2;  // this lane's two words of the row
  const bool v0  = w0 < p.wn;
  const bool v1  = (w0 + 1) < p.wn;
  const int row0 = tile_m * R + rgrp * RG;

  K k;
  k.lds = lds;
  // A and B are read through raw buffer descriptors: every access is one per-lane 32-bit byte
  // offset from a wave-uniform base, and the hardware range check returns 0 for rows >= m of A and
  // rows >= l of B -- exactly the zero padding the algorithm wants, so no edge branches.
  k.a_rsrc = make_rsrc(Ab, (uint32_t)(((int64_t)(p.m - 1) * p.a_stride + words_of_dev(p.l)) * 8));
  k.b_rsrc = make_rsrc(Bb, (uint32_t)(((int64_t)(p.l - 1) * p.b_stride + p.wn) * 8));
  k.a_rs   = (uint32_t)p.a_stride * 8u;
  k.b_rs   = (uint32_t)p.b_stride * 8u;
  k.a_lane = (uint32_t)row0 * k.a_rs;
  k.b_lane = (uint32_t)w0 * 8u;
  k.coloff = (uint32_t)(c * 16) | 0x0100u;  // byte0 = column offset, byte1 = 0x01 (buffer select)
  k.bhi    = rgrp & 15;
  k.c      = c;
#pragma unroll
  for (int t = 0; t < RG; ++t) { k.acc[t][0] = k.acc[t][1] = k.acc[t][2] = k.acc[t][3] = 0u; }

  const int total_chunks = (p.l + 31) / 32;
  const int q_begin      = ks * p.chunks_per_split;
  int q_end              = q_begin + p.chunks_per_split;
  if (q_end > total_chunks) q_end = total_chunks;

  // the two halves of the workgroup run role-specialised copies of the loop; the selector is made
  // wave-uniform for the compiler (scalar branch, no exec masking around the barriers)
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
  if (q_begin < q_end) {
    if (grp == 0) k.template run<0>(q_begin, q_end);
    else k.template run<1>(q_begin, q_end);
  }

  // epilogue: C tile out.  One running row pointer (pinned, so hipcc cannot hoist RG 64-bit row
  // addresses above the main loop); the column guards are loop-invariant per lane.
  if (v0) {
    word *cp       = Cb + (int64_t)row0 * p.c_stride + w0;
    const int rows = (p.m - row0) < RG ? (p.m - row0) : RG;  // may be <= 0
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      if (t < rows) {
        const word x0 = (word)k.acc[t][0] | ((word)k.acc[t][1] << 32);
        const word x1 = (word)k.acc[t][2] | ((word)k.acc[t][3] << 32);
        if constexpr (!XOR_OUT) {
          cp[0] = x0;
          if (v1) cp[1] = x1;
        } else {
          // C ^= tile: a no-return L2 atomic needs no destination registers and is what makes
          // inner-dimension splits (ksplit > 1) race-free; XOR is exact, so order is moot
          atomicXor(reinterpret_cast<unsigned long long *>(cp), (unsigned long long)x0);
          if (v1) atomicXor(reinterpret_cast<unsigned long long *>(cp + 1), (unsigned long long)x1);
        }
      }
      cp += p.c_stride;
      asm volatile("" : "+v"(cp));
    }
  }
}

}  // namespace

// (rg, ug) instantiations
#define LEAF_DB_VARIANTS(X) X(32, 4) X(32, 2) X(32, 8) X(24, 4) X(24, 8) X(16, 4) X(16, 8)

extern "C" hipError_t gf2_launch_m4rm_leaf_db(hipStream_t stream, LeafArgs a, int rg, int ug) {
  const int R = 32 * rg;
  a.wn        = (int32_t)words_of(a.n);
  a.tiles_m   = (a.m + R - 1) / R;
  a.tiles_n   = (a.wn + LEAF_TW - 1) / LEAF_TW;
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0) return hipSuccess;
  const int total_chunks = (a.l + 31) / 32;
  if (a.ksplit < 1) a.ksplit = 1;
  int cps = (total_chunks + a.ksplit - 1) / a.ksplit;
  if (cps < 1) cps = 1;
  a.chunks_per_split = cps;
  a.ksplit           = total_chunks > 0 ? (total_chunks + cps - 1) / cps : 1;
  if (a.ksplit < 1) a.ksplit = 1;
  if (a.ksplit > 1 && a.mode == 0) return hipErrorInvalidValue;  // caller must pre-zero C and pass mode 1
  const long long nwg = (long long)a.tiles_m * a.tiles_n * a.ksplit * a.batch;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nwg), block(LEAF_THREADS);
  if (ug == 0) ug = 4;
#define LEAF_CASE(RGV, UGV)                                                                          \
  if (rg == RGV && ug == UGV) {                                                                      \
    if (a.mode == 0) hipLaunchKernelGGL((m4rm_leaf_db_kernel<RGV, UGV, false>), grid, block, 0, stream, a); \
    else             hipLaunchKernelGGL((m4rm_leaf_db_kernel<RGV, UGV, true>), grid, block, 0, stream, a);  \
    return hipGetLastError();                                                                        \
  }
  LEAF_DB_VARIANTS(LEAF_CASE)
#undef LEAF_CASE
  return hipErrorInvalidValue;
}
