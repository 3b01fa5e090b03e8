// m4rm_leaf.hip -- M4RM leaf, generation 1 (two-phase, k = 8): C (^)= A*B over GF(2) by the Method of
// Four Russians, hand-written for gfx950 (MI355X / CDNA4).  No MFMA: this is a lookup + XOR path.
// The engine uses it for tiles shorter than 1024 rows and as the fallback that needs no packed A;
// everything from 192 rows on runs generation 4 (m4rm8q_leaf.hip).
//
// Replaces (result-identical, not structure-identical) the reference's leaf
//   _mzd_mul_m4rm           /root/reference m4ri/brilliantrussian.c:1032-1190
//   mzd_make_table          m4ri/brilliantrussian.c:163-211
//   _mzd_combine_8          m4ri/xor_template.h:12-227
//
// MI355X-first design (see DESIGN.md "leaf kernel"):
//   * C-stationary: one 512-thread workgroup owns a (32*RG rows) x 2048-column tile of C and keeps
//     it in VGPRs (RG*4 dwords per lane) for the whole inner loop -- C is read/written once.
//   * Per 16-bit stage of the inner dimension the workgroup builds two 256-entry tables
//     T_z[x] = XOR_{b in x} B[16s + 8z + b, tile columns] straight into LDS (2 x 64 KiB, 256 B
//     per entry), Gray-code order so every entry costs one XOR + one ds_write_b128.
//   * Use phase: a wave reads FOUR table rows per ds_read_b128 -- lane = (row-group g = lane>>4,
//     16-byte column slot c = lane&15).  With c = lane&15 every one of the instruction's four
//     16-lane service groups touches 16 distinct slots of the 256-byte bank row, so the gather is
//     bank-conflict free for ANY combination of table indices (MI355X_MICROARCH.md, LDS table).
//   * The LDS byte address of a lookup is produced by ONE v_perm_b32: it drops byte j of the
//     lane's A chunk into bits 8..15 (= index*256) next to the lane's column offset, and selects
//     the table (bit 16) from a constant byte.  Two lookups are folded into the accumulator by one
//     v_bitop3_b32 (3-input XOR, gfx950) per dword.
//   * A and B stream from L2/HBM through plain global loads issued one phase ahead (VMEM pipe is
//     otherwise idle); the LDS pipe is the binding resource by design.
#include <hip/hip_runtime.h>
#include "gf2_common.h"

// (rg, ug) instantiations.  Two variants of this kernel were measured and removed again: software-
// pipelined gathers (two register sets; +-1 %) and B rows staged once per workgroup through LDS
// (-3 %); DESIGN.md 3.1 has the table.
#define LEAF_VARIANTS(X) X(32, 4) X(24, 4) X(16, 4)
#define LEAF_DEFAULT_UG(rg) 4

namespace {

__device__ __forceinline__ int64_t words_of_dev(int64_t ncols) { return (ncols + 63) >> 6; }

// Raw buffer descriptor from wave-uniform inputs.  The readfirstlanes make the uniformity provable
// to hipcc; without them it may park descriptor words in VGPRs and wrap every buffer_load in a
// waterfall loop (cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes) {
  const uint64_t b  = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint32_t nb = __builtin_amdgcn_readfirstlane(bytes);
  void *p           = reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)nb, 0x00020000);
}

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// selector for v_perm_b32(a, coloff, sel): result = (byte j of a) << 8 | coloff.byte0 | table << 16
__device__ __forceinline__ constexpr uint32_t perm_sel(int j, int z) {
  return 0x0c000000u | ((z ? 0x01u : 0x0cu) << 16) | ((uint32_t)(4 + j) << 8) | 0x00u;
}

template <int RG, int UG, bool XOR_OUT>
__global__ __launch_bounds__(LEAF_THREADS) void m4rm_leaf_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LEAF_NT * 65536];
  constexpr int R = 32 * RG;  // tile rows: 32 row groups (8 waves x 4) x RG rows

  const int tid  = threadIdx.x;
  const int c    = tid & 15;   // 16-byte column slot of the 256-byte table entry
  const int rgrp = tid >> 4;   // use phase: row group 0..31;  build phase: (table, high nibble)

  // ---- block -> (batch, tile_n, ksplit, tile_m); consecutive logical ids share a B panel, and
  // the XCD remap keeps them on one XCD's L2 (blocks are dispatched round-robin over 8 XCDs).
  uint32_t lid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x;
    if ((nwg & 7u) == 0u) lid = (lid & 7u) * (nwg >> 3) + (lid >> 3);
  }
  const int tile_m = lid % p.tiles_m; lid /= p.tiles_m;
  const int ks     = lid % p.ksplit;  lid /= p.ksplit;
  const int tile_n = lid % p.tiles_n; lid /= p.tiles_n;
  const int64_t bat = lid;

  // A and B are read through raw buffer descriptors: every access is one per-lane 32-bit byte
  // offset from a wave-uniform base (no 64-bit address VGPRs live across the main loop), and the
  // hardware range check returns 0 for rows >= m of A and rows >= l of B -- which is exactly the
  // zero padding the algorithm wants, so the main loop carries no edge branches.
  const word *Ab = p.A + bat * p.a_bs;
  const word *Bb = p.B + bat * p.b_bs;
  word *__restrict__ Cb = p.C + bat * p.c_bs;
  const uint32_t a_bytes = (uint32_t)(((int64_t)(p.m - 1) * p.a_stride + words_of_dev(p.l)) * 8);
  const uint32_t b_bytes = (uint32_t)(((int64_t)(p.l - 1) * p.b_stride + p.wn) * 8);
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(Ab, a_bytes);
  const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(Bb, b_bytes);

  const int w0     = tile_n * LEAF_TW + c * 2;  // this lane's two words of the row
  const bool v0    = w0 < p.wn;
  const bool v1    = (w0 + 1) < p.wn;
  const int row0   = tile_m * R + rgrp * RG;
  const uint32_t a_rs = (uint32_t)p.a_stride * 8u;  // row strides in bytes
  const uint32_t b_rs = (uint32_t)p.b_stride * 8u;

  // build-phase role
  const int bz  = rgrp >> 4;  // table 0/1
  const int bhi = rgrp & 15;  // high nibble of the entries this thread writes
  unsigned char *const tbl_wr = lds + bz * 65536 + bhi * 16 * 256 + c * 16;
  const uint32_t coloff = (uint32_t)(c * 16) | 0x0100u;  // byte0 = column offset, byte1 = 0x01

  uint32_t acc[RG][4];
#pragma unroll
  for (int t = 0; t < RG; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0u; }

  const int total_stages = (p.l + LEAF_STAGE - 1) / LEAF_STAGE;
  const int s_begin      = ks * p.stages_per_split;          // even by construction
  int s_end              = s_begin + p.stages_per_split;
  if (s_end > total_stages) s_end = total_stages;

  // B rows of one stage for this thread's table: 8 rows x 16 bytes (columns outside the matrix
  // may hold a neighbour's bits when B is a window; they only reach C columns that are never
  // stored).
  uint4 brow[8];
  const uint32_t b_lane = (uint32_t)bz * 8u * b_rs + (uint32_t)w0 * 8u;
  auto load_b = [&](int s) {  // every thread fetches its 8 rows itself (16x redundant, L1/L2 hits)
    // one running offset VGPR (the empty asm keeps hipcc from materialising 8 hoisted offsets)
    uint32_t off = b_lane + (uint32_t)s * LEAF_STAGE * b_rs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      brow[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
      off += b_rs;
      asm volatile("" : "+v"(off));
    }
  };
  uint32_t areg[RG];
  const uint32_t a_lane = (uint32_t)row0 * a_rs;
  auto load_a = [&](int q) {
    uint32_t off = a_lane + (uint32_t)q * 4u;
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      areg[t] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(a_rsrc, (int)off, 0, 0);
      off += a_rs;
      asm volatile("" : "+v"(off));
    }
  };

  if (s_begin < s_end) load_b(s_begin);

  for (int q = s_begin >> 1; 2 * q < s_end; ++q) {
    load_a(q);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int s = 2 * q + half;
      // ---------------- build: 16 entries of table bz, high nibble bhi ----------------------
      {
        // the B rows become visible to the optimiser only HERE (volatile asm stays behind the
        // preceding barrier): otherwise hipcc hoists this build's first XORs up to where the rows
        // were requested, one use phase earlier, and waits out the whole load latency there.
#pragma unroll
        for (int j = 0; j < 8; ++j)
          asm volatile("" : "+v"(brow[j].x), "+v"(brow[j].y), "+v"(brow[j].z), "+v"(brow[j].w));
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
          // keep the Gray chain a chain (one XOR + one ds_write_b128 per entry); un-pinned, hipcc
          // re-associates the 16 entries into independent trees and holds all 64 dwords at once.
          asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
          const int gcode = i ^ (i >> 1);
          *reinterpret_cast<uint4 *>(tbl_wr + gcode * 256) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
        }
      }
      // next stage's B rows: requested now, consumed by the next build, one use phase later.
      // (Measured: issuing them inside the use phase instead is SLOWER -- the use phase is bound by
      // each wave's serial issue, and 8 x buffer_load_dwordx4 cost a wave ~270 clk of issue time.)
      // Unconditional on purpose (rows past the end read as 0 through the descriptor's range
      // check): a branch makes hipcc resolve the merge with v_movs of the loaded registers, i.e.
      // an immediate vmcnt(0).  The sched_barriers keep the next build's XORs from being hoisted up
      // to the loads.
      load_b(s + 1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- use: RG rows x 2 lookups ------------------------------------------
      // Rows go through in groups of UG: 2*UG ds_read_b128 per group, then their XORs.  The
      // sched_barriers pin that shape: un-pinned, hipcc hoists every read of the phase above the
      // XORs and spills the tile.
      static_assert(RG % UG == 0, "RG must be a multiple of UG");
      constexpr int NG = RG / UG;
      uint4 t0[UG], t1[UG];
      auto issue = [&](int g) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
          const uint32_t a0 = __builtin_amdgcn_perm(areg[g * UG + u], coloff, perm_sel(2 * half + 0, 0));
          const uint32_t a1 = __builtin_amdgcn_perm(areg[g * UG + u], coloff, perm_sel(2 * half + 1, 1));
          t0[u]             = *reinterpret_cast<const uint4 *>(lds + a0);
          t1[u]             = *reinterpret_cast<const uint4 *>(lds + a1);
        }
      };
      auto fold = [&](int g) {
#pragma unroll
        for (int u = 0; u < UG; ++u) {
          uint32_t *a = acc[g * UG + u];
          a[0] = xor3(a[0], t0[u].x, t1[u].x);
          a[1] = xor3(a[1], t0[u].y, t1[u].y);
          a[2] = xor3(a[2], t0[u].z, t1[u].z);
          a[3] = xor3(a[3], t0[u].w, t1[u].w);
          // pin the accumulation here: XOR is associative, and without this hipcc re-associates
          // the whole phase into one late XOR tree and keeps every loaded table row live.
          asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        }
      };
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        issue(g);
        fold(g);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---------------- epilogue: C tile out -----------------------------------------------------
  // One running row pointer (pinned, so hipcc cannot hoist RG 64-bit row addresses above the main
  // loop -- that alone cost >60 VGPRs); the column guards are loop-invariant per lane.
  if (v0) {
    word *cp       = Cb + (int64_t)row0 * p.c_stride + w0;
    const int rows = (p.m - row0) < RG ? (p.m - row0) : RG;  // may be <= 0
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      if (t < rows) {
        const word x0 = (word)acc[t][0] | ((word)acc[t][1] << 32);
        const word x1 = (word)acc[t][2] | ((word)acc[t][3] << 32);
        if constexpr (!XOR_OUT) {
          cp[0] = x0;
          if (v1) cp[1] = x1;
        } else {
          // C ^= tile.  A no-return L2 atomic needs no destination registers and is also what
          // makes inner-dimension splits (ksplit > 1) race-free; XOR is exact, so order is moot.
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

// Host launcher.  rg: tile height / 32 (rows = 32*rg); ug: rows per read group, 0 picks the tuned
// default for rg.  (`pipe` is what is left of the removed variants' selector: must be 0.)
extern "C" hipError_t gf2_launch_m4rm_leaf_variant(hipStream_t stream, LeafArgs a, int rg, int ug, int pipe) {
  const int R = 32 * rg;
  a.wn        = (int32_t)words_of(a.n);
  a.tiles_m   = (a.m + R - 1) / R;
  a.tiles_n   = (a.wn + LEAF_TW - 1) / LEAF_TW;
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0) return hipSuccess;
  const int total_stages = (a.l + LEAF_STAGE - 1) / LEAF_STAGE;
  if (a.ksplit < 1) a.ksplit = 1;
  int sps = (total_stages + a.ksplit - 1) / a.ksplit;
  sps     = (sps + 1) & ~1;  // even: a split always starts on a 32-bit A chunk
  if (sps < 2) sps = 2;
  a.stages_per_split = sps;
  a.ksplit           = total_stages > 0 ? (total_stages + sps - 1) / sps : 1;
  if (a.ksplit < 1) a.ksplit = 1;
  if (a.ksplit > 1 && a.mode == 0) return hipErrorInvalidValue;  // caller must pre-zero C and pass mode 1
  const long long nwg = (long long)a.tiles_m * a.tiles_n * a.ksplit * a.batch;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nwg), block(LEAF_THREADS);
  if (ug == 0) ug = LEAF_DEFAULT_UG(rg);
  if (pipe != 0) return hipErrorInvalidValue;
#define LEAF_CASE(RGV, UGV)                                                                        \
  if (rg == RGV && ug == UGV) {                                                                      \
    if (a.mode == 0) hipLaunchKernelGGL((m4rm_leaf_kernel<RGV, UGV, false>), grid, block, 0, stream, a); \
    else             hipLaunchKernelGGL((m4rm_leaf_kernel<RGV, UGV, true>), grid, block, 0, stream, a);  \
    return hipGetLastError();                                                                      \
  }
  LEAF_VARIANTS(LEAF_CASE)
#undef LEAF_CASE
  return hipErrorInvalidValue;
}

extern "C" hipError_t gf2_launch_m4rm_leaf(hipStream_t stream, LeafArgs a, int rg) {
  return gf2_launch_m4rm_leaf_variant(stream, a, rg, 0, 0);
}
