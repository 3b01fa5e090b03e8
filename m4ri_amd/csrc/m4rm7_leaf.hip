// m4rm7_leaf.hip -- M4RM leaf, generation 2: 7-bit tables, double-buffered, every wave symmetric.
//
// Same result, tile shape and lane layout as m4rm_leaf.hip (C-stationary 32*RG rows x 2048 columns in
// VGPRs; 256-byte table entries; four table rows per ds_read_b128, bank-conflict free; one
// v_perm_b32 per lookup address; one v_bitop3_b32 per dword folds two lookups), but the LDS pipe is
// kept busy through the whole inner loop:
//
//   * measured on MI355X (profiles/r01_ubench*.log): ds_write_b128 issues at ~77 B/clk/CU (13 clk
//     per wave-instruction, the VGPR->LDS data path), ds_read_b128 at 256 B/clk/CU.  The two-phase
//     kernel pays them back to back (1700 + 2048 clk per 16 inner bits) plus two pipe drains; the
//     write path and the read return path are different resources, so they can overlap.
//   * k = 7: a table is 128 entries x 256 B = 32 KiB, a stage is TWO tables = 14 inner bits, and two
//     stages' tables (128 KiB) are resident: while every wave gathers from stage s's tables it also
//     writes stage s+1's -- one entry (4 XORs + one ds_write_b128) after each of its 8 row groups.
//     Every thread builds 8 entries of one table from 7 rows of B (4 for the base, 3 for a 3-bit
//     Gray chain): the same B traffic per inner bit as before, no role specialisation, ONE barrier
//     per 14 bits.
//   * A is consumed in a pre-packed form (a7_pack_kernel): every 28 inner bits of a row become one
//     dword of four byte-aligned 7-bit indices, the odd bytes carrying 0x80 = "second table of the
//     stage", so the address of a lookup is still a single v_perm_b32.
//
// Replaces (result-identical) _mzd_mul_m4rm, mzd_make_table and _mzd_combine_N of the reference
// (/root/reference m4ri/brilliantrussian.c:1032-1190, :163-211, m4ri/xor_template.h:12-227); the
// reference itself lowers k below 8 for wide matrices (brilliantrussian.c:1075-1089).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gf2_common.h"

namespace {

constexpr int K7_BITS  = 7;             // bits per table index
constexpr int K7_STAGE = 2 * K7_BITS;   // inner bits per stage (two tables)
constexpr int K7_CHUNK = 2 * K7_STAGE;  // inner bits per packed A dword (two stages)

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

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

// v_perm_b32(a, coloff, sel): byte j of a -> bits 8..15 (index | table<<7), coloff.byte0 -> bits
// 0..7 (column slot), buffer -> bit 16 (taken from coloff.byte1 == 0x01)
__device__ __forceinline__ constexpr uint32_t perm_sel(int j, int buf) {
  return 0x0c000000u | ((buf ? 0x01u : 0x0cu) << 16) | ((uint32_t)(4 + j) << 8) | 0x00u;
}

// ---- A -> A7: row r, chunk q (inner bits 28q .. 28q+27) -> one dword of four 7-bit indices.
// Stored CHUNK-major, A7[q][r] with m_pad (multiple of 4) rows per chunk, so that the four
// consecutive rows a lane handles per read group are one 16-byte load.  The transposition goes
// through LDS: a workgroup reads a 64-row x 28-word (= 64 chunks exactly) tile of A along the rows
// and writes 64 chunks x 64 rows of A7 along the rows of A7 -- both sides coalesced.
constexpr int PK_ROWS = 64, PK_CHUNKS = 64, PK_WORDS = PK_CHUNKS * K7_CHUNK / 64;  // 28 words
__global__ __launch_bounds__(256) void a7_pack_kernel(const word *__restrict__ A, int64_t a_stride, int64_t a_bs,
                                                      uint32_t *__restrict__ A7, int64_t m_pad, int64_t apk_bs,
                                                      int64_t m, int64_t l, int64_t row_tiles, int64_t chunk_tiles) {
  __shared__ word tile[PK_ROWS][PK_WORDS + 1];
  const int64_t nq = (l + K7_CHUNK - 1) / K7_CHUNK;
  const int64_t wa = (l + 63) >> 6;
  int64_t bid      = blockIdx.x;
  const int64_t ct = bid % chunk_tiles; bid /= chunk_tiles;
  const int64_t rt = bid % row_tiles;   bid /= row_tiles;
  const int64_t b  = bid;
  const int64_t r0 = rt * PK_ROWS, q0 = ct * PK_CHUNKS, w0 = ct * PK_WORDS;
  for (int i = threadIdx.x; i < PK_ROWS * (PK_WORDS + 1); i += 256) {
    const int r = i / (PK_WORDS + 1), w = i - r * (PK_WORDS + 1);
    word v = 0;
    if (r0 + r < m && w0 + w < wa) v = A[b * a_bs + (r0 + r) * a_stride + (w0 + w)];
    tile[r][w] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < PK_ROWS * PK_CHUNKS; i += 256) {
    const int r = i % PK_ROWS, ql = i / PK_ROWS;
    const int64_t q = q0 + ql;
    if (q >= nq || r0 + r >= m_pad) continue;
    const int bit0l = ql * K7_CHUNK;  // bit offset inside the tile
    const int w = bit0l >> 6, sh = bit0l & 63;
    const word lo = tile[r][w], hi = tile[r][w + 1];
    uint32_t v = (uint32_t)((lo >> sh) | (sh ? (hi << (64 - sh)) : 0)) & 0x0fffffffu;
    const int64_t left = l - q * K7_CHUNK;  // inner bits that exist from this chunk on (>= 1)
    if (left < K7_CHUNK) v &= (1u << left) - 1u;
    // rows m .. m_pad-1 come out as index 0 of both tables (zero entries) because the tile is 0 there
    A7[b * apk_bs + q * m_pad + r0 + r] =
        (v & 0x7fu) | ((((v >> 7) & 0x7fu) | 0x80u) << 8) | (((v >> 14) & 0x7fu) << 16) | ((((v >> 21) & 0x7fu) | 0x80u) << 24);
  }
}

template <int RG, int UG, bool PIPE, bool XOR_OUT>
__global__ __launch_bounds__(LEAF_THREADS) void m4rm7_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 65536];  // [buffer][table][128][256 B]
  constexpr int R  = 32 * RG;  // tile rows: 32 row groups (8 waves x 4) x RG rows
  constexpr int NG = RG / UG;  // row groups per stage (>= 8: a thread writes one table entry with each of the first 8)
  static_assert(RG % UG == 0 && NG >= 8, "need at least 8 row groups per stage");

  const int tid  = threadIdx.x;
  const int c    = tid & 15;         // 16-byte column slot of the 256-byte table entry
  const int rgrp = tid >> 4;         // row group 0..31
  const int bz   = tid >> 8;         // build role: table 0/1 of the stage
  const int bhi  = (tid >> 4) & 15;  //             bits 3..6 of the entries this thread writes

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

  const uint32_t *Apkb = p.Apk + bat * p.apk_bs;
  const word *Bb      = p.B + bat * p.b_bs;
  word *__restrict__ Cb = p.C + bat * p.c_bs;

  const int nq = (p.l + K7_CHUNK - 1) / K7_CHUNK;
  // A7 and B are read through raw buffer descriptors: per-lane 32-bit offsets from a wave-uniform
  // base, and the hardware range check returns 0 for rows >= m of A7 and rows >= l of B -- exactly
  // the zero padding the algorithm wants, so the main loop has no edge branches.
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(Apkb, (uint32_t)((int64_t)nq * p.apk_stride * 4));  // apk_stride = m_pad
  const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(Bb, (uint32_t)(((int64_t)(p.l - 1) * p.b_stride + p.wn) * 8));

  const int w0   = tile_n * LEAF_TW + c * 2;  // this lane's two words of the row
  const bool v0  = w0 < p.wn;
  const bool v1  = (w0 + 1) < p.wn;
  const int row0 = tile_m * R + rgrp * RG;
  const uint32_t a_qs   = (uint32_t)p.apk_stride * 4u;  // bytes between chunks of A7 (m_pad rows)
  const uint32_t b_rs   = (uint32_t)p.b_stride * 8u;
  const uint32_t a_lane = (uint32_t)row0 * 4u;
  // B offsets = wave-uniform part (table, tile column: SGPRs) + the lane's 16-byte slot; keeping the
  // uniform part out of VGPRs avoids spilled offsets (a scratch reload costs a vmcnt(0) drain)
  const uint32_t b_uni  = (uint32_t)__builtin_amdgcn_readfirstlane(bz) * K7_BITS * b_rs + (uint32_t)tile_n * (LEAF_TW * 8u);
  const uint32_t b_slot = (uint32_t)c * 16u;
  const uint32_t coloff = (uint32_t)(c * 16) | 0x0100u;  // byte0 = column slot, byte1 = 0x01 (buffer bit)
  unsigned char *const wr_base = lds + bz * 32768 + bhi * 8 * 256 + c * 16;

  uint32_t acc[RG][4];
#pragma unroll
  for (int t = 0; t < RG; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0u; }

  const int q_begin = ks * p.chunks_per_split;
  int q_end         = q_begin + p.chunks_per_split;
  if (q_end > nq) q_end = nq;

  // B rows of the table this thread helps to build: rows 3..6 of the 7 (-> base) and rows 0..2
  // (-> Gray chain).  Columns outside the matrix may hold a neighbour's bits when B is a window;
  // they only reach C columns that are never stored.
  uint4 bhi_rows[4], blo_rows[3];
  auto load_hi = [&](int stage) {
    uint32_t off = (b_uni + ((uint32_t)stage * K7_STAGE + 3u) * b_rs) + b_slot;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bhi_rows[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
      off += b_rs;
      asm volatile("" : "+v"(off));  // one running offset VGPR instead of hoisted per-row offsets
    }
  };
  auto load_lo = [&](int stage) {
    uint32_t off = (b_uni + (uint32_t)stage * K7_STAGE * b_rs) + b_slot;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      blo_rows[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
      off += b_rs;
      asm volatile("" : "+v"(off));
    }
  };
  uint32_t cur[4];
  auto make_base = [&]() {
    // the rows become visible to the optimiser only here (volatile asm stays behind the previous
    // barrier); un-pinned, hipcc hoists these XORs up to the loads and waits out their latency
#pragma unroll
    for (int j = 0; j < 4; ++j)
      asm volatile("" : "+v"(bhi_rows[j].x), "+v"(bhi_rows[j].y), "+v"(bhi_rows[j].z), "+v"(bhi_rows[j].w));
    cur[0] = cur[1] = cur[2] = cur[3] = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool on = (bhi >> j) & 1;
      cur[0] ^= on ? bhi_rows[j].x : 0u;
      cur[1] ^= on ? bhi_rows[j].y : 0u;
      cur[2] ^= on ? bhi_rows[j].z : 0u;
      cur[3] ^= on ? bhi_rows[j].w : 0u;
    }
  };
  // entry number i (0..7) of the thread's 8: Gray step + one ds_write_b128 into buffer `buf`
  auto put_entry = [&](int i, int buf) {
    if (i > 0) {
      const int j = __builtin_ctz(i);
      if (i == 1) {
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
          asm volatile("" : "+v"(blo_rows[jj].x), "+v"(blo_rows[jj].y), "+v"(blo_rows[jj].z), "+v"(blo_rows[jj].w));
      }
      cur[0] ^= blo_rows[j].x;
      cur[1] ^= blo_rows[j].y;
      cur[2] ^= blo_rows[j].z;
      cur[3] ^= blo_rows[j].w;
    }
    // keep the Gray chain a chain (one XOR + one ds_write_b128 per entry)
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
    const int gcode = i ^ (i >> 1);
    *reinterpret_cast<uint4 *>(wr_base + buf * 65536 + gcode * 256) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
  };

  static_assert(UG == 4, "the A refill is one 16-byte load per read group of 4 rows");
  uint32_t areg[RG];
  auto load_a4 = [&](int g, int q) {
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  a_rsrc, (int)(a_lane + (uint32_t)q * a_qs + (uint32_t)g * 16u), 0, 0));
    areg[g * 4 + 0] = v.x; areg[g * 4 + 1] = v.y; areg[g * 4 + 2] = v.z; areg[g * 4 + 3] = v.w;
  };
#pragma unroll
  for (int g = 0; g < RG / 4; ++g) load_a4(g, q_begin);

  // one stage: gather from the tables of stage s = 2q+J (buffer J) while building those of stage
  // s+1 into buffer J^1; in the second stage of a chunk the A registers are refilled on the way
  auto stage = [&](auto jtag, int q) {
    constexpr int J = decltype(jtag)::value;
    const int s     = 2 * q + J;
    // on entry: cur = base of the tables of stage s+1 (made late in the previous stage), blo_rows =
    // their chain rows, bhi_rows = the base rows of stage s+2.  Nothing but gathers happens right
    // behind the barrier: all 8 waves come out of it together, and whatever non-LDS work sits here
    // (VMEM issue, base XORs) would idle the LDS pipe for every one of them at once.
    uint4 t0[PIPE ? 2 : 1][UG], t1[PIPE ? 2 : 1][UG];
    auto issue = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        const uint32_t a0 = __builtin_amdgcn_perm(areg[g * UG + u], coloff, perm_sel(2 * J + 0, J));
        const uint32_t a1 = __builtin_amdgcn_perm(areg[g * UG + u], coloff, perm_sel(2 * J + 1, J));
        t0[slot][u]       = *reinterpret_cast<const uint4 *>(lds + a0);
        t1[slot][u]       = *reinterpret_cast<const uint4 *>(lds + a1);
      }
      // second stage of a chunk: these four rows' last indices are out, refill their A registers
      // with the next chunk right away (one 16-byte load)
      if constexpr (J == 1) load_a4(g, q + 1);
    };
    auto fold = [&](int g, int slot) {
#pragma unroll
      for (int u = 0; u < UG; ++u) {
        uint32_t *a = acc[g * UG + u];
        a[0] = xor3(a[0], t0[slot][u].x, t1[slot][u].x);
        a[1] = xor3(a[1], t0[slot][u].y, t1[slot][u].y);
        a[2] = xor3(a[2], t0[slot][u].z, t1[slot][u].z);
        a[3] = xor3(a[3], t0[slot][u].w, t1[slot][u].w);
        // pin the accumulation here (XOR is associative: un-pinned, hipcc re-associates the whole
        // stage into one late XOR tree and keeps every loaded table row live)
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      }
    };
    if constexpr (PIPE) { issue(0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // software pipeline: group g+1's gathers go out before group g is folded, so the LDS queue
      // never drains while this wave is busy with XORs, table writes and VMEM issue
      if constexpr (PIPE) { if (g + 1 < NG) issue(g + 1, (g + 1) & 1); }
      else issue(g, 0);
      // the 8 table entries go out with the FIRST 8 groups, so the chain rows are dead early and
      // their successors (first needed one group into the next stage) get most of a stage to arrive
      if (g < 8) put_entry(g, J ^ 1);
      if (g == 8 || (NG == 8 && g == 7)) load_lo(s + 2);
      if (g == (NG > 10 ? 10 : NG - 1)) {
        make_base();     // base of stage s+2's tables (their entries are written during stage s+1)
        load_hi(s + 3);  // and the base rows after that: a whole stage of latency budget
      }
      __builtin_amdgcn_sched_barrier(0);
      fold(g, PIPE ? (g & 1) : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  if (q_begin < q_end) {
    // prologue: tables of the first stage (buffer 0), then the rows for building the second
    load_hi(2 * q_begin);
    load_lo(2 * q_begin);
    make_base();
#pragma unroll
    for (int i = 0; i < 8; ++i) put_entry(i, 0);
    load_hi(2 * q_begin + 1);
    load_lo(2 * q_begin + 1);
    make_base();
    load_hi(2 * q_begin + 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    for (int q = q_begin; q < q_end; ++q) {
      stage(std::integral_constant<int, 0>{}, q);
      stage(std::integral_constant<int, 1>{}, q);
    }
  }

  // epilogue: C tile out.  One running row pointer (pinned, so hipcc cannot hoist RG 64-bit row
  // addresses above the main loop); the column guards are loop-invariant per lane.
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

// words of workspace the packed copy of A needs for a launch (uint32 units rounded to 64-bit words)
extern "C" int64_t gf2_m4rm7_a7_words(int64_t m, int64_t l, int64_t batch) {
  const int64_t nq = (l + K7_CHUNK - 1) / K7_CHUNK;
  return (batch * ((m + 3) & ~(int64_t)3) * nq + 1) / 2;
}

// Host launchers.  gf2_launch_a7_pack fills `a7_ws` (gf2_m4rm7_a7_words words) with the packed copy of
// A; gf2_launch_m4rm7 runs the leaf on it.  rg: tile height / 32 (rows = 32*rg).
static bool k7_geometry(LeafArgs &a, word *a7_ws, int rg, int64_t &nq, int64_t &m_pad) {
  const int R = 32 * rg;
  a.wn        = (int32_t)words_of(a.n);
  a.tiles_m   = (a.m + R - 1) / R;
  a.tiles_n   = (a.wn + LEAF_TW - 1) / LEAF_TW;
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0 || a.l <= 0) return false;
  nq          = (a.l + K7_CHUNK - 1) / K7_CHUNK;
  m_pad       = ((int64_t)a.m + 3) & ~(int64_t)3;
  a.Apk        = reinterpret_cast<const uint32_t *>(a7_ws);
  a.apk_stride = m_pad;
  a.apk_bs     = m_pad * nq;
  return true;
}

extern "C" hipError_t gf2_launch_a7_pack(hipStream_t stream, LeafArgs a, word *a7_ws) {
  int64_t nq, m_pad;
  if (!k7_geometry(a, a7_ws, 32, nq, m_pad)) return hipSuccess;
  if ((uint64_t)m_pad * (uint64_t)nq * 4 >= (1ull << 32)) return hipErrorInvalidValue;
  const int64_t row_tiles = (m_pad + PK_ROWS - 1) / PK_ROWS, chunk_tiles = (nq + PK_CHUNKS - 1) / PK_CHUNKS;
  const int64_t g = row_tiles * chunk_tiles * a.batch;
  if (g > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(a7_pack_kernel, dim3((unsigned)g), dim3(256), 0, stream, a.A, a.a_stride, a.a_bs,
                     reinterpret_cast<uint32_t *>(a7_ws), m_pad, a.apk_bs, (int64_t)a.m, (int64_t)a.l, row_tiles, chunk_tiles);
  return hipGetLastError();
}

extern "C" hipError_t gf2_launch_m4rm7(hipStream_t stream, LeafArgs a, word *a7_ws, int rg, int ug, int pipe) {
  int64_t nq, m_pad;
  if (!k7_geometry(a, a7_ws, rg, nq, m_pad)) return hipSuccess;
  if ((uint64_t)m_pad * (uint64_t)nq * 4 >= (1ull << 32)) return hipErrorInvalidValue;
  if (a.ksplit < 1) a.ksplit = 1;
  int cps = (int)((nq + a.ksplit - 1) / a.ksplit);
  if (cps < 1) cps = 1;
  a.chunks_per_split = cps;
  a.ksplit           = (int)((nq + cps - 1) / cps);
  if (a.ksplit > 1 && a.mode == 0) return hipErrorInvalidValue;  // caller must pre-zero C and pass mode 1
  const long long nwg = (long long)a.tiles_m * a.tiles_n * a.ksplit * a.batch;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nwg), block(LEAF_THREADS);
#define K7_CASE(RGV, UGV, PV)                                                                        \
  if (rg == RGV && ug == UGV && pipe == PV) {                                                          \
    if (a.mode == 0) hipLaunchKernelGGL((m4rm7_kernel<RGV, UGV, PV != 0, false>), grid, block, 0, stream, a); \
    else             hipLaunchKernelGGL((m4rm7_kernel<RGV, UGV, PV != 0, true>), grid, block, 0, stream, a);  \
    return hipGetLastError();                                                                        \
  }
  K7_CASE(32, 4, 0)
#undef K7_CASE
  return hipErrorInvalidValue;
}
