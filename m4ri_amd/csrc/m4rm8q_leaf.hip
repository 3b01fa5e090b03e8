// m4rm8q_leaf.hip -- M4RM leaf, generation 4: 8-bit tables with 64-byte entries, FOUR tables
// interleaved per LDS bank row, double-buffered, software-pipelined gathers, table building on the
// four older waves.
//
// One more turn of the screw of generation 3 (retired; its description is in DESIGN.md 3.1).  The leaf is bound by LDS-array cycles,
// and for a tile of fixed area the gathers cost the same while the table writes shrink with the
// entry size, because more rows share every entry:
//
//     entry 256 B (gen 2): 1024 x 2048 tile, 2560 clk per 14 inner bits  -> 183 clk/bit
//     entry 128 B (gen 3): 2048 x 1024 tile, 2560 clk per 16 inner bits  -> 160 clk/bit
//     entry  64 B (here) : 4096 x  512 tile, 4608 clk per 32 inner bits  -> 144 clk/bit
//
// A stage is a whole 32-bit word of A = four 256-entry tables (4 x 16 KiB), two stages resident.
//   * LDS bank row x holds [T0[x] | T1[x] | T2[x] | T3[x]], 64 bytes each.
//   * lane = (row group lane>>2, 16-byte slot lane&3).  A row takes four gathers per stage; in
//     gather i the row group reads table (rot + i) & 3 with rot = (row group >> 1) & 3.  Each of
//     ds_read_b128's four 16-lane service groups holds four row groups -- {0,3,5,6}, {1,2,4,7},
//     {8,11,13,14}, {9,10,12,15} -- whose `rot` values are 0,1,2,3 in every case, so the four row
//     groups always sit in four different quarters of the bank row: conflict-free for ANY indices.
//   * The packed A (a4_pack_kernel of a4_pack.hip, or the fused Winograd pass of aux_kernels.hip)
//     holds a row's four index bytes already rotated by `rot`, so byte i of the dword IS gather i's
//     index and the four v_perm selectors are compile-time constants.
//   * Software pipeline: row g+1's four gathers are issued before row g's XORs, so a wave keeps
//     4..8 reads in flight while its VALU works; the A dwords live in a ring of the next 16 rows.
//   * Table building belongs to waves 0..3 (16 entries per thread and stage: 4 B rows for the
//     base, 4 for a 4-bit Gray chain).  On every SIMD the wave dispatched first wins issue
//     arbitration; with symmetric work it reached the stage's barrier ~15 % early and idled there.
//     Giving it ALL of the building (B loads, base XORs, chain XORs, ds_writes) evens the two out:
//     -6.5 % on the launch.  (The younger waves as builders: no gain.)  16 consecutive builder lanes
//     = 4 slots x 4 tables of ONE entry index = one whole bank row per ds_write service group, so
//     the table writes are conflict-free as well.
//
//   * A launch covers a RANGE of the batch's tiles (LeafArgs::tile_base/tile_count) and may split the
//     inner dimension (ksplit); the engine uses both to run full rounds of 256 workgroups unsplit and
//     only the short last round split.  Splits combine without atomics: mode 2 stores every
//     (tile, split) into its own dense slab and gf2_launch_reduce_partials folds the slabs into C.
//   * Waves whose 512 rows all lie below the matrix (last row tile of a ragged m) skip their gathers; the gather-only waves own
//     the tile's FIRST 2048 rows, so a partly filled tile keeps its rows away from the table building (round 4).
//
// Everything else is generation 3's: C-stationary tile in VGPRs (128 dwords per lane), one
// v_perm_b32 per lookup address, one v_bitop3_b32 per dword folds two lookups, Gray-code table
// build, chunk-major A, range-checked buffer descriptors, one barrier per stage.
//
// Replaces (result-identical) _mzd_mul_m4rm, mzd_make_table and _mzd_combine_N of the reference
// (/root/reference m4ri/brilliantrussian.c:1032-1190, :163-211, m4ri/xor_template.h:12-227).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "gf2_common.h"

// which half of the workgroup builds the tables: 0 = waves 0..3 (dispatched first), 1 = waves 4..7
#ifndef K8Q_BUILDER_HALF
#define K8Q_BUILDER_HALF 0
#endif
// rows in flight ahead of the XORs, per role
#ifndef K8Q_PD_BUILDER
#define K8Q_PD_BUILDER 1
#endif
// rows of A held ahead by the gather-only waves (the builders hold K8_AR = 16)
#ifndef K8Q_AR_OTHER
#define K8Q_AR_OTHER 32  // a whole stage ahead: the same on full tiles, a little better on partly filled ones (their rows live on these waves)
#endif
#ifndef K8Q_PD_OTHER
#define K8Q_PD_OTHER 1  // 2 and 3 fit in the gather-only waves' registers and measure the same, on full and on partly filled tiles
#endif

namespace {

constexpr int K8_BITS  = 8;             // bits per table index
constexpr int K8_STAGE = 4 * K8_BITS;   // inner bits per stage (four tables) = one dword of A
constexpr int K8_CHUNK = K8_STAGE;      // inner bits per A dword
constexpr int K8_TW    = 8;             // tile width in words (512 columns, 64 B per entry)
constexpr int K8_RG    = 32;            // rows per lane
constexpr int K8_R     = 128 * K8_RG;   // tile rows: 128 row groups (8 waves x 16) x 32 rows
constexpr int K8_AR    = 16;            // rows of A held ahead (ring)

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

// v_perm_b32(a, coloff, sel): byte j of a -> bits 8..15 (table index), coloff.byte0 -> bits 0..7
// (table quarter + column slot), buffer -> bit 16 (taken from coloff.byte1 == 0x01)
__device__ __forceinline__ constexpr uint32_t perm_sel(int j, int buf) {
  return 0x0c000000u | ((buf ? 0x01u : 0x0cu) << 16) | ((uint32_t)(4 + j) << 8) | 0x00u;
}

#ifndef K8Q_DEBUG_SKIP
#define K8Q_DEBUG_SKIP 0
#endif
// developer decomposition of the launch time (tools/r03_leaf_decomposition.sh; results are garbage with any bit set):
// 1 = nobody builds tables (no B loads, no table XORs, no ds_writes), 2 = nobody gathers, 4 = no stage barrier
#ifndef K8Q_EXP
#define K8Q_EXP 0
#endif

template <bool XOR_OUT>
__global__ __launch_bounds__(LEAF_THREADS) void m4rm8q_kernel(const LeafArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 65536];  // [buffer][256 bank rows][T0|T1|T2|T3][64 B]
  constexpr int RG = K8_RG, AR_MAX = 32;

  const int tid  = threadIdx.x;
  const int c    = tid & 3;          // 16-byte column slot of the 64-byte table entry
  const int rgrp = tid >> 2;         // row group 0..127
  const int rot  = (tid >> 3) & 3;   // table this lane reads in the FIRST of a row's four gathers
  // build role (builder waves only): 16 consecutive lanes = 4 slots x 4 tables of one entry index
  const int bz   = (tid >> 2) & 3;   // table 0..3 of the stage
  const int bhi  = (tid >> 4) & 15;  // bits 4..7 of the 16 entries this thread writes

  // block -> (tile, inner split); tile -> (batch, tile_n, tile_m).  Consecutive tiles share a B
  // panel, and the XCD remap keeps them on one XCD's L2 (blocks are dispatched round-robin over 8
  // XCDs)
  uint32_t lid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x;
    if ((nwg & 7u) == 0u) lid = (lid & 7u) * (nwg >> 3) + (lid >> 3);
  }
  const int ks = lid % p.ksplit;
  const uint32_t slab = lid;  // (tile - tile_base) * ksplit + split: this workgroup's slab in mode 2
  int64_t t    = p.tile_base + lid / p.ksplit;
  const int tile_m = (int)(t % p.tiles_m); t /= p.tiles_m;
  const int tile_n = (int)(t % p.tiles_n); t /= p.tiles_n;
  const int64_t bat = t;

  const uint32_t *Apkb = p.Apk + bat * p.apk_bs;
  const word *Bb      = p.B + bat * p.b_bs;
  word *__restrict__ Cb = p.C + bat * p.c_bs;

  const int nq = 2 * ((p.l + 63) / 64);  // stages = dwords of A per row
  // The packed A and B are read through raw buffer descriptors: per-lane 32-bit offsets from a wave-uniform
  // base, and the hardware range check returns 0 for rows >= m of the packed A and rows >= l of B -- exactly
  // the zero padding the algorithm wants, so the main loop has no edge branches.
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(Apkb, (uint32_t)((int64_t)nq * p.apk_stride * 4));  // apk_stride = m_pad
  const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(Bb, (uint32_t)(((int64_t)(p.l - 1) * p.b_stride + p.wn) * 8));

  const int w0   = tile_n * K8_TW + c * 2;  // this lane's two words of the row
  const bool v0  = w0 < p.wn;
  const bool v1  = (w0 + 1) < p.wn;
  // the gather-only waves (4..7) own the FIRST 2048 rows of the tile, the builder waves the last: the rows of a partly filled tile then
  // do not share their waves' issue slots with the table building (a tile of at most 2048 rows: 1.80 instead of 2.10 us per stage;
  // same box, alternating builds: 464 x 66000 x 66000 2.34 -> 2.07 ms, 2048 x 65536 x 65536 2.10 -> 1.80 ms, full tiles unchanged --
  // profiles/r04_leaf_partial_tiles_variants.log).  Such a tile stays issue-bound on its one gather wave per SIMD (32 rows x 4
  // gathers = ~520 instructions a stage), not LDS-bound: a deeper gather pipeline (K8Q_PD_OTHER 2, 3), the A dwords further ahead and
  // builder waves with four stages of B rows ahead all measured the same.  Bits 1..2 of the row group, which set `rot`, are untouched.
  const int rowg = rgrp ^ 64;        // row group of the TILE this lane owns
  const int row0 = tile_m * K8_R + rowg * RG;
  const uint32_t a_qs   = (uint32_t)p.apk_stride * 4u;  // bytes between chunks of the packed A (m_pad rows)
  const uint32_t b_rs   = (uint32_t)p.b_stride * 8u;
  const uint32_t a_lane = (uint32_t)row0 * 4u;
  // B offsets = wave-uniform part (tile column, stage: SGPRs) + ONE per-lane VGPR (table, 16-byte
  // slot).  Keeping the uniform part out of VGPRs matters: at 256 VGPRs a spilled offset costs a
  // scratch reload + vmcnt(0), i.e. a full drain of the A/B prefetches, twice per stage pair
  const uint32_t b_uni  = (uint32_t)__builtin_amdgcn_readfirstlane(tile_n) * (K8_TW * 8u);
  const uint32_t b_slot = (uint32_t)bz * K8_BITS * b_rs + (uint32_t)c * 16u;
  // per-lane perm operand of gather i: byte0 = table quarter ((rot + i) & 3) * 64 + column slot,
  // byte1 = 0x01 (buffer bit source)
  uint32_t coloff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) coloff[i] = (uint32_t)(((rot + i) & 3) * 64 + c * 16) | 0x0100u;
  unsigned char *const wr_base = lds + bhi * 16 * 256 + bz * 64 + c * 16;

  uint32_t acc[RG][4];
#pragma unroll
  for (int t = 0; t < RG; ++t) { acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0u; }

  const int q_begin = ks * p.chunks_per_split;
  int q_end         = q_begin + p.chunks_per_split;
  if (q_end > nq) q_end = nq;

  // B rows of the table this thread helps to build: rows 4..7 of the 8 (-> base) and rows 0..3
  // (-> Gray chain).  Columns outside the matrix may hold a neighbour's bits when B is a window;
  // they only reach C columns that are never stored.
  uint4 bhi_rows[4], blo_rows[4];
  auto load_hi = [&](int stage) {
    uint32_t off = (b_uni + ((uint32_t)stage * K8_STAGE + 4u) * b_rs) + b_slot;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if K8Q_DEBUG_SKIP == 2  // developer probe (tools/prof_leaf_traffic_operands.sh): no B loads, results are garbage
      bhi_rows[j] = make_uint4(off, (uint32_t)stage, (uint32_t)j, 0u);
#else
      bhi_rows[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
#endif
      off += b_rs;
      asm volatile("" : "+v"(off));  // one running offset VGPR instead of hoisted per-row offsets
    }
  };
  auto load_lo = [&](int stage) {
    uint32_t off = (b_uni + (uint32_t)stage * K8_STAGE * b_rs) + b_slot;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if K8Q_DEBUG_SKIP == 2
      blo_rows[j] = make_uint4(off, (uint32_t)stage, (uint32_t)j, 1u);
#else
      blo_rows[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0));
#endif
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
  // entry number i (0..15) of the thread's 16: Gray step + one ds_write_b128 into buffer `buf`
  auto put_entry = [&](int i, int buf) {
    if (i > 0) {
      const int j = __builtin_ctz(i);
      if (i == 1) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
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

  // the A dwords of the next 16 rows (a whole stage would cost the registers the pipeline needs)
  uint32_t areg[AR_MAX];
  auto load_a4 = [&](int slot, int g, int q) {  // rows 4g..4g+3 of stage q -> areg[4*slot ..]
#if K8Q_DEBUG_SKIP == 1  // developer probe: no A loads
    const uint32_t fake = a_lane + (uint32_t)q * a_qs + (uint32_t)g * 16u;
    const uint4 v = make_uint4(fake, fake * 3u, fake * 5u, fake * 7u);
#else
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  a_rsrc, (int)(a_lane + (uint32_t)q * a_qs + (uint32_t)g * 16u), 0, 0));
#endif
    areg[slot * 4 + 0] = v.x; areg[slot * 4 + 1] = v.y; areg[slot * 4 + 2] = v.z; areg[slot * 4 + 3] = v.w;
  };

  // one stage: gather from the four tables of stage s (buffer J = s & 1) while the builder waves
  // write those of stage s+1 into buffer J^1
  auto stage = [&](auto jtag, auto btag, auto atag, int s) {
    constexpr int J        = decltype(jtag)::value;
    constexpr bool BUILDER = decltype(btag)::value;
    constexpr bool ACTIVE  = decltype(atag)::value;  // false: every row of this wave lies below the matrix
    if constexpr (!ACTIVE) {
      // a wave of padding rows gathers nothing (a partly filled last row tile then costs the LDS
      // array only its real rows); a builder still delivers its share of the next tables
      if constexpr (BUILDER && !(K8Q_EXP & 1)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) put_entry(i, J ^ 1);
        load_lo(s + 2);
        make_base();
        load_hi(s + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(K8Q_EXP & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    // builder, on entry: cur = base of this thread's table of stage s+1 (made late in the previous
    // stage), blo_rows = its chain rows, bhi_rows = the base rows of stage s+2
    // pipeline depth: rows whose gathers are in flight ahead of the XORs
    constexpr int PD = BUILDER ? K8Q_PD_BUILDER : K8Q_PD_OTHER;
    constexpr int AR = BUILDER ? K8_AR : K8Q_AR_OTHER;  // rows of A held ahead in this role
    uint4 tp[PD + 1][4];
    auto issue = [&](int g) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t ad  = __builtin_amdgcn_perm(areg[g % AR], coloff[i], perm_sel(i, J));
        tp[g % (PD + 1)][i] = *reinterpret_cast<const uint4 *>(lds + ad);
      }
    };
#pragma unroll
    for (int g = 0; g < PD; ++g) issue(g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const int j = g + PD;  // the row whose gathers go out now
      if (j < RG) issue(j);
      if (j < RG && j % 4 == 3) {  // rows 4k..4k+3 have all issued: their ring slot takes the rows 16 ahead
        const int k = j / 4, ahead = 4 * k + AR;
        load_a4(k % (AR / 4), (ahead % RG) / 4, s + ahead / RG);
      }
      if constexpr (BUILDER && !(K8Q_EXP & 1)) {
        // the 16 table entries go out with the first 16 rows, so the chain rows are dead early and
        // their successors (first needed two rows into the next stage) get half a stage to arrive
        if (g < 16) put_entry(g, J ^ 1);
        if (g == 17) load_lo(s + 2);
        if (g == 21) {
          make_base();     // base of stage s+2's table (its entries are written during stage s+1)
          load_hi(s + 3);  // and the base rows after that: a whole stage of latency budget
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      uint32_t *a     = acc[g];
      const uint4 *tt = tp[g % (PD + 1)];
      a[0] = xor3(xor3(a[0], tt[0].x, tt[1].x), tt[2].x, tt[3].x);
      a[1] = xor3(xor3(a[1], tt[0].y, tt[1].y), tt[2].y, tt[3].y);
      a[2] = xor3(xor3(a[2], tt[0].z, tt[1].z), tt[2].z, tt[3].z);
      a[3] = xor3(xor3(a[3], tt[0].w, tt[1].w), tt[2].w, tt[3].w);
      // pin the accumulation here (XOR is associative: un-pinned, hipcc re-associates the whole
      // stage into one late XOR tree and keeps every loaded table row live)
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!(K8Q_EXP & 4)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  };

  // the two roles are two loops (wave-uniform branch), so each gets its own register allocation:
  // the gather-only waves carry no B rows at all
  auto run = [&](auto btag, auto atag) {
    constexpr bool BUILDER = decltype(btag)::value;
    constexpr bool ACTIVE  = decltype(atag)::value;
    constexpr int AR       = BUILDER ? K8_AR : K8Q_AR_OTHER;
    if constexpr (ACTIVE) {
#pragma unroll
      for (int g = 0; g < AR / 4; ++g) load_a4(g, g, q_begin);  // q = stage here: one dword of A per stage
    }
    if constexpr ((K8Q_EXP & 1) != 0) {  // keep the tables "written" for the optimiser: one store nobody ever executes
      if (p.m == -12345) *reinterpret_cast<uint4 *>(lds + (tid & 8191) * 16) = make_uint4(tid, 1u, 2u, 3u);
    }
    if constexpr (BUILDER && !(K8Q_EXP & 1)) {
      // prologue: tables of the first stage (buffer 0: q_begin is even), then the rows for the second
      load_hi(q_begin);
      load_lo(q_begin);
      make_base();
#pragma unroll
      for (int i = 0; i < 16; ++i) put_entry(i, 0);
      load_hi(q_begin + 1);
      load_lo(q_begin + 1);
      make_base();
      load_hi(q_begin + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // two stages per trip so the buffer parity is a compile-time constant; an odd tail runs one
    // extra stage whose B rows and A dwords lie past the end and read as 0
    for (int q = q_begin; q < q_end; q += 2) {
      stage(std::integral_constant<int, 0>{}, btag, atag, q);
      stage(std::integral_constant<int, 1>{}, btag, atag, q + 1);
    }
  };
  if (q_begin < q_end) {
    // a wave owns 512 consecutive rows of the tile; in the last row tile some waves own only padding
    const bool active  = !(K8Q_EXP & 2) && __builtin_amdgcn_readfirstlane(tile_m * K8_R + ((tid >> 6) ^ 4) * (16 * RG)) < p.m;
    const bool builder = __builtin_amdgcn_readfirstlane(tid >> 8) == K8Q_BUILDER_HALF;
    if (builder) { if (active) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
    else         { if (active) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }
  }

  // epilogue: C tile out.  One running row pointer (pinned, so hipcc cannot hoist RG 64-bit row
  // addresses above the main loop); the column guards are loop-invariant per lane.
  // Mode 2 (an inner-dimension split without atomics): the whole tile, padding included, goes
  // into this workgroup's own dense slab; gf2_launch_reduce_partials folds the slabs into C.
  const bool to_slab = !XOR_OUT && p.mode == 2;
  if (v0 || to_slab) {
    word *cp         = to_slab ? p.Cpart + (int64_t)slab * LEAF_PART_WORDS + (int64_t)rowg * RG * K8_TW + c * 2
                               : Cb + (int64_t)row0 * p.c_stride + w0;
    const int64_t cst = to_slab ? (int64_t)K8_TW : p.c_stride;
    const int rows    = to_slab ? RG : ((p.m - row0) < RG ? (p.m - row0) : RG);  // may be <= 0
    const bool s1     = to_slab || v1;
    const bool rmw    = XOR_OUT && p.ksplit == 1;
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      if (t < rows) {
        const word x0 = (word)acc[t][0] | ((word)acc[t][1] << 32);
        const word x1 = (word)acc[t][2] | ((word)acc[t][3] << 32);
        if constexpr (!XOR_OUT) {
          cp[0] = x0;
          if (s1) cp[1] = x1;
        } else if (rmw) {
          // C ^= tile with ONE owner per tile (no inner split): a plain read-modify-write -- the atomics below manage ~0.5 TB/s,
          // which is what the inner-dimension strip of a ragged 50000 x 12000 x 90000 paid for its 560 MB of C (1.2 ms of 9.9)
          cp[0] ^= x0;
          if (v1) cp[1] ^= x1;
        } else {
          // C ^= tile: a no-return L2 atomic needs no destination registers and is what makes
          // inner-dimension splits (ksplit > 1) race-free; XOR is exact, so order is moot
          atomicXor(reinterpret_cast<unsigned long long *>(cp), (unsigned long long)x0);
          if (v1) atomicXor(reinterpret_cast<unsigned long long *>(cp + 1), (unsigned long long)x1);
        }
      }
      cp += cst;
      asm volatile("" : "+v"(cp));
    }
  }
}

}  // namespace

// The split count a launch with inner dimension l really uses when asked for `ksplit`: every split
// takes an even number of stages (it starts in table buffer 0), so the count can come out smaller.
// The engine sizes and folds the mode-2 slabs with this number.
extern "C" int gf2_m4rm8q_effective_ksplit(int64_t l, int ksplit) {
  const int64_t nq = 2 * ((l + 63) / 64);
  if (ksplit < 1) ksplit = 1;
  int64_t cps = (nq + ksplit - 1) / ksplit;
  cps         = (cps + 1) & ~(int64_t)1;
  if (cps < 2) cps = 2;
  return (int)((nq + cps - 1) / cps);
}

// Host launcher.  A must already be packed chunk-major WITH the byte rotation (gf2_launch_a4_pack_rot
// of a4_pack.hip, rot = 1, or gf2_launch_winograd_down2_pack) into `a4_ws`.  Tiles are 4096 rows
// x 512 columns.
extern "C" hipError_t gf2_launch_m4rm8q(hipStream_t stream, LeafArgs a, word *a4_ws) {
  a.wn        = (int32_t)words_of(a.n);
  a.tiles_m   = (a.m + K8_R - 1) / K8_R;
  a.tiles_n   = (a.wn + K8_TW - 1) / K8_TW;
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0 || a.l <= 0) return hipSuccess;
  const int64_t nq    = 2 * (((int64_t)a.l + 63) / 64);
  const int64_t m_pad = ((int64_t)a.m + 3) & ~(int64_t)3;
  a.Apk        = reinterpret_cast<const uint32_t *>(a4_ws);
  a.apk_stride = m_pad;
  a.apk_bs     = m_pad * nq;
  if ((uint64_t)m_pad * (uint64_t)nq * 4 >= (1ull << 32)) return hipErrorInvalidValue;
  if (a.ksplit < 1) a.ksplit = 1;
  int cps = (int)((nq + a.ksplit - 1) / a.ksplit);
  cps     = (cps + 1) & ~1;  // even: a split starts in table buffer 0
  if (cps < 2) cps = 2;
  a.chunks_per_split = cps;
  a.ksplit           = (int)((nq + cps - 1) / cps);
  if (a.ksplit > 1 && a.mode == 0) return hipErrorInvalidValue;  // splits combine by atomics (1) or slabs (2)
  if (a.mode == 2 && a.Cpart == nullptr) return hipErrorInvalidValue;
  const long long ntiles = (long long)a.tiles_m * a.tiles_n * a.batch;
  if (a.tile_count == 0) { a.tile_base = 0; a.tile_count = ntiles; }
  if (a.tile_base < 0 || a.tile_base + a.tile_count > ntiles) return hipErrorInvalidValue;
  const long long nwg = (long long)a.tile_count * a.ksplit;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)nwg), block(LEAF_THREADS);
  if (a.mode != 1) hipLaunchKernelGGL((m4rm8q_kernel<false>), grid, block, 0, stream, a);
  else             hipLaunchKernelGGL((m4rm8q_kernel<true>), grid, block, 0, stream, a);
  return hipGetLastError();
}
