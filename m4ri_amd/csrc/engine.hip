// engine.hip -- host-side scheduler of the MI355X GF(2) multiply engine + the device-resident C ABI
// (include/m4ri_amd.h, part 2).
//
// Replaces the reference's recursive, memory-frugal Strassen-Winograd driver
//   _mzd_mul_even / _mzd_addmul_even      /root/reference m4ri/strassen.c:41-208, :367-526
// by a BREADTH-FIRST schedule sized for 288 GB of HBM: for L levels
//   "down" passes per operand    (a parent -> its Winograd operand combinations; the deepest
//                                 min(L, 4) levels in ONE fused pass, the A side written straight
//                                 into the leaf's packed form),
//   ONE batched M4RM leaf launch (all 7^L products at once: >> 256 workgroups),
//   "up" passes                  (products -> the quadrants of the parent, fused the same way; the
//                                 last one writes, or XORs into, the caller's C).
// The depth-first reference needs 2-3 quadrant temporaries per level and runs 7^L small leaves one
// after another; on a GPU that starves the chip (a 4096^3 leaf is 8 workgroups).  Breadth-first
// keeps the 7^L leaf operands and products -- 16.9 GiB at n = 65536, L = 4, intermediate levels never
// materialised -- and turns the whole product into four large launches on one stream.
// Remainders that do not fit the even 2^L split are peeled with direct leaf launches exactly like
// strassen.c:170-204.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

extern "C" {
hipError_t gf2_launch_m4rm_leaf(hipStream_t stream, LeafArgs a, int rg);
hipError_t gf2_launch_m4rm_small(hipStream_t stream, LeafArgs a);
int gf2_m4rm_small_ksplit(int64_t tiles, int64_t wl, int cus, int64_t c_words);
hipError_t gf2_launch_a4_pack_rot(hipStream_t stream, LeafArgs a, word *a4_ws, int rot);
int gf2_winograd_down2_pack_ok(const word *gparent, int64_t p_stride, int64_t p_bs, const word *a4, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_down3(hipStream_t s, int bside, const word *anc, int64_t p_stride, int64_t p_bs, word *gchild,
                                     int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_up3(hipStream_t s, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs,
                                   int64_t nparents, int64_t crows, int64_t cw);
int gf2_winograd_down3_pack_ok(const word *a4, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_down3_pack(hipStream_t s, const word *anc, int64_t p_stride, int64_t p_bs, word *a4,
                                          int64_t nparents, int64_t crows, int64_t cw, int rot);
hipError_t gf2_launch_winograd_down2_pack(hipStream_t s, const word *gparent, int64_t p_stride, int64_t p_bs, word *a4,
                                          int64_t nparents, int64_t crows, int64_t cw, int rot);
hipError_t gf2_launch_winograd_down4(hipStream_t s, int bside, const word *anc, int64_t p_stride, int64_t p_bs, word *gchild,
                                     int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_up4(hipStream_t s, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs,
                                   int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_down4_pack(hipStream_t s, const word *anc, int64_t p_stride, int64_t p_bs, word *a4,
                                          int64_t nparents, int64_t crows, int64_t cw, int rot);
hipError_t gf2_launch_m4rm8q(hipStream_t stream, LeafArgs a, word *a4_ws);
// scheme_passes.hip: 2, 3 or 4 fused levels whose last two are one application of a rank-R scheme for the 4 x 4 x 4 block product
// (R < 49: R, 7 R or R^2 leaves per ancestor instead of 49, 343, 2401)
int gf2_scheme444_rank(void);
int64_t gf2_scheme444_leaves(int levels);
int gf2_scheme444_ok(int levels, int64_t a_rows, int64_t a_cw, int64_t b_rows, int64_t b_cw);
hipError_t gf2_launch_scheme_down(hipStream_t s, int levels, int bside, const word *anc, int64_t p_stride, int64_t p_bs, word *child, int64_t nparents,
                                  int64_t crows, int64_t cw);
hipError_t gf2_launch_scheme_down_pack(hipStream_t s, int levels, const word *anc, int64_t p_stride, int64_t p_bs, word *a4, int64_t nparents,
                                       int64_t crows, int64_t cw);
hipError_t gf2_launch_scheme_up(hipStream_t s, int levels, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs, int64_t nparents,
                                int64_t crows, int64_t cw);
int gf2_m4rm8q_effective_ksplit(int64_t l, int ksplit);
int64_t gf2_m4rm8_a4_words(int64_t m, int64_t l, int64_t batch);
hipError_t gf2_launch_winograd_down(hipStream_t s, int bside, const word *parent, int64_t p_stride,
                                    int64_t p_bs, word *child, int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_up(hipStream_t s, int acc, const word *prod, word *parent, int64_t o_stride,
                                  int64_t o_bs, int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_down2(hipStream_t s, int bside, const word *gparent, int64_t p_stride, int64_t p_bs,
                                     word *gchild, int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_up2(hipStream_t s, int acc, const word *prod, word *gparent, int64_t o_stride,
                                   int64_t o_bs, int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_reduce_partials(hipStream_t s, int acc, word *C, int64_t cs, int64_t cbs, int64_t m, int64_t wn, int64_t tile_rows,
                                      int64_t tw, int64_t tiles_m, int64_t tiles_n, int64_t tile_base, int64_t ntiles, int ks,
                                      const word *Cpart);
hipError_t gf2_launch_zero_tiles(hipStream_t s, word *C, int64_t cs, int64_t cbs, int64_t m, int64_t wn, int64_t tile_rows, int64_t tw,
                                 int64_t tiles_m, int64_t tiles_n, int64_t tile_base, int64_t ntiles);
hipError_t gf2_launch_rowwise(hipStream_t s, int op, word *C, int64_t cs, const word *A, int64_t as,
                              const word *B, int64_t bs, int64_t rows, int64_t w);
hipError_t gf2_launch_xor_masked(hipStream_t s, word *C, int64_t cs, const word *A, int64_t as, const word *B, int64_t bs,
                                 int64_t rows, int64_t ncols);
hipError_t gf2_launch_mask_tail(hipStream_t s, word *M, int64_t stride, int64_t rows, int64_t ncols);
hipError_t gf2_launch_fill_splitmix(hipStream_t s, word *M, int64_t stride, int64_t rows, int64_t ncols, uint64_t seed);
hipError_t gf2_launch_fill_splitmix_rows(hipStream_t s, word *M, int64_t stride, int64_t row0, int64_t rows, int64_t ncols, uint64_t seed);
}

namespace {

constexpr int MAX_LEVELS      = 6;
constexpr int64_t PART_SLABS  = 768;  // slabs (256 KiB each) a split generation-4 launch may use: 192 MiB of the workspace
#ifndef LEAF_SPLIT_COST_BITS
#define LEAF_SPLIT_COST_BITS 256
#endif
#ifndef LEAF_MIN_SPLIT_BITS
#define LEAF_MIN_SPLIT_BITS 512  // fewest inner bits one split of a leaf launch may get (generation 1: atomics)
#endif
#ifndef LEAF_MIN_SPLIT_BITS_G4
#define LEAF_MIN_SPLIT_BITS_G4 128  // the same for generation 4 (slabs + reduce pass)
#endif
int g_max_fuse = 4;  // deepest levels covered by one fused pass each way (1..4); m4ri_amd_set_max_fuse
// Round 4: leaves of 4096 inner bits (was 8192).  The leaf runs on the chip's power limit and the passes do not, so a level more
// trades a power-bound eighth of the leaf for HBM-bound pass time -- a tie while the fourth level cost a separate pass, a win since
// the four-level passes (aux_kernels.hip): 65536^3 28.4 -> 27.7 ms, 32768^3 4.13 -> 3.96, 131072 x 16384 x 131072 37.9 -> 32.5,
// 131072^3 201 -> 195 (profiles/r04_depth_rule_sweep.log)
constexpr int DEFAULT_CUTOFF  = 4096;  // engine default: split while l/2 >= this ...
constexpr int DEFAULT_CUTOFF_M = 4096; // ... and m/2 >= this (one generation-4 tile row)
constexpr int DEFAULT_CUTOFF_N = 4096; // ... and n/2 >= this (8 column tiles)
constexpr int NUM_DEVICES_MAX = 16;

#define HIPTRY(expr)                                                  \
  do {                                                                \
    hipError_t e_ = (expr);                                           \
    if (e_ != hipSuccess) return (int)e_;                             \
  } while (0)

struct Engine {
  int device            = -1;
  word *ws              = nullptr;  // grow-only workspace
  size_t ws_cap         = 0;
  size_t ws_used        = 0;
  word *apk             = nullptr;  // packed-A scratch of the current call (inside ws)
  size_t apk_words      = 0;
  word *part            = nullptr;  // slabs of split leaf launches (inside ws, PART_SLABS tiles)
  word *df_pool         = nullptr;  // quarter-size temporaries of depth-first levels: a stack beside the workspace
  size_t df_cap         = 0, df_used = 0;
  int profiling         = 0;        // 0 off, 1 per call, 2 cumulative over calls (m4ri_amd_set_profiling)
  m4ri_amd_stats stats  = {};
  struct Pending { hipEvent_t e0, e1; long call; };
  std::vector<Pending> pending;     // leaf launches awaiting readout
  long call_seq         = 0;        // products issued so far (tags the pending launches)
  double cum_ms         = 0;        // cumulative mode: leaf time and launches read out so far
  long cum_launches     = 0;
  std::vector<hipEvent_t> event_pool;
  hipStream_t pending_stream = nullptr;
  int cus               = 0;        // compute units of the device (workgroups resident at once: one per CU)
  hipStream_t last_stream = nullptr;  // stream of the previous product: the workspace is shared, so a product
  hipEvent_t last_done    = nullptr;  // on ANOTHER stream first waits for this event (recorded after every product)
  bool have_last          = false;
  std::mutex mu;                      // one host thread at a time plans on this device's workspace; other devices do not wait
#ifdef M4RI_AMD_DEV_EXPERIMENTS
  hipStream_t aux_stream = nullptr;   // second stream of the overlap experiment (developer builds; lives as long as the process)
  hipEvent_t aux_ev[2]   = {nullptr, nullptr};
#endif
};

std::mutex g_cfg_mu;  // the process-wide knobs (workspace budget, fuse depth)
Engine g_engines[NUM_DEVICES_MAX];

// the engine of the calling thread's device, locked: host threads issuing on DIFFERENT devices (the ranks of multi.hip) run
// side by side, threads on one device take turns
struct EngineLock {
  Engine *e = nullptr;
  std::unique_lock<std::mutex> lk;
  EngineLock() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NUM_DEVICES_MAX) return;
    e  = &g_engines[dev];
    lk = std::unique_lock<std::mutex>(e->mu);
    e->device = dev;
    if (e->cus == 0) {
      int n = 0;
      e->cus = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
  }
};

int ws_reserve(Engine *e, size_t words) {
  e->ws_used = 0;
  if (words <= e->ws_cap) return 0;
  HIPTRY(hipDeviceSynchronize());
  if (e->ws) HIPTRY(hipFree(e->ws));
  e->ws = nullptr; e->ws_cap = 0;
  HIPTRY(hipMalloc(reinterpret_cast<void **>(&e->ws), words * sizeof(word)));
  e->ws_cap = words;
  return 0;
}

word *ws_take(Engine *e, size_t words) {
  words = (words + 31) & ~(size_t)31;  // 256-byte granules: keeps every sub-buffer 16-byte aligned
  word *p = e->ws + e->ws_used;
  e->ws_used += words;
  return p;
}

hipEvent_t take_event(Engine *e) {
  if (!e->event_pool.empty()) { hipEvent_t ev = e->event_pool.back(); e->event_pool.pop_back(); return ev; }
  hipEvent_t ev = nullptr;
  if (hipEventCreate(&ev) != hipSuccess) return nullptr;
  return ev;
}

// ---- leaf launch ------------------------------------------------------------------------------
// Two leaf kernels: m4rm8q = generation 4 (4096 x 512 tiles, packed A) and m4rm_leaf = generation 1 (1024-row tiles and
// shorter, plain A).  Generations 2 (k = 7) and 3 (128-byte entries, 2048 x 1024 tiles) were retired in round 2: generation 4
// is at least as fast on every shape from 192 rows on, short tiles included (pick_leaf).
struct LeafKind { int gen; int rg; int rows; double rate; };
const LeafKind LEAF_KINDS[4] = {{4, 32, 4096, 6.7}, {1, 32, 1024, 4.1}, {1, 24, 768, 3.4}, {1, 16, 512, 2.8}};
constexpr int LEAF_KIND_FALLBACK = 1;  // generation 1, 1024 rows: needs no packed A

LeafKind pick_leaf(int64_t m, int64_t l, int64_t n) {
  static const int forced_gen = getenv("M4RI_AMD_LEAF_GEN") ? atoi(getenv("M4RI_AMD_LEAF_GEN")) : 0;  // developer override
  // A few rows against a LARGE B -- the last block of a rows-in-blocks plan, a thin strip -- is generation 1's: its stage is lighter
  // (a partly filled generation-4 tile is bound by the issue of its one gather wave per SIMD: 1.8 us a stage whatever the rows) and
  // with 16384 columns or more its 2048-column tiles and the inner split fill the chip by themselves.  profiles/r04_thin_products_by_leaf_generation.log:
  // 464 x 66000 x 66000 2.08 -> 0.96 ms, 848 x 50000 x 50000 1.10 -> 0.74, 232 x 33000 x 33000 0.58 -> 0.24, 464 x 16384 x 16384 0.143 -> 0.124;
  // 1000 x 16384 x 16384 (0.150 vs 0.188), 464 x 65536 x 4096, 1024 x 1024 x 65536 and everything from ~1300 rows on stay with generation 4.
  if (!forced_gen && n >= 16384 && ((m <= 512 && (double)l * (double)n >= 268435456.0) || (m <= 1024 && (double)l * (double)n >= 1073741824.0))) {
    LeafKind best = LEAF_KINDS[LEAF_KIND_FALLBACK];
    double best_cost = 1e300;
    for (const LeafKind &k : LEAF_KINDS) {
      if (k.gen != 1) continue;
      const double cost = (double)(((m + k.rows - 1) / k.rows) * k.rows) / k.rate;
      if (cost < best_cost) { best_cost = cost; best = k; }
    }
    return best;
  }
  // Generation 4 from 192 rows on, however few of its 4096 tile rows that fills: its waves that own only padding skip their
  // gathers, so a short tile costs the LDS array its real rows, and its 512-column tiles put four times as many
  // workgroups on the chip as the 2048-column tiles of generation 1 (single products: 1024 x 1024 x 65536 70.8 vs 140.7 us,
  // 1100^3 54 vs 124 us, 6000^3 85 vs 152 us; 8800^3 and up: a tie; below ~180 rows generation 1 wins --
  // tools/small_shape_leaf_gens.py).  The throughput model below, calibrated on full batches, is left to pick among the
  // short generation-1 tiles.
  if (!forced_gen && m >= 192) return LEAF_KINDS[0];
  LeafKind best = LEAF_KINDS[LEAF_KIND_FALLBACK];
  double best_cost = 1e300;
  for (const LeafKind &k : LEAF_KINDS) {
    if (forced_gen && k.gen != forced_gen) continue;
    const double padded = (double)(((m + k.rows - 1) / k.rows) * k.rows);
    const double cost   = padded / k.rate;
    if (cost < best_cost) { best_cost = cost; best = k; }
  }
  return best;
}

// Small products take the light one-launch kernel (m4rm_small.hip, "generation 5"): below this much work the three dependent
// launches of generation 4 (pack, split leaf, reduce) cost more than its better tables save.  M4RI_AMD_SMALL_LEAF=0 never, =1 whenever
// the kernel can (tests), unset: the measured rule.
bool small_leaf_wanted(int64_t m, int64_t l, int64_t n, int64_t batch) {
  static const int sw = getenv("M4RI_AMD_SMALL_LEAF") ? atoi(getenv("M4RI_AMD_SMALL_LEAF")) : -1;
  if (sw == 0) return false;
  if (m > INT32_MAX / 2 || l > INT32_MAX / 2 || n > INT32_MAX / 2) return false;
  if ((double)((m + 255) / 256) * (double)((words_of(n) + 7) / 8) * (double)batch * 256.0 > 2147483647.0) return false;  // workgroups of one launch
  if (sw > 0) return true;
  static const double max_work = getenv("M4RI_AMD_SMALL_LEAF_WORK") ? atof(getenv("M4RI_AMD_SMALL_LEAF_WORK")) : 17179869184.0;  // 2^34 bit operations
  return (double)m * (double)l * (double)n * (double)batch <= max_work;
}

// Do the fused bottom `fuse` levels of a product with these LEAF dimensions run through the rank-R scheme of the 4 x 4 x 4 block
// product (scheme_passes.hip)?  ONE rule for the time model (depth_model_seconds) and the schedule (bfs_product): generation 4's
// packed A (the scheme's A-side pass writes it), leaf shapes the scheme kernels take, one packed operand within a buffer descriptor.
bool scheme_applies(int fuse, int64_t leaf_m, int64_t leaf_l, int64_t leaf_n) {
  if (fuse < 2 || leaf_m <= 0 || leaf_l <= 0 || leaf_n <= 0 || leaf_l % 64 != 0 || leaf_n % 64 != 0) return false;
  if (pick_leaf(leaf_m, leaf_l, leaf_n).gen != 4) return false;
  if (gf2_scheme444_ok(fuse, leaf_m, leaf_l / 64, leaf_l, leaf_n / 64) == 0) return false;
  return (uint64_t)gf2_m4rm8_a4_words(leaf_m, leaf_l, 1) * 8 < (1ull << 32);
}

// words of packed-A scratch a leaf launch of this shape may need (max over the packed kernels)
size_t packed_a_words(int64_t m, int64_t l, int64_t batch) {
  return (size_t)gf2_m4rm8_a4_words(m, l, batch);
}

// can a leaf launch of this shape use the packed-A kernel `kind` with the scratch the engine holds?
bool packed_a_fits(const Engine *e, const LeafKind &kind, int64_t m, int64_t l, int64_t batch) {
  if (kind.gen < 4 || batch <= 0) return false;
  const size_t need = (size_t)gf2_m4rm8_a4_words(m, l, batch);
  return e->apk != nullptr && need <= e->apk_words && (uint64_t)need * 8 / (uint64_t)batch < (1ull << 32);
}

// a_prepacked: the engine's packed-A scratch already holds A in the form the picked kernel reads
// (written by the fused down pass); A itself is then not touched.
int launch_leaf_one(Engine *e, hipStream_t st, word *C, int64_t cs, int64_t cbs, const word *A, int64_t as, int64_t abs_,
                    const word *B, int64_t bs, int64_t bbs, int64_t m, int64_t l, int64_t n, int64_t batch,
                    bool add, int ksplit_req, bool a_prepacked) {
  if (m == 0 || n == 0 || batch == 0) return 0;
  if (m > INT32_MAX || l > INT32_MAX || n > INT32_MAX) return (int)hipErrorInvalidValue;
  // 32-bit byte offsets inside one operand (raw buffer addressing)
  if ((uint64_t)m * (uint64_t)as * 8 >= (1ull << 32) || (uint64_t)l * (uint64_t)bs * 8 >= (1ull << 32))
    return (int)hipErrorInvalidValue;
  const int64_t wn   = words_of(n);
  if (!a_prepacked && ksplit_req <= 0 && l > 0 && small_leaf_wanted(m, l, n, batch)) {
    // a small product: ONE launch of the light kernel (m4rm_small.hip) instead of pack + split leaf + reduce
    const int64_t tiles = ((m + 255) / 256) * ((wn + 7) / 8) * batch;
    static const int ks_env = getenv("M4RI_AMD_SMALL_KS") ? atoi(getenv("M4RI_AMD_SMALL_KS")) : 0;  // developer: force the inner split of the small leaf
    const int ks        = ks_env > 0 ? (ks_env < words_of(l) ? ks_env : (int)words_of(l)) : gf2_m4rm_small_ksplit(tiles, words_of(l), e->cus, batch * m * wn);
    if (ks > 1 && !add) {  // the splits meet by atomic XOR: they start from zero
      if (cs == wn && (batch == 1 || cbs == m * wn)) HIPTRY(hipMemsetAsync(C, 0, (size_t)batch * m * wn * 8, st));
      else
        for (int64_t b = 0; b < batch; ++b) HIPTRY(gf2_launch_rowwise(st, 2, C + b * cbs, cs, nullptr, 0, nullptr, 0, m, wn));
    }
    LeafArgs a{};
    a.A = A; a.B = B; a.C = C;
    a.a_stride = as; a.b_stride = bs; a.c_stride = cs;
    a.a_bs = abs_; a.b_bs = bbs; a.c_bs = cbs;
    a.m = (int32_t)m; a.l = (int32_t)l; a.n = (int32_t)n;
    a.batch = (int32_t)batch; a.ksplit = ks;
    a.mode  = (add || ks > 1) ? 1 : 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e->profiling) {
      e0 = take_event(e); e1 = take_event(e);
      if (e0 && e1) HIPTRY(hipEventRecord(e0, st));
    }
    HIPTRY(gf2_launch_m4rm_small(st, a));
    if (e->profiling && e0 && e1) {
      HIPTRY(hipEventRecord(e1, st));
      e->pending.push_back({e0, e1, e->call_seq});
      e->pending_stream = st;
    }
    e->stats.leaf_launches += 1;
    e->stats.leaf_products += batch;
    e->stats.leaf_m = (int32_t)m; e->stats.leaf_l = (int32_t)l; e->stats.leaf_n = (int32_t)n;
    e->stats.leaf_gen = 5;
    e->stats.leaf_bytes += 8.0 * (double)batch * ((double)m * words_of(l) + (double)l * wn + (double)m * wn * (add ? 2 : 1));
    return 0;
  }
  LeafKind kind      = pick_leaf(m, l, n);
  const int64_t tw   = kind.gen == 4 ? 8 : LEAF_TW;  // tile width in words
  const int64_t tiles = ((m + kind.rows - 1) / kind.rows) * ((wn + tw - 1) / tw) * batch;
  const int64_t sbits  = kind.gen == 4 ? 32 : 16;  // inner bits per stage (barrier to barrier)
  const int64_t stages = (l + sbits - 1) / sbits;
  int ksplit = ksplit_req;
  int64_t tail_tiles = 0;  // generation 4, hybrid plan: the last tail_tiles tiles go in a second launch ...
  int tail_ksplit    = 1;  // ... with this inner-dimension split
  if (ksplit <= 0) {
    // One workgroup per CU is resident at a time (LDS), so a launch runs in ceil(workgroups / CUs)
    // rounds and a short last round idles most of the chip: 784 tiles on 256 CUs take 4 rounds for
    // 3.06 rounds of work.  Splitting the inner dimension trades that for more, shorter workgroups
    // (combined by atomic XOR into a zeroed C).  Cost in stages: rounds x (stages per split +
    // prologue/epilogue, + the atomic epilogue when split); at least LEAF_MIN_SPLIT_BITS inner bits
    // per split, and a split only when it pays >= 3 %.
    // per-workgroup overheads in inner bits, calibrated on generation 4 (tools/small_sizes_timing.py):
    // prologue + epilogue ~ 192 bits of stage time, the atomic epilogue of a split ~ 512 more
    // (generation 4 writes slabs instead and pays one reduce pass: LEAF_SPLIT_COST_BITS)
    // Generation 4's slabs cost a fixed part (the reduce pass is one more launch) and a part that grows with the number
    // of slabs written and read back -- measured on single products (tools/leaf_split_sweep.sh): 512^3 42 -> 29 us,
    // 1024^3 50 -> 35 us, 4096^3 71 -> 60 us with splits down to 128 bits, while 512 x 512 x 65536, whose 128 tiles
    // already cover half the chip, must NOT split (42 vs 53 us).
    const double fixed = 192.0 / (double)sbits;
    auto split_cost = [&](int64_t ntiles, int64_t ks) {
      return kind.gen == 4 ? ((double)LEAF_SPLIT_COST_BITS + 2.0 * (double)(ntiles * ks)) / (double)sbits : 512.0 / (double)sbits;
    };
    auto cost = [&](int64_t ks) {
      const int64_t rounds = (tiles * ks + e->cus - 1) / e->cus;
      return (double)rounds * ((double)((stages + ks - 1) / ks) + fixed + (ks > 1 ? split_cost(tiles, ks) : 0.0));
    };
    int64_t cap = stages * sbits / (kind.gen == 4 ? LEAF_MIN_SPLIT_BITS_G4 : LEAF_MIN_SPLIT_BITS);  // inner bits per split: see the constants
    if (cap < 1) cap = 1;
    if (cap > 32) cap = 32;
    ksplit           = 1;
    double best_cost = cost(1) * 0.97;
    for (int64_t ks = 2; ks <= cap; ++ks)
      if (cost(ks) < best_cost) { best_cost = cost(ks); ksplit = (int)ks; }
    // Generation 4 can also run the FULL rounds unsplit and only the tiles of the short last round
    // split (two launches over disjoint tile ranges): the remainder pays for atomics, nobody else
    if (kind.gen == 4 && tiles > e->cus && tiles % e->cus != 0) {
      const int64_t rem = tiles % e->cus, full_rounds = tiles / e->cus;
      for (int64_t ks = 2; ks <= cap; ++ks) {
        const int64_t rounds = (rem * ks + e->cus - 1) / e->cus;
        // (the tail's slabs and their reduce pass once per launch, at half the price per slab of a launch that is split as a whole: with
        // the calibration above 16384^3 -- 392 tiles, 1.53 rounds -- stayed unsplit at 0.673 ms where three splits of its 136-tile tail
        // give 0.632; profiles/r04_leaf_split_model_ab.log holds the table of 48 shapes before and after: nothing else moves)
        const double tail_cost = kind.gen == 4 ? ((double)LEAF_SPLIT_COST_BITS + 1.0 * (double)(rem * ks)) / (double)sbits : split_cost(rem, ks);
        const double c = (double)full_rounds * ((double)stages + fixed) + (double)rounds * ((double)((stages + ks - 1) / ks) + fixed) + tail_cost;
        if (c < best_cost * 0.99) { best_cost = c; tail_tiles = rem; tail_ksplit = (int)ks; }
      }
      if (tail_tiles) ksplit = 1;
      static const int forced_tail = getenv("M4RI_AMD_TAIL_KSPLIT") ? atoi(getenv("M4RI_AMD_TAIL_KSPLIT")) : 0;  // developer: split the short last round this way
      if (forced_tail > 0) { tail_tiles = forced_tail > 1 ? rem : 0; tail_ksplit = forced_tail; ksplit = 1; }
    }
  }
  if (kind.gen == 4) {  // the kernel rounds splits to whole stage pairs: work with the counts it will use
    ksplit      = gf2_m4rm8q_effective_ksplit(l, ksplit);
    tail_ksplit = gf2_m4rm8q_effective_ksplit(l, tail_ksplit);
  }
  // generation 4 combines the splits through per-split slabs + one reduce pass when they fit the
  // workspace's slab region (no atomics, no zeroing); otherwise, and in the older kernels, by atomic
  // XOR into a zeroed C
  const bool slabs = kind.gen == 4 && e->part != nullptr && packed_a_fits(e, kind, m, l, batch) &&
                     (tail_tiles > 0 ? tail_tiles * tail_ksplit : (ksplit > 1 ? tiles * ksplit : PART_SLABS + 1)) <= PART_SLABS;
  if (l == 0 || (ksplit > 1 && !add && !slabs)) {  // empty inner dimension, or atomics need a zeroed C
    if (!add) {
      if (cs == wn && (batch == 1 || cbs == m * wn))  // one contiguous block
        HIPTRY(hipMemsetAsync(C, 0, (size_t)batch * m * wn * 8, st));
      else
        for (int64_t b = 0; b < batch; ++b)
          HIPTRY(gf2_launch_rowwise(st, 2, C + b * cbs, cs, nullptr, 0, nullptr, 0, m, wn));
    }
    if (l == 0) return 0;
  }
  LeafArgs a{};
  a.A = A; a.B = B; a.C = C;
  a.a_stride = as; a.b_stride = bs; a.c_stride = cs;
  a.a_bs = abs_; a.b_bs = bbs; a.c_bs = cbs;
  a.m = (int32_t)m; a.l = (int32_t)l; a.n = (int32_t)n;
  a.batch = (int32_t)batch; a.ksplit = ksplit;
  a.mode  = (ksplit > 1 && slabs) ? 2 : (add || ksplit > 1) ? 1 : 0;
  a.Cpart = e->part;
  // generation 4 consumes A in a packed, chunk-major form (one streaming pass into the call's
  // scratch first); it needs that scratch and 32-bit offsets inside one packed operand
  if (a_prepacked) {
    if (kind.gen < 4 || !packed_a_fits(e, kind, m, l, batch)) return (int)hipErrorInvalidValue;  // caller checked
  } else if (kind.gen == 4) {
    if (!packed_a_fits(e, kind, m, l, batch)) kind = LEAF_KINDS[LEAF_KIND_FALLBACK];
    else {
      const size_t need = (size_t)gf2_m4rm8_a4_words(m, l, batch);
      HIPTRY(gf2_launch_a4_pack_rot(st, a, e->apk, 1));  // the index bytes pre-rotated for the leaf's constant selectors
      e->stats.aux_bytes += 8.0 * (double)batch * (double)m * words_of(l) + 8.0 * (double)need;
    }
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e->profiling) {  // the events bracket the leaf kernel alone
    e0 = take_event(e); e1 = take_event(e);
    if (e0 && e1) HIPTRY(hipEventRecord(e0, st));
  }
#ifdef M4RI_AMD_DEV_EXPERIMENTS  // developer builds only (profiles/r05_overlap_power/): the batch in this many launches
  static const int exp_groups = getenv("M4RI_AMD_LEAF_GROUPS") ? atoi(getenv("M4RI_AMD_LEAF_GROUPS")) : 0;
#else
  constexpr int exp_groups = 0;
#endif
  if (kind.gen == 4 && exp_groups > 1 && batch % exp_groups == 0 && ksplit == 1) {
    for (int g = 0; g < exp_groups; ++g) {
      LeafArgs part = a;
      part.tile_base = (tiles / exp_groups) * g; part.tile_count = tiles / exp_groups;
      HIPTRY(gf2_launch_m4rm8q(st, part, e->apk));
    }
  } else if (kind.gen == 4 && tail_tiles > 0) {
    LeafArgs head = a, tail = a;
    head.tile_base = 0; head.tile_count = tiles - tail_tiles;
    tail.tile_base = tiles - tail_tiles; tail.tile_count = tail_tiles;
    tail.ksplit = tail_ksplit; tail.mode = slabs ? 2 : 1;
    const int64_t tm = (m + kind.rows - 1) / kind.rows, tn = (wn + tw - 1) / tw;
    if (!add && !slabs)  // the split tiles are combined by atomic XOR: they start from zero
      HIPTRY(gf2_launch_zero_tiles(st, C, cs, cbs, m, wn, kind.rows, tw, tm, tn, tail.tile_base, tail_tiles));
    HIPTRY(gf2_launch_m4rm8q(st, head, e->apk));
    HIPTRY(gf2_launch_m4rm8q(st, tail, e->apk));
    if (slabs) HIPTRY(gf2_launch_reduce_partials(st, add ? 1 : 0, C, cs, cbs, m, wn, kind.rows, tw, tm, tn, tail.tile_base, tail_tiles, tail_ksplit, e->part));
  } else if (kind.gen == 4) {
    HIPTRY(gf2_launch_m4rm8q(st, a, e->apk));
    if (a.mode == 2)
      HIPTRY(gf2_launch_reduce_partials(st, add ? 1 : 0, C, cs, cbs, m, wn, kind.rows, tw, (m + kind.rows - 1) / kind.rows, (wn + tw - 1) / tw,
                                        0, tiles, ksplit, e->part));
  }
  else HIPTRY(gf2_launch_m4rm_leaf(st, a, kind.rg));
  if (e->profiling && e0 && e1) {
    HIPTRY(hipEventRecord(e1, st));
    e->pending.push_back({e0, e1, e->call_seq});
    e->pending_stream = st;
  }
  e->stats.leaf_launches += 1;
  e->stats.leaf_products += batch;
  e->stats.leaf_m = (int32_t)m; e->stats.leaf_l = (int32_t)l; e->stats.leaf_n = (int32_t)n;
  e->stats.leaf_gen = kind.gen;
  e->stats.leaf_bytes += 8.0 * (double)batch * ((double)m * words_of(l) + (double)l * wn + (double)m * wn * (add ? 2 : 1));
  return 0;
}

// The kernels address one operand of one product through a raw buffer descriptor: 32-bit byte
// offsets, so A (m rows) and B (l rows) of a single product must each stay below 4 GiB.  Strassen
// leaves always do; a direct product on huge operands (a remainder strip against a 262144-row A, an
// unsplit 8 GiB matrix) is cut into row chunks of A/C and inner-dimension chunks of A/B here.
int launch_leaf(Engine *e, hipStream_t st, word *C, int64_t cs, int64_t cbs, const word *A, int64_t as, int64_t abs_,
                const word *B, int64_t bs, int64_t bbs, int64_t m, int64_t l, int64_t n, int64_t batch,
                bool add, int ksplit_req, bool a_prepacked = false) {
  constexpr uint64_t LIMIT = (1ull << 32) - (1ull << 20);
  if (batch == 1 && !a_prepacked) {
    if ((uint64_t)m * (uint64_t)as * 8 >= LIMIT && m > 1) {  // rows of A and C: whole tiles when the stride allows
      int64_t m1 = (int64_t)(LIMIT / ((uint64_t)as * 8));     // (a window of a very wide parent may allow fewer)
      m1 = m1 >= 4096 ? m1 / 4096 * 4096 : m1 >= 32 ? m1 / 32 * 32 : m1;
      if (m1 < 1) return (int)hipErrorInvalidValue;  // one row of A beyond 4 GiB
      for (int64_t r0 = 0; r0 < m; r0 += m1) {
        const int64_t mr = (m - r0) < m1 ? (m - r0) : m1;
        if (int rc = launch_leaf(e, st, C + r0 * cs, cs, 0, A + r0 * as, as, 0, B, bs, 0, mr, l, n, 1, add, ksplit_req)) return rc;
      }
      return 0;
    }
    if ((uint64_t)l * (uint64_t)bs * 8 >= LIMIT && l > 64) {  // inner dimension: word-aligned slabs, later ones accumulate
      int64_t l1 = (int64_t)(LIMIT / ((uint64_t)bs * 8));
      l1 = l1 >= 4096 ? l1 / 4096 * 4096 : l1 / 64 * 64;
      if (l1 < 64) return (int)hipErrorInvalidValue;  // 64 rows of B beyond 4 GiB
      for (int64_t k0 = 0; k0 < l; k0 += l1) {
        const int64_t lk = (l - k0) < l1 ? (l - k0) : l1;
        if (int rc = launch_leaf(e, st, C, cs, 0, A + k0 / 64, as, 0, B + k0 * bs, bs, 0, m, lk, n, 1, add || k0 > 0, ksplit_req)) return rc;
      }
      return 0;
    }
  }
  return launch_leaf_one(e, st, C, cs, cbs, A, as, abs_, B, bs, bbs, m, l, n, batch, add, ksplit_req, a_prepacked);
}

int reserve_apk(Engine *e, size_t words) {
  words = (words + 31) & ~(size_t)31;
  if (int rc = ws_reserve(e, words + (size_t)PART_SLABS * LEAF_PART_WORDS)) return rc;
  e->apk = ws_take(e, words);
  e->apk_words = words;
  e->part = ws_take(e, (size_t)PART_SLABS * LEAF_PART_WORDS);
  return 0;
}

// ---- level planning ----------------------------------------------------------------------------
int64_t ipow7(int d) { int64_t r = 1; while (d-- > 0) r *= 7; return r; }
bool closer(int64_t a, int64_t cutoff) { return 3 * a < 4 * cutoff; }  // strassen.c:39

// ---- the engine's own depth: a time model -------------------------------------------------------------------------------
// Rounds 1 - 4a split "while every half keeps 4096 rows, inner bits and columns".  That rule is right for cubes and wrong by up to
// 13 % where one dimension is short and the others long (131072 x 8192 x 131072 wants leaves of 1024 inner bits: 18.3 against
// 20.3 ms), and it knows nothing of what the strips of a ragged shape cost.  The depth is now the minimum of a small model of the
// schedule's time, checked against every depth of 64 shapes (tools/depth_model_sweep.py, profiles/r04_depth_model_sweep.log,
// r04_depth_model_validation.log, r04_row_blocks_sweep*.log: on the first 27 the sum of the regrets against the best measured depth
// fell from 81 % to 11 %; over all 64 the plan is within 2.7 % of the best alternative known, a 12288^3 excepted at 7.9 %):
//   leaf launch  7^L products in tiles of 4096 rows x 512 columns (a partly filled tile costs a whole one), 256 tiles per round,
//                a tile takes (inner bits / 32 + 4.5) stages of 2.43 us; a last partial round costs its fill + 0.05 + 20 stages
//                (it runs split); launches of at most one round run split all over the chip at 1.08 of ideal + 45 us
//   passes       bytes of the fused / single passes of bfs_product at 5.3 TB/s + 4 us per launch; a four-level up pass whose leaf
//                rows are not a multiple of 32 words falls back to atomics (the children written once more)
//   strips       the three thin products around the even block, the inner one as a read-modify-write of C
// The constants are one box's; only the ORDER of the depths matters, and that is set by ratios that move together.
// Pure host arithmetic (m4ri_amd_plan_levels; tests/test_host_logic.py pins the table).
double depth_model_seconds(int64_t m, int64_t l, int64_t n, int L, double batch = 1.0) {  // `batch` products of this shape scheduled as one
  // (an unsplit product has no strips: the leaf takes any l and n as they are)
  const int64_t mm = m >> L, ll = L ? (l / (64ll << L)) * 64 : l, nn = L ? (n / (64ll << L)) * 64 : n;
  if (mm == 0 || ll == 0 || nn == 0) return 1e30;
  constexpr double UNIT = 2.43e-6, FIXED = 4.5, BW = 5.3e12, LAUNCH = 4e-6, CUS = 256.0;
  auto leaf = [&](int64_t pm, int64_t pl, int64_t pn, double count) {  // `count` products of one shape in one launch
    if (pm <= 0 || pl <= 0 || pn <= 0) return 0.0;
    const double ln = (double)pl * (double)pn;
    if (count == 1 && pn >= 16384 && ((pm <= 512 && ln >= 268435456.0) || (pm <= 1024 && ln >= 1073741824.0)))
      return 45e-6 + ln * (pm <= 512 ? 2.1e-13 : 3.2e-13);  // a few rows against a large B: generation 1 (pick_leaf)
    const double tiles = count * (double)((pm + 4095) / 4096) * (double)((words_of(pn) + 7) / 8);
    const double units = (double)((pl + 31) / 32) + FIXED;
    if (tiles > CUS) {
      const double full = std::floor(tiles / CUS), fr = tiles / CUS - full;
      return (full + (fr > 0 ? std::min(1.0, fr + 0.05 + 20.0 / units) : 0.0)) * units * UNIT + 20e-6;
    }
    return tiles * units / CUS * UNIT * 1.08 + 45e-6;  // (+ the pack, the split's reduce pass)
  };
  double p7 = batch;
  for (int d = 0; d < L; ++d) p7 *= 7;
  // the fused bottom levels run through the rank-R 4 x 4 x 4 scheme where the leaves allow it: R, 7 R or R^2 products instead of 7^2, 7^3, 7^4
  const int mfuse = L < g_max_fuse ? L : g_max_fuse;
  const bool scheme = scheme_applies(mfuse, mm, ll, nn);
  const double srat = scheme ? (double)gf2_scheme444_leaves(mfuse) / (double)ipow7(mfuse) : 1.0;
  p7 *= srat;
  double t = leaf(mm, ll, nn, p7);
  // passes over the even block
  const int64_t me = mm << L, le = ll << L, ne = nn << L;
  const double sa = 8.0 * (double)me * (double)words_of(le), sb = 8.0 * (double)le * (double)words_of(ne), sc = 8.0 * (double)me * (double)words_of(ne);
  const int fuse = L < g_max_fuse ? L : (g_max_fuse > 0 ? g_max_fuse : 1);
  auto r = [](int d) { double x = 1; while (d-- > 0) x *= 1.75; return x; };
  double factor = 0;
  for (int d = 0; d < L - fuse; ++d) factor += r(d) + r(d + 1);
  if (L > 0) factor += r(L - fuse) + r(L) * srat;
  double bytes = (sa + sb + sc) * factor * batch;
  if (L < 2 || (mm % 32) != 0 || (words_of(ll) % 16) != 0) bytes += 2.0 * sa * r(L) * batch;  // the separate pack pass of A: no fused form below two levels or for such leaves
  if (fuse == 4 && (words_of(nn) % 32) != 0) bytes += sc * (r(L) + 1) * batch;  // atomic up pass: zeroed output, children folded by read-modify-write
  const int launches = 3 * ((L - fuse > 0 ? L - fuse : 0) + (L > 0 ? 1 : 0)) + 1;
  t += bytes / BW + launches * LAUNCH;
  // strips of a ragged shape: columns beyond the even block, inner bits beyond it (C read and written once more), rows beyond it
  const int64_t rm = m - me, rl = l - le, rn = n - ne;
  if (rn > 0) t += leaf(me, le, rn, 1) + 2.0 * sa / BW + 15e-6;
  if (rl > 0) t += leaf(me, rl, n, 1) + 2.0 * 8.0 * (double)me * (double)words_of(n) / BW + 15e-6;
  if (rm > 0) t += leaf(rm, l, n, 1) + 15e-6;
  return t;
}

// depth of ONE product of these dimensions by the model (cutoff == 0)
int model_levels(int64_t m, int64_t l, int64_t n, double *seconds, double batch = 1.0) {
  int L = 0;
  // every depth whose leaves keep a whole tile of rows (half-filled tiles cost whole ones: 16384 x 65536 x 65536 takes 8.6 ms with
  // leaves of 4096 rows and 12.3 ms with 2048); a deeper one has to win by 1 %
  // ... and 1024 inner bits and columns: below that no shape of the sweeps gained (65536 x 4096 x 65536 with leaves of 512 inner
  // bits: 3.02 against 2.67 ms), and the model is not trusted where per-launch constants decide
  double best = depth_model_seconds(m, l, n, 0, batch);
  for (int d = 1; d <= MAX_LEVELS && (m >> d) >= DEFAULT_CUTOFF_M && (l >> d) >= 1024 && (n >> d) >= 1024; ++d) {
    const double t = depth_model_seconds(m, l, n, d, batch);
    if (t < 0.99 * best) { best = t; L = d; }
  }
  if (seconds) *seconds = best;
  return L;
}

// Rows in blocks.  The leaf's tile is 4096 rows, so a depth is only worth its passes when m / 2^L is (close to) a multiple of 4096:
// 65664 = 65536 + 128 rows ran ONE level (39.7 ms against 27.5 for 65536^3), 36864 = 32768 + 4096 one level (7.1 ms).  Rows, unlike
// inner bits and columns, can be cut anywhere without a reduction: C's rows [r0, r1) = A's rows [r0, r1) * B.  So the engine cuts
// m into blocks of k * 4096 * 2^L rows, largest first, each at its own depth (the rest recursively), when the model says the sum is
// 8 % cheaper than the best single product; B's down pass is repeated per block, which the model counts.
struct RowBlock { int64_t rows; int levels; };
double plan_row_blocks(int64_t m, int64_t l, int64_t n, std::vector<RowBlock> &out, int depth = 0) {
  double t1 = 0;
  const int L1 = model_levels(m, l, n, &t1);
  std::vector<RowBlock> best{{m, L1}};
  double best_t = t1;
  static const bool off = getenv("M4RI_AMD_ROW_BLOCKS") && getenv("M4RI_AMD_ROW_BLOCKS")[0] == '0';  // developer switch: one product
  if (!off && depth < 6) {
    for (int d = 1; d <= MAX_LEVELS; ++d) {
      const int64_t unit = (int64_t)DEFAULT_CUTOFF_M << d, k = m / unit;
      if (k < 1) break;
      const int64_t rows = k * unit;
      if (rows == m) continue;
      double tb = 0;
      const int Lb = model_levels(rows, l, n, &tb);
      if (Lb < d) continue;  // the block would not use the depth it was cut for
      std::vector<RowBlock> rest;
      const double t = tb + plan_row_blocks(m - rows, l, n, rest, depth + 1);
      if (t < 0.92 * best_t) {  // 8 %: on ragged shapes the model flatters the blocks by up to 8 points (profiles/r04_row_blocks_sweep.log)
        best_t = t;
        best.assign(1, RowBlock{rows, Lb});
        best.insert(best.end(), rest.begin(), rest.end());
      }
    }
  }
  out = best;
  return best_t;
}

int plan_levels(int64_t m, int64_t l, int64_t n, int cutoff) {
  int L = 0;
  if (cutoff == 0) {
    std::vector<RowBlock> blocks;  // the depth of the first (largest) block of rows: what the stats of a call report
    plan_row_blocks(m, l, n, blocks);
    L = blocks.empty() ? 0 : blocks[0].levels;
  } else {
    // the reference's rule: recurse until one dimension is "closer to cutoff than to its half"
    int64_t a = m, b = l, c = n;
    while (L < MAX_LEVELS && !(closer(a, cutoff) || closer(b, cutoff) || closer(c, cutoff))) { a /= 2; b /= 2; c /= 2; ++L; }
  }
  if (const char *f = getenv("M4RI_AMD_LEVELS")) {  // developer override (tools/depth_model_sweep.py): this many levels, if the shape has them
    const int want = atoi(f);
    if (want >= 0 && want <= MAX_LEVELS) L = want;
  }
  // every level halves l and n on word boundaries and m on rows: need a non-empty even block
  while (L > 0 && ((m >> L) == 0 || (l / (64ll << L)) == 0 || (n / (64ll << L)) == 0)) --L;
  return L;
}


// breadth-first Strassen-Winograd on the even block: C (m x n) (+)= A (m x l) * B (l x n), with
// m % 2^L == 0 and l, n % (64 * 2^L) == 0.
// `batch` > 1: that many independent products of one shape, X_b = X + b * x_bs words, through the SAME launches (every pass takes
// a count of parents, the leaf launch a count of products): what fills the chip when one product's leaves do not.
int bfs_product(Engine *e, hipStream_t st, DMat C, DMat A, DMat B, bool add, int L, size_t a7_extra, int64_t batch = 1,
                int64_t c_bs = 0, int64_t a_bs = 0, int64_t b_bs = 0) {
  const int64_t m = A.nrows, l = A.ncols, n = B.ncols;
  // The deepest levels are done by ONE fused pass each way: up to four of them (g_max_fuse), whose
  // intermediate levels are never materialised -- that saves their buffers and a write + a read of
  // the 7/4-times-larger operands per skipped level.  Levels above go one at a time.
  const int fuse = L < g_max_fuse ? L : g_max_fuse;            // levels covered by the bottom pass
  auto materialised = [&](int d) { return d <= L - fuse || d == L; };
  // With a fused last pass and a leaf that reads packed A, the pass writes the packed form itself:
  // the row-major A operands of the leaves are never materialised and the pack pass disappears.
  const LeafKind leaf_kind = pick_leaf(m >> L, (l / (64ll << L)) * 64, (n / (64ll << L)) * 64);  // (the dimensions launch_leaf_one will see)
  // Four fused levels as TWO applications of the rank-R scheme for the 4 x 4 x 4 block product (scheme_passes.hip): R^2 leaves per
  // ancestor of the fused pass instead of 7^4, packed A written by the pass itself.  Needs generation 4's packed A and leaf shapes the
  // scheme kernels take; everything else keeps the Winograd passes.
  const int64_t leaf_m = m >> L, leaf_l = l >> L, leaf_n = n >> L;
  const bool scheme = scheme_applies(fuse, leaf_m, leaf_l, leaf_n);
  const int64_t leaves1 = scheme ? ipow7(L - fuse) * gf2_scheme444_leaves(fuse) : ipow7(L);  // leaf products of ONE product
  const int64_t leaves  = batch * leaves1;                                                      // products of the leaf launch
  bool prepack = scheme;
  if (!scheme && fuse >= 2 && leaf_kind.gen == 4) {
    static const word aligned16[2] __attribute__((aligned(16))) = {0, 0};
    const int d0      = L - fuse;
    const word *pa    = d0 == 0 ? A.p : aligned16;  // deeper levels live in the 256-byte aligned workspace
    const int64_t pas = d0 == 0 ? A.stride : (l >> d0) / 64;
    const uint64_t a4_bytes = (uint64_t)gf2_m4rm8_a4_words(m >> L, l >> L, 1) * 8;  // one packed operand: 32-bit offsets
    prepack = a4_bytes < (1ull << 32) &&
              (fuse >= 3 ? gf2_winograd_down3_pack_ok(aligned16, m >> L, (l >> L) / 64) != 0
                         : gf2_winograd_down2_pack_ok(pa, pas, d0 == 0 ? a_bs : (m >> d0) * pas, aligned16, m >> L, (l >> L) / 64) != 0);
  }
  // workspace plan
  size_t need = 0;
  auto pad = [](size_t w) { return (w + 31) & ~(size_t)31; };
  for (int d = 1; d <= L; ++d) {
    if (!materialised(d)) continue;
    const int64_t md = m >> d, wl = (l >> d) / 64, wnn = (n >> d) / 64, cnt = d == L ? leaves : batch * ipow7(d);
    need += (prepack && d == L ? 0 : pad((size_t)cnt * md * wl)) + pad((size_t)cnt * (l >> d) * wnn) + pad((size_t)cnt * md * wnn);
  }
  // C += A*B through a three-level up pass: the pass writes a temporary and one XOR pass folds it into
  // C (the accumulating three-level kernel needs > 256 registers per lane and runs at a quarter of the rate)
  const bool acc_via_tmp = add && fuse == 3 && L == 3 && !scheme;
  if (acc_via_tmp && batch > 1) return (int)hipErrorInvalidValue;  // (engine_mul_batch sends such a batch one product at a time)
  if (acc_via_tmp) need += pad((size_t)m * (n / 64));
  {
    const size_t a7_bfs = packed_a_words(m >> L, l >> L, leaves);
    if (a7_bfs > a7_extra) a7_extra = a7_bfs;
  }
  a7_extra = pad(a7_extra);
  if (int rc = ws_reserve(e, need + a7_extra + (size_t)PART_SLABS * LEAF_PART_WORDS)) return rc;
  e->apk = ws_take(e, a7_extra);
  e->apk_words = a7_extra;
  e->part = ws_take(e, (size_t)PART_SLABS * LEAF_PART_WORDS);
  std::vector<word *> Al(L + 1, nullptr), Bl(L + 1, nullptr), Pl(L + 1, nullptr);
  if (prepack && !packed_a_fits(e, leaf_kind, m >> L, l >> L, leaves)) return (int)hipErrorInvalidValue;  // cannot happen: a7_extra covers it
  for (int d = 1; d <= L; ++d) {
    if (!materialised(d)) continue;
    const int64_t md = m >> d, wl = (l >> d) / 64, wnn = (n >> d) / 64, cnt = d == L ? leaves : batch * ipow7(d);
    if (!(prepack && d == L)) Al[d] = ws_take(e, (size_t)cnt * md * wl);
    Bl[d] = ws_take(e, (size_t)cnt * (l >> d) * wnn);
    Pl[d] = ws_take(e, (size_t)cnt * md * wnn);
  }
  word *acc_tmp = acc_via_tmp ? ws_take(e, (size_t)m * (n / 64)) : nullptr;
  e->stats.workspace_bytes = (double)e->ws_cap * 8.0;
  // down passes: level d -> d+1, and d -> L for the fused pass at the bottom
  for (int d = 0; d < L;) {
    const int step    = d == L - fuse ? fuse : 1;
    const int64_t cnt = batch * ipow7(d);
    const int64_t cm = m >> (d + step), cl = l >> (d + step), cn = n >> (d + step);
    const word *pa = d == 0 ? A.p : Al[d];
    const int64_t pas = d == 0 ? A.stride : (l >> d) / 64, pabs = d == 0 ? a_bs : (m >> d) * pas;
    const word *pb = d == 0 ? B.p : Bl[d];
    const int64_t pbs = d == 0 ? B.stride : (n >> d) / 64, pbbs = d == 0 ? b_bs : (l >> d) * pbs;
    const int rot = leaf_kind.gen == 4 ? 1 : 0;  // the leaf's pre-rotated index bytes (pack mode)
    if (scheme && step == fuse && step >= 2) {  // the fused bottom levels through the 4 x 4 x 4 scheme, one pass each way (scheme_passes.hip)
      const double rr = (double)gf2_scheme444_leaves(step), blocks = (double)(1 << (2 * step));
      HIPTRY(gf2_launch_scheme_down_pack(st, step, pa, pas, pabs, e->apk, cnt, cm, cl / 64));
      HIPTRY(gf2_launch_scheme_down(st, step, 1, pb, pbs, pbbs, Bl[d + step], cnt, cl, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * (blocks + rr) * ((double)cm * (cl / 64) + (double)cl * (cn / 64));
    } else if (step == 4) {  // four levels in one pass each way: the top one formed on the fly (aux_kernels.hip)
      if (prepack) HIPTRY(gf2_launch_winograd_down4_pack(st, pa, pas, pabs, e->apk, cnt, cm, cl / 64, rot));
      else HIPTRY(gf2_launch_winograd_down4(st, 0, pa, pas, pabs, Al[d + 4], cnt, cm, cl / 64));
      HIPTRY(gf2_launch_winograd_down4(st, 1, pb, pbs, pbbs, Bl[d + 4], cnt, cl, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * 2657.0 * ((double)cm * (cl / 64) + (double)cl * (cn / 64));  // 256 in + 2401 out
    } else if (step == 3) {
      if (prepack) HIPTRY(gf2_launch_winograd_down3_pack(st, pa, pas, pabs, e->apk, cnt, cm, cl / 64, rot));
      else HIPTRY(gf2_launch_winograd_down3(st, 0, pa, pas, pabs, Al[d + 3], cnt, cm, cl / 64));
      HIPTRY(gf2_launch_winograd_down3(st, 1, pb, pbs, pbbs, Bl[d + 3], cnt, cl, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * 407.0 * ((double)cm * (cl / 64) + (double)cl * (cn / 64));  // 64 in + 343 out
    } else if (step == 2) {
      if (prepack) HIPTRY(gf2_launch_winograd_down2_pack(st, pa, pas, pabs, e->apk, cnt, cm, cl / 64, rot));
      else HIPTRY(gf2_launch_winograd_down2(st, 0, pa, pas, pabs, Al[d + 2], cnt, cm, cl / 64));
      HIPTRY(gf2_launch_winograd_down2(st, 1, pb, pbs, pbbs, Bl[d + 2], cnt, cl, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * 65.0 * ((double)cm * (cl / 64) + (double)cl * (cn / 64));  // 16 in + 49 out
    } else {
      HIPTRY(gf2_launch_winograd_down(st, 0, pa, pas, pabs, Al[d + 1], cnt, cm, cl / 64));
      HIPTRY(gf2_launch_winograd_down(st, 1, pb, pbs, pbbs, Bl[d + 1], cnt, cl, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * 11.0 * ((double)cm * (cl / 64) + (double)cl * (cn / 64));  // 4 in + 7 out
    }
    d += step;
  }
#ifdef M4RI_AMD_DEV_EXPERIMENTS
  // developer experiment, NOT in the product build (profiles/r05_overlap_power/; build with -DM4RI_AMD_DEV_EXPERIMENTS): the three
  // four-level passes once more on a second stream UNDER the leaf launch (same sources, same destinations, same values; the up pass
  // reads half-written products and its output is overwritten by the real one -- unsafe by construction, correct only because the
  // real up pass waits for it) -- what the leaf loses to HBM-bound work beside it bounds what a pipelined schedule could win
  static const int exp_overlap = getenv("M4RI_AMD_OVERLAP_EXP") ? atoi(getenv("M4RI_AMD_OVERLAP_EXP")) : 0;
  const bool overlap_now = exp_overlap && L == 4 && fuse == 4 && prepack && !scheme;
  if (overlap_now) {
    if (!e->aux_stream) { HIPTRY(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking)); HIPTRY(hipEventCreateWithFlags(&e->aux_ev[0], hipEventDisableTiming)); HIPTRY(hipEventCreateWithFlags(&e->aux_ev[1], hipEventDisableTiming)); }
    HIPTRY(hipEventRecord(e->aux_ev[0], st));
    HIPTRY(hipStreamWaitEvent(e->aux_stream, e->aux_ev[0], 0));
    const int64_t cm = m >> 4, cl = l >> 4, cn = n >> 4;
    if (exp_overlap & 1) HIPTRY(gf2_launch_winograd_down4_pack(e->aux_stream, A.p, A.stride, 0, e->apk, 1, cm, cl / 64, 1));
    if (exp_overlap & 2) HIPTRY(gf2_launch_winograd_down4(e->aux_stream, 1, B.p, B.stride, 0, Bl[4], 1, cl, cn / 64));
    if (exp_overlap & 4) HIPTRY(gf2_launch_winograd_up4(e->aux_stream, 0, Pl[4], C.p, C.stride, 0, 1, cm, cn / 64));
    HIPTRY(hipEventRecord(e->aux_ev[1], e->aux_stream));
  }
#endif
  // all 7^L leaf products in one launch
  {
    const int64_t lm = m >> L, ll = l >> L, ln = n >> L, cnt = leaves;
    if (int rc = launch_leaf(e, st, Pl[L], ln / 64, lm * (ln / 64), Al[L], ll / 64, lm * (ll / 64), Bl[L], ln / 64,
                             ll * (ln / 64), lm, ll, ln, cnt, false, 0, prepack))
      return rc;
  }
#ifdef M4RI_AMD_DEV_EXPERIMENTS
  if (overlap_now) HIPTRY(hipStreamWaitEvent(st, e->aux_ev[1], 0));
#endif
  // up passes: the fused pass at the bottom first (L -> L - fuse), then level d+1 -> d
  for (int d = L; d > 0;) {
    const int step    = d == L ? fuse : 1;
    const int dst     = d - step;
    const int64_t cnt = batch * ipow7(dst);
    const int64_t cm = m >> d, cn = n >> d;
    word *out          = dst == 0 ? C.p : Pl[dst];
    const int64_t ostr = dst == 0 ? C.stride : (n >> dst) / 64;
    const int64_t obs  = dst == 0 ? c_bs : (m >> dst) * ostr;
    const int acc      = (dst == 0 && add) ? 1 : 0;
    if (scheme && step == fuse && step >= 2) {
      const double rr = (double)gf2_scheme444_leaves(step), blocks = (double)(1 << (2 * step));
      HIPTRY(gf2_launch_scheme_up(st, step, acc, Pl[d], out, ostr, obs, cnt, cm, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * (double)cm * (cn / 64) * (rr + (acc ? 2.0 : 1.0) * blocks);
    } else if (step == 4) {
      HIPTRY(gf2_launch_winograd_up4(st, acc, Pl[d], out, ostr, obs, cnt, cm, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * (double)cm * (cn / 64) * (2401.0 + 512.0 + (acc ? 0.0 : 256.0));  // products in, every word read + written once, the clear
    } else if (step == 3 && acc && acc_tmp) {
      HIPTRY(gf2_launch_winograd_up3(st, 0, Pl[d], acc_tmp, n / 64, 0, cnt, cm, cn / 64));
      HIPTRY(gf2_launch_rowwise(st, 0, C.p, C.stride, C.p, C.stride, acc_tmp, n / 64, m, n / 64));
      e->stats.aux_bytes += 8.0 * cnt * (double)cm * (cn / 64) * 407.0 + 8.0 * 3.0 * (double)m * (n / 64);
    } else if (step == 3) {
      HIPTRY(gf2_launch_winograd_up3(st, acc, Pl[d], out, ostr, obs, cnt, cm, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * (double)cm * (cn / 64) * (acc ? 471.0 : 407.0);
    } else if (step == 2) {
      HIPTRY(gf2_launch_winograd_up2(st, acc, Pl[d], out, ostr, obs, cnt, cm, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * (double)cm * (cn / 64) * (acc ? 81.0 : 65.0);
    } else {
      HIPTRY(gf2_launch_winograd_up(st, acc, Pl[d], out, ostr, obs, cnt, cm, cn / 64));
      e->stats.aux_bytes += 8.0 * cnt * (double)cm * (cn / 64) * (acc ? 15.0 : 11.0);
    }
    d = dst;
  }
  return 0;
}

// upper bound of the words bfs_product reserves for an L-level schedule (level-L operands and products,
// packed A, the materialised upper levels, slabs, the accumulate temporary)
size_t bfs_words_bound(int64_t m, int64_t l, int64_t n, int L) {
  const int fuse = L < g_max_fuse ? L : g_max_fuse;
  size_t w = 0;
  for (int d = 1; d <= L; ++d) {
    if (!(d <= L - fuse || d == L)) continue;
    const size_t md = (size_t)(m >> d), wl = (size_t)((l >> d) / 64), ld = (size_t)(l >> d), wn = (size_t)((n >> d) / 64);
    w += (size_t)ipow7(d) * (md * wl + ld * wn + md * wn) + 96;
  }
  w += packed_a_words(m >> L, l >> L, ipow7(L)) + packed_a_words(m, l, 1) + (size_t)PART_SLABS * LEAF_PART_WORDS;
  w += (size_t)m * (size_t)(n / 64 + 1);
  return w;
}

int64_t g_ws_budget = 0;  // bytes the breadth-first workspace may take; 0 = what the device has left (m4ri_amd_set_workspace_budget)

double workspace_budget(const Engine *e) {
  if (g_ws_budget > 0) return (double)g_ws_budget;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 1e30;
  return 0.92 * ((double)free_b + (double)e->ws_cap * 8.0);  // the workspace we already hold counts as free
}

// the breadth-first product with its remainder strips (strassen.c:170-204): L levels fit the memory
int bfs_with_strips(Engine *e, hipStream_t st, DMat C, DMat A, DMat B, bool add, int L) {
  const int64_t m = A.nrows, l = A.ncols, n = B.ncols;
  if (L == 0) {
    if (int rc = reserve_apk(e, packed_a_words(m, l, 1))) return rc;
    return launch_leaf(e, st, C.p, C.stride, 0, A.p, A.stride, 0, B.p, B.stride, 0, m, l, n, 1, add, 0);
  }
  const int64_t me = m - m % (1ll << L), le = l - l % (64ll << L), ne = n - n % (64ll << L);
  // packed-A scratch big enough for the batched leaves and for every remainder strip
  size_t a7_strips = 0;
  {
    const size_t s1 = n > ne ? packed_a_words(m, l, 1) : 0;
    const size_t s2 = m > me ? packed_a_words(m - me, l, 1) : 0;
    const size_t s3 = l > le ? packed_a_words(me, l - le, 1) : 0;
    a7_strips = s1 > s2 ? s1 : s2;
    if (s3 > a7_strips) a7_strips = s3;
  }
  if (int rc = bfs_product(e, st, dview(C, 0, 0, me, ne), dview(A, 0, 0, me, le), dview(B, 0, 0, le, ne), add, L, a7_strips)) return rc;
  // remainder strips: right columns, bottom rows, trailing inner slab
  if (n > ne)
    if (int rc = launch_leaf(e, st, C.p + ne / 64, C.stride, 0, A.p, A.stride, 0, B.p + ne / 64, B.stride, 0, m, l, n - ne, 1, add, 0)) return rc;
  if (m > me)
    if (int rc = launch_leaf(e, st, C.p + me * C.stride, C.stride, 0, A.p + me * A.stride, A.stride, 0, B.p, B.stride, 0, m - me, l, ne, 1, add, 0)) return rc;
  if (l > le)
    if (int rc = launch_leaf(e, st, C.p, C.stride, 0, A.p + le / 64, A.stride, 0, B.p + le * B.stride, B.stride, 0, me, l - le, ne, 1, true, 0)) return rc;
  return 0;
}

// C (+)= A*B with L Strassen-Winograd levels.  The breadth-first schedule keeps 3 * (7/4)^L operand
// sizes resident; when that does not fit, the TOP level runs depth-first like the reference
// (strassen.c:111-150, in the product form of winograd_scatter): seven sub-products one after the
// other, each through this function again with L - 1 levels, three quarter-size temporaries.
int product(Engine *e, hipStream_t st, DMat C, DMat A, DMat B, bool add, int L, double budget) {
  const int64_t m = A.nrows, l = A.ncols, n = B.ncols;
  if (L == 0 || (double)bfs_words_bound(m, l, n, L) * 8.0 <= budget) return bfs_with_strips(e, st, C, A, B, add, L);
  const int64_t me = m - m % 2, le = l - l % 128, ne = n - n % 128;  // halves on rows / whole words
  if (me == 0 || le == 0 || ne == 0) return bfs_with_strips(e, st, C, A, B, add, 0);
  const int64_t hm = me / 2, hl = le / 2, hn = ne / 2, wl = hl / 64, wn = hn / 64;
  // X (hm x hl), Y (hl x hn), P (hm x hn): from the depth-first stack (outside the workspace, which the sub-products
  // re-carve); engine_mul reserved it for the whole recursion, so nothing is allocated or synchronised here
  const size_t xw = (size_t)hm * wl, yw = (size_t)hl * wn, pw = (size_t)hm * wn;
  const size_t df_mark = e->df_used;
  if (e->df_used + xw + yw + pw > e->df_cap) return (int)hipErrorOutOfMemory;  // cannot happen: df_words() is the same walk
  word *tmp = e->df_pool + e->df_used;
  e->df_used += (xw + yw + pw + 31) & ~(size_t)31;
  DMat X{tmp, hm, hl, wl}, Y{tmp + xw, hl, hn, wn}, P{tmp + xw + yw, hm, hn, wn};
  auto qa = [&](int i, int j) { return dview(A, i * hm, j * hl, hm, hl); };
  auto qb = [&](int i, int j) { return dview(B, i * hl, j * hn, hl, hn); };
  auto qc = [&](int i, int j) { return dview(C, i * hm, j * hn, hm, hn); };
  auto xor3v = [&](DMat D, DMat U, DMat V) { return (int)gf2_launch_rowwise(st, 0, D.p, D.stride, U.p, U.stride, V.p, V.stride, D.nrows, words_of(D.ncols)); };
  auto copyv = [&](DMat D, DMat U) { return (int)gf2_launch_rowwise(st, 1, D.p, D.stride, U.p, U.stride, nullptr, 0, D.nrows, words_of(D.ncols)); };
  int rc = 0;
  bool touched[2][2] = {{add, add}, {add, add}};  // has this quadrant of C received its first term?
  for (int j = 0; j < 7 && rc == 0; ++j) {
    // operand combinations of winograd_child: [A11, A12, S4, A22, S1, S2, S3] x [B11, B21, B22, T4, T1, T2, T3]
    DMat a = X, b = Y;
    switch (j) {
      case 0: a = qa(0, 0); b = qb(0, 0); break;
      case 1: a = qa(0, 1); b = qb(1, 0); break;
      case 2: rc = xor3v(X, qa(1, 0), qa(1, 1)); if (!rc) rc = xor3v(X, X, qa(0, 0)); if (!rc) rc = xor3v(X, X, qa(0, 1)); b = qb(1, 1); break;
      case 3: a = qa(1, 1);
              rc = xor3v(Y, qb(1, 1), qb(0, 1)); if (!rc) rc = xor3v(Y, Y, qb(0, 0)); if (!rc) rc = xor3v(Y, Y, qb(1, 0)); break;
      case 4: rc = xor3v(X, qa(1, 0), qa(1, 1)); if (!rc) rc = xor3v(Y, qb(0, 1), qb(0, 0)); break;
      case 5: rc = xor3v(X, qa(1, 0), qa(1, 1)); if (!rc) rc = xor3v(X, X, qa(0, 0));
              if (!rc) rc = xor3v(Y, qb(1, 1), qb(0, 1)); if (!rc) rc = xor3v(Y, Y, qb(0, 0)); break;
      default: rc = xor3v(X, qa(0, 0), qa(1, 0)); if (!rc) rc = xor3v(Y, qb(1, 1), qb(0, 1)); break;
    }
    if (rc) break;
    rc = product(e, st, P, a, b, false, L - 1, budget);
    if (rc) break;
    // product j goes to the quadrants winograd_scatter names
    static const int targets[7][4] = {{1, 1, 1, 1}, {1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 1, 0, 1}, {0, 1, 1, 1}, {0, 0, 1, 1}};
    for (int q = 0; q < 4 && rc == 0; ++q) {
      if (!targets[j][q]) continue;
      DMat cq = qc(q >> 1, q & 1);
      rc = touched[q >> 1][q & 1] ? xor3v(cq, cq, P) : copyv(cq, P);
      touched[q >> 1][q & 1] = true;
    }
  }
  e->df_used = df_mark;  // later users of the stack are ordered behind these launches by the stream
  if (rc) return rc;
  // remainder strips of THIS level, by direct (chunked) leaf products like strassen.c:170-204
  if (n > ne) { if ((rc = reserve_apk(e, packed_a_words(m, l, 1)))) return rc;
                if ((rc = launch_leaf(e, st, C.p + ne / 64, C.stride, 0, A.p, A.stride, 0, B.p + ne / 64, B.stride, 0, m, l, n - ne, 1, add, 0))) return rc; }
  if (m > me) { if ((rc = reserve_apk(e, packed_a_words(m - me, l, 1)))) return rc;
                if ((rc = launch_leaf(e, st, C.p + me * C.stride, C.stride, 0, A.p + me * A.stride, A.stride, 0, B.p, B.stride, 0, m - me, l, ne, 1, add, 0))) return rc; }
  if (l > le) { if ((rc = reserve_apk(e, packed_a_words(me, l - le, 1)))) return rc;
                if ((rc = launch_leaf(e, st, C.p, C.stride, 0, A.p + le / 64, A.stride, 0, B.p + le * B.stride, B.stride, 0, me, l - le, ne, 1, true, 0))) return rc; }
  return 0;
}

// words of the depth-first stack product() will use for these dimensions: the same decisions, no launches
size_t df_words(const Engine *e, int64_t m, int64_t l, int64_t n, int L, double budget) {
  if (L == 0 || (double)bfs_words_bound(m, l, n, L) * 8.0 <= budget) return 0;
  const int64_t me = m - m % 2, le = l - l % 128, ne = n - n % 128;
  if (me == 0 || le == 0 || ne == 0) return 0;
  const int64_t hm = me / 2, hl = le / 2, hn = ne / 2;
  const size_t here = (((size_t)hm * (hl / 64) + (size_t)hl * (hn / 64) + (size_t)hm * (hn / 64)) + 31) & ~(size_t)31;
  return here + df_words(e, hm, hl, hn, L - 1, budget);
}

int engine_mul(Engine *e, hipStream_t st, DMat C, DMat A, DMat B, bool add, int cutoff) {
  const int64_t m = A.nrows, l = A.ncols, n = B.ncols;
  if (m == 0 || n == 0) return 0;
  // the engine's own plan may cut the rows into blocks, each at its depth (plan_row_blocks); a caller's cutoff or the developer's
  // M4RI_AMD_LEVELS mean one product at that depth
  std::vector<RowBlock> blocks;
  if (cutoff == 0 && !getenv("M4RI_AMD_LEVELS")) plan_row_blocks(m, l, n, blocks);
  if (blocks.empty()) blocks.push_back(RowBlock{m, plan_levels(m, l, n, cutoff)});
  for (RowBlock &b : blocks)  // every level halves l and n on word boundaries and m on rows: need a non-empty even block
    while (b.levels > 0 && ((b.rows >> b.levels) == 0 || (l / (64ll << b.levels)) == 0 || (n / (64ll << b.levels)) == 0)) --b.levels;
  const int L = blocks[0].levels;
  e->stats.levels = L;
  // depth-first levels (workspace larger than the device has left) take their temporaries from a grow-only stack
  // sized here, before the first launch: product() itself never allocates, frees or synchronises
  const double budget = workspace_budget(e);
  size_t need = 0;
  for (const RowBlock &b : blocks) { const size_t w = df_words(e, b.rows, l, n, b.levels, budget); if (w > need) need = w; }
  if (need > e->df_cap) {
    HIPTRY(hipDeviceSynchronize());
    if (e->df_pool) HIPTRY(hipFree(e->df_pool));
    e->df_pool = nullptr; e->df_cap = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&e->df_pool), need * sizeof(word)));
    e->df_cap = need;
  }
  int64_t r0 = 0;
  for (const RowBlock &b : blocks) {  // one after the other on the stream: they share the workspace
    e->df_used = 0;
    if (int rc = product(e, st, dview(C, r0, 0, b.rows, C.ncols), dview(A, r0, 0, b.rows, A.ncols), B, add, b.levels, budget)) return rc;
    r0 += b.rows;
  }
  return 0;
}

// `batch` products of ONE shape, X_b = X + b * x_bs words: scheduled as one product with `batch` times the parents in every pass and
// `batch` times the products in the leaf launch (bfs_product), at the depth the model picks for the batch.  What does not fit that
// form -- a ragged shape (remainder strips), a workspace beyond the budget, rows worth cutting into blocks -- goes one product at a
// time through engine_mul: same bits either way.
int engine_mul_batch(Engine *e, hipStream_t st, DMat C, int64_t c_bs, DMat A, int64_t a_bs, DMat B, int64_t b_bs, int64_t batch, bool add, int cutoff) {
  const int64_t m = A.nrows, l = A.ncols, n = B.ncols;
  if (m == 0 || n == 0 || batch <= 0) return 0;
  auto one_by_one = [&]() {
    for (int64_t b = 0; b < batch; ++b) {
      DMat c = C, a = A, bb = B;
      c.p += b * c_bs; a.p += b * a_bs; bb.p += b * b_bs;
      if (int rc = engine_mul(e, st, c, a, bb, add, cutoff)) return rc;
    }
    return 0;
  };
  if (batch == 1 || l == 0) return one_by_one();
  int L = 0;
  if (cutoff == 0 && !getenv("M4RI_AMD_LEVELS")) {
    L = model_levels(m, l, n, nullptr, (double)batch);
    std::vector<RowBlock> blocks;
    plan_row_blocks(m, l, n, blocks);
    if (blocks.size() != 1) return one_by_one();  // rows in blocks: a single product's business
  } else {
    L = plan_levels(m, l, n, cutoff);
  }
  while (L > 0 && ((m >> L) == 0 || (l / (64ll << L)) == 0 || (n / (64ll << L)) == 0)) --L;
  if (L > 0 && (m % (1ll << L) != 0 || l % (64ll << L) != 0 || n % (64ll << L) != 0)) return one_by_one();  // strips
  if (L > 0 && (double)bfs_words_bound(m, l, n, L) * 8.0 * (double)batch > workspace_budget(e)) return one_by_one();
  const int fuse = L < g_max_fuse ? L : g_max_fuse;
  if (add && fuse == 3 && L == 3 && !scheme_applies(fuse, m >> L, l >> L, n >> L)) return one_by_one();  // (the accumulate temporary is one product's)
  e->stats.levels = L;
  e->df_used = 0;
  if (L == 0) {
    if (int rc = reserve_apk(e, packed_a_words(m, l, batch))) return rc;
    return launch_leaf(e, st, C.p, C.stride, c_bs, A.p, A.stride, a_bs, B.p, B.stride, b_bs, m, l, n, batch, add, 0);
  }
  return bfs_product(e, st, C, A, B, add, L, 0, batch, c_bs, a_bs, b_bs);
}

void reset_stats(Engine *e) {
  const double ws = (double)e->ws_cap * 8.0;
  // keep un-read profiling events out of the next call's sum (cumulative mode keeps them for the total)
  if (e->profiling != 2) {
    for (auto &pr : e->pending) { e->event_pool.push_back(pr.e0); e->event_pool.push_back(pr.e1); }
    e->pending.clear();
  }
  e->call_seq += 1;
  e->stats = m4ri_amd_stats{};
  e->stats.workspace_bytes = ws;
}

}  // namespace

// One workspace per device: products issued on different streams are ordered against each other
// (a product on the same stream as the previous one is ordered by the stream itself).
int order_after_previous(Engine *e, hipStream_t st) {
  if (e->have_last && e->last_stream != st && e->last_done) HIPTRY(hipStreamWaitEvent(st, e->last_done, 0));
  return 0;
}
int mark_done(Engine *e, hipStream_t st) {
  if (!e->last_done) HIPTRY(hipEventCreateWithFlags(&e->last_done, hipEventDisableTiming));
  HIPTRY(hipEventRecord(e->last_done, st));
  e->last_stream = st;
  e->have_last   = true;
  return 0;
}

// ================================ C ABI (part 2 of include/m4ri_amd.h) ==========================
extern "C" {

int m4ri_amd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int m4ri_amd_init(int device) {
  HIPTRY(hipSetDevice(device));
  EngineLock el;
  return el.e ? 0 : (int)hipErrorInvalidDevice;
}

int m4ri_amd_mul_dev(word *C, int64_t c_stride, const word *A, int64_t a_stride, const word *B,
                     int64_t b_stride, int64_t m, int64_t l, int64_t n, int add, int cutoff, void *stream) {
  EngineLock el;
  Engine *e = el.e;
  if (!e || cutoff < 0 || m < 0 || l < 0 || n < 0) return (int)hipErrorInvalidValue;
  reset_stats(e);
  if (cutoff > 0) { cutoff = cutoff / 64 * 64; if (cutoff < 64) cutoff = 64; }  // strassen.c:351-354
  DMat dC{C, m, n, c_stride}, dA{const_cast<word *>(A), m, l, a_stride}, dB{const_cast<word *>(B), l, n, b_stride};
  if (int rc = order_after_previous(e, (hipStream_t)stream)) return rc;
  if (int rc = engine_mul(e, (hipStream_t)stream, dC, dA, dB, add != 0, cutoff)) return rc;
  return mark_done(e, (hipStream_t)stream);
}

// `batch` independent products of one shape, C_b (+)= A_b * B_b with X_b = X + b * x_bs words, with the Strassen-Winograd levels of
// m4ri_amd_mul_dev and every launch shared by the whole batch: several sub-products of a sharded level on one rank (multi.hip)
// fill the chip together where each alone leaves its last round of tiles half empty.  Same bits as `batch` calls of m4ri_amd_mul_dev.
int m4ri_amd_mul_batch_dev(word *C, int64_t c_stride, int64_t c_bs, const word *A, int64_t a_stride, int64_t a_bs, const word *B,
                           int64_t b_stride, int64_t b_bs, int64_t m, int64_t l, int64_t n, int64_t batch, int add, int cutoff, void *stream) {
  EngineLock el;
  Engine *e = el.e;
  if (!e || cutoff < 0 || m < 0 || l < 0 || n < 0 || batch < 0) return (int)hipErrorInvalidValue;
  if (batch == 0) return 0;
  reset_stats(e);
  if (cutoff > 0) { cutoff = cutoff / 64 * 64; if (cutoff < 64) cutoff = 64; }  // strassen.c:351-354
  DMat dC{C, m, n, c_stride}, dA{const_cast<word *>(A), m, l, a_stride}, dB{const_cast<word *>(B), l, n, b_stride};
  if (int rc = order_after_previous(e, (hipStream_t)stream)) return rc;
  if (int rc = engine_mul_batch(e, (hipStream_t)stream, dC, c_bs, dA, a_bs, dB, b_bs, batch, add != 0, cutoff)) return rc;
  return mark_done(e, (hipStream_t)stream);
}

int m4ri_amd_m4rm_dev(word *C, int64_t c_stride, const word *A, int64_t a_stride, const word *B,
                      int64_t b_stride, int64_t m, int64_t l, int64_t n, int add, int ksplit, void *stream) {
  EngineLock el;
  Engine *e = el.e;
  if (!e || m < 0 || l < 0 || n < 0) return (int)hipErrorInvalidValue;
  reset_stats(e);
  if (int rc = order_after_previous(e, (hipStream_t)stream)) return rc;
  if (int rc = reserve_apk(e, packed_a_words(m, l, 1))) return rc;
  if (int rc = launch_leaf(e, (hipStream_t)stream, C, c_stride, 0, A, a_stride, 0, B, b_stride, 0, m, l, n, 1, add != 0, ksplit)) return rc;
  return mark_done(e, (hipStream_t)stream);
}

// `batch` products of one shape in one launch, C_b (+)= A_b * B_b with X_b = X + b * x_bs words: plain M4RM leaves, no
// Strassen levels -- for the many small equal products of a blocked algorithm (the levels of the triangular inverse,
// trsm.hip), where one launch per product costs more than the product.
int m4ri_amd_m4rm_batch_dev(word *C, int64_t c_stride, int64_t c_bs, const word *A, int64_t a_stride, int64_t a_bs, const word *B,
                            int64_t b_stride, int64_t b_bs, int64_t m, int64_t l, int64_t n, int64_t batch, int add, void *stream) {
  EngineLock el;
  Engine *e = el.e;
  if (!e || m < 0 || l < 0 || n < 0 || batch < 0) return (int)hipErrorInvalidValue;
  if (batch == 0 || m == 0 || n == 0) return 0;
  reset_stats(e);
  if (int rc = order_after_previous(e, (hipStream_t)stream)) return rc;
  if (int rc = launch_leaf(e, (hipStream_t)stream, C, c_stride, c_bs, A, a_stride, a_bs, B, b_stride, b_bs, m, l, n, batch, add != 0, 0)) return rc;
  return mark_done(e, (hipStream_t)stream);
}

int m4ri_amd_xor_dev(word *C, int64_t c_stride, const word *A, int64_t a_stride, const word *B,
                     int64_t b_stride, int64_t rows, int64_t ncols, void *stream) {
  return (int)gf2_launch_xor_masked((hipStream_t)stream, C, c_stride, A, a_stride, B, b_stride, rows, ncols);
}

int m4ri_amd_fill_dev(word *M, int64_t stride, int64_t rows, int64_t ncols, uint64_t seed, void *stream) {
  return (int)gf2_launch_fill_splitmix((hipStream_t)stream, M, stride, rows, ncols, seed);
}

int m4ri_amd_fill_rows_dev(word *M, int64_t stride, int64_t row0, int64_t rows, int64_t ncols, uint64_t seed, void *stream) {
  return (int)gf2_launch_fill_splitmix_rows((hipStream_t)stream, M, stride, row0, rows, ncols, seed);
}

int m4ri_amd_mask_tail_dev(word *M, int64_t stride, int64_t rows, int64_t ncols, void *stream) {
  return (int)gf2_launch_mask_tail((hipStream_t)stream, M, stride, rows, ncols);
}

double m4ri_amd_model_seconds(int64_t m, int64_t l, int64_t n, int levels) {
  if (m <= 0 || l <= 0 || n <= 0 || levels < 0 || levels > MAX_LEVELS) return 0.0;
  return depth_model_seconds(m, l, n, levels);
}

// levels < 0: the depth the model itself picks for the batch
double m4ri_amd_model_seconds_batch(int64_t m, int64_t l, int64_t n, int levels, int64_t batch) {
  if (m <= 0 || l <= 0 || n <= 0 || levels > MAX_LEVELS || batch < 1) return 0.0;
  if (levels < 0) { double t = 0; model_levels(m, l, n, &t, (double)batch); return t; }
  return depth_model_seconds(m, l, n, levels, (double)batch);
}

int m4ri_amd_plan_small_leaf(int64_t m, int64_t l, int64_t n, int64_t batch, int cus) {
  if (m <= 0 || l <= 0 || n <= 0 || batch <= 0 || !small_leaf_wanted(m, l, n, batch)) return 0;
  const int64_t wn = words_of(n), tiles = ((m + 255) / 256) * ((wn + 7) / 8) * batch;
  return gf2_m4rm_small_ksplit(tiles, words_of(l), cus > 0 ? cus : 256, batch * m * wn);
}

int m4ri_amd_plan_row_blocks(int64_t m, int64_t l, int64_t n, int64_t *rows, int *levels, int cap) {
  if (m <= 0 || l <= 0 || n <= 0) return 0;
  std::vector<RowBlock> blocks;
  plan_row_blocks(m, l, n, blocks);
  for (size_t i = 0; i < blocks.size() && (int)i < cap; ++i) {
    if (rows) rows[i] = blocks[i].rows;
    if (levels) levels[i] = blocks[i].levels;
  }
  return (int)blocks.size();
}

int m4ri_amd_plan_levels(int64_t m, int64_t l, int64_t n, int cutoff) {  // pure host logic: no device needed
  if (m <= 0 || l <= 0 || n <= 0 || cutoff < 0) return 0;
  if (cutoff > 0) { cutoff = cutoff / 64 * 64; if (cutoff < 64) cutoff = 64; }  // strassen.c:351-354
  return plan_levels(m, l, n, cutoff);
}

int64_t m4ri_amd_set_workspace_budget(int64_t bytes) {
  std::lock_guard<std::mutex> lk(g_cfg_mu);
  const int64_t old = g_ws_budget;
  if (bytes >= 0) g_ws_budget = bytes;
  return old;
}

int m4ri_amd_set_max_fuse(int levels) {
  std::lock_guard<std::mutex> lk(g_cfg_mu);
  const int old = g_max_fuse;
  if (levels >= 1 && levels <= 4) g_max_fuse = levels;
  return old;
}

void m4ri_amd_set_profiling(int on) {
  EngineLock el;
  Engine *e = el.e;
  if (!e) return;
  for (auto &pr : e->pending) { e->event_pool.push_back(pr.e0); e->event_pool.push_back(pr.e1); }
  e->pending.clear();
  e->cum_ms = 0; e->cum_launches = 0;
  e->profiling = on < 0 ? 0 : on > 2 ? 2 : on;
}

int m4ri_amd_get_stats(m4ri_amd_stats *out) {
  EngineLock el;
  Engine *e = el.e;
  if (!e || !out) return (int)hipErrorInvalidValue;
  if (!e->pending.empty()) {
    HIPTRY(hipEventSynchronize(e->pending.back().e1));
    double sum = 0;
    for (auto &pr : e->pending) {
      float ms = 0;
      HIPTRY(hipEventElapsedTime(&ms, pr.e0, pr.e1));
      if (pr.call == e->call_seq) sum += ms;  // the most recent product's launches
      e->cum_ms += ms; e->cum_launches += 1;
      e->event_pool.push_back(pr.e0); e->event_pool.push_back(pr.e1);
    }
    e->pending.clear();
    e->stats.leaf_ms += sum;
  }
  e->stats.cum_leaf_ms       = e->cum_ms;
  e->stats.cum_leaf_launches = e->cum_launches;
  *out = e->stats;
  return 0;
}

void gf2_release_staging(void);  // mzd_api.hip: the host entry points' staging arena
void gf2_release_multi(void);    // multi.hip: the per-rank arenas of the multi-device path

void m4ri_amd_release_workspace(void) {
  gf2_release_staging();
  gf2_release_multi();
  EngineLock el;
  Engine *e = el.e;
  if (!e || !e->ws) return;
  (void)hipDeviceSynchronize();
  (void)hipFree(e->ws);
  e->ws = nullptr; e->ws_cap = 0; e->ws_used = 0;
  e->apk = nullptr; e->apk_words = 0; e->part = nullptr;
  if (e->df_pool) (void)hipFree(e->df_pool);
  e->df_pool = nullptr; e->df_cap = 0; e->df_used = 0;
}

}  // extern "C"
