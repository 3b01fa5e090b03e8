// multi.hip -- ONE product over several GPUs: the multi-device meaning of the reference's block-parallel
// entry points
//   mzd_mul_mp / mzd_addmul_mp  -> _mzd_mul_mp4 / _mzd_addmul_mp4      /root/reference m4ri/mp.c:158-324
// (C cut 2 x 2, four OpenMP sections, every section two products).  On a node of MI355X the units that
// are handed out are not blocks of C but the SUB-PRODUCTS OF THE TOP STRASSEN-WINOGRAD LEVEL(S)
// (strassen.c:111-150: 7 per level): 7 or 49 independent products of (n/2)^3 or (n/4)^3, the additions
// of the level done where the data lives, and the only traffic between GPUs the operands of the
// sub-products one way and their results the other -- the "reduce" of the scheme, over xGMI.
//
// Layout ("slab-cyclic"): with S = 2^levels row blocks per matrix, rank r of W holds, of EVERY block,
// rows [cut(r), cut(r+1)) (cut(r) = rows_per_block * r / W).  Its local parent is those S slabs stacked,
// i.e. a matrix with the same quadrant structure as the global one and 1/W of its rows.  Consequences:
//   * the Winograd operand combinations (S1..S4 / T1..T4) of row slab r need only row slab r of the
//     four quadrants: every rank runs the ordinary fused down pass (aux_kernels.hip) on its local
//     parent -- 1/W of the pass work each, no communication;
//   * what a rank gets out is slab r of every child; child j is multiplied on rank owner(j) = j % W, so
//     slab r of child j travels r -> owner(j): a full-mesh exchange in which every directed xGMI link
//     carries 1/W of one operand -- no ring, no hop through a third GPU, all 7 links of a GPU busy;
//   * products come back the same way (slab r of P_j: owner(j) -> r) and the ordinary up pass on the
//     local slabs yields the local parent of C -- in the layout A and B came in, so products chain.
// Matrices whose dimensions do not divide are zero-padded in the local parents (rows to S, columns to
// 64*S): padding costs < 128 columns and removes every remainder strip.
//
// Two front ends over one plan (m4ri_amd_shard_plan, pure host arithmetic, exported):
//   * the per-rank device API (m4ri_amd_shard_down_dev / _up_dev + the piece table): one process per
//     GPU moves the pieces itself -- bench.py does it with RCCL send/recv via torch.distributed;
//   * ONE process, all devices (the second half of this file): distributed device-resident matrices
//     (m4ri_amd_dmat), m4ri_amd_dmat_mul with the row-slab and the Strassen-sharded schedule, pieces
//     pulled by hipMemcpyPeerAsync on per-device copy streams under the products; mzd_mul_mp /
//     m4ri_amd_mul_multi on host mzd_t = upload + that + download.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

extern "C" {
hipError_t gf2_launch_winograd_down(hipStream_t s, int bside, const word *parent, int64_t p_stride, int64_t p_bs, word *child,
                                    int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_up(hipStream_t s, int acc, const word *prod, word *parent, int64_t o_stride, int64_t o_bs,
                                  int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_down2(hipStream_t s, int bside, const word *gparent, int64_t p_stride, int64_t p_bs, word *gchild,
                                     int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_winograd_up2(hipStream_t s, int acc, const word *prod, word *gparent, int64_t o_stride, int64_t o_bs,
                                   int64_t nparents, int64_t crows, int64_t cw);
hipError_t gf2_launch_mask_tail(hipStream_t s, word *M, int64_t stride, int64_t rows, int64_t ncols);
// the rank-R scheme of the 4 x 4 x 4 block product (scheme_passes.hip): two sharded levels = the scheme once, R sub-products instead of 49
int gf2_scheme444_rank(void);
int gf2_scheme444_ok(int levels, int64_t a_rows, int64_t a_cw, int64_t b_rows, int64_t b_cw);
hipError_t gf2_launch_scheme_down(hipStream_t s, int levels, int bside, const word *anc, int64_t p_stride, int64_t p_bs, word *child, int64_t nparents,
                                  int64_t crows, int64_t cw);
hipError_t gf2_launch_scheme_up(hipStream_t s, int levels, int acc, const word *prod, word *anc, int64_t o_stride, int64_t o_bs, int64_t nparents,
                                int64_t crows, int64_t cw);
}

namespace {

#define HIPTRY(expr)                                  \
  do {                                                \
    hipError_t e_ = (hipError_t)(expr);               \
    if (e_ != hipSuccess) return (int)e_;             \
  } while (0)

int64_t cut_of(int64_t rows, int world, int r) { return rows * (int64_t)r / (int64_t)world; }
int64_t roundup(int64_t x, int64_t q) { return (x + q - 1) / q * q; }
int64_t pad32(int64_t w) { return (w + 31) & ~(int64_t)31; }  // 256-byte granules

// Sub-products of `levels` sharded levels.  Two levels are ONE application of the rank-R scheme for the 4 x 4 x 4 block product where
// its passes take the slabs (row-major children on both sides: whole 64-word groups per child row, no empty slab) and R < 49: R = 47
// sub-products instead of Strassen-Winograd's 49 -- on 8 ranks 6 rounds instead of 7.  Every rank must come to the same answer: the
// rule only looks at the plan.
int nprod_of(int world, int levels, int64_t bm, int64_t bl, int64_t cwl, int64_t cwn) {
  if (levels != 2) return 7;
  const int R = gf2_scheme444_rank();
  if (R >= 49 || cwl % 64 != 0 || cwn % 64 != 0 || bm < world || bl < world) return 49;
  if (!gf2_scheme444_ok(2, 64, cwl, 64, cwn)) return 49;  // (the M4RI_AMD_SCHEME switch; the passes' own 32-bit index limits)
  if ((bm / world + 1) * cwl > 0x3fffffffLL || (bl / world + 1) * cwn > 0x3fffffffLL) return 49;
  return R;
}
bool plan_uses_scheme(const m4ri_amd_shard_plan *p) { return p->levels == 2 && p->nprod != 49; }

int owned_count(const m4ri_amd_shard_plan *p, int rank) {
  return rank < p->nprod ? (p->nprod - rank + p->world - 1) / p->world : 0;
}

}  // namespace

extern "C" {

int64_t m4ri_amd_shard_cut(int64_t rows, int world, int r) { return cut_of(rows, world, r); }

int m4ri_amd_shard_owner(const m4ri_amd_shard_plan *p, int j) { return j % p->world; }

int m4ri_amd_shard_plan_make(m4ri_amd_shard_plan *p, int world, int64_t m, int64_t l, int64_t n, int levels) {
  if (!p || world < 1 || m <= 0 || l <= 0 || n <= 0 || levels < 0 || levels > 2) return -1;
  if (levels == 0) {
    // fewest "rounds" of sub-products on the busiest rank, as a fraction of the whole: ceil(P / W) / P with P = 7 or the second
    // level's count (49, or the scheme's 47); the second level only when it wins and its sub-products stay large (>= 2048 on every side)
    const int64_t M4 = roundup(m, 4), L4 = roundup(l, 256), N4 = roundup(n, 256);
    const int np2   = nprod_of(world, 2, M4 / 4, L4 / 4, L4 / 256, N4 / 256);
    const double f1 = (double)((7 + world - 1) / world) / 7.0, f2 = (double)((np2 + world - 1) / world) / (double)np2;
    levels = (f2 < f1 - 1e-9 && m / 4 >= 2048 && l / 4 >= 2048 && n / 4 >= 2048) ? 2 : 1;
  }
  memset(p, 0, sizeof *p);
  p->world  = world;
  p->levels = levels;
  p->blocks = 1 << levels;
  p->m = m; p->l = l; p->n = n;
  p->M = roundup(m, p->blocks);
  p->L = roundup(l, 64ll * p->blocks);
  p->N = roundup(n, 64ll * p->blocks);
  p->bm  = p->M / p->blocks;
  p->bl  = p->L / p->blocks;
  p->cwl = p->L / p->blocks / 64;
  p->cwn = p->N / p->blocks / 64;
  p->nprod = nprod_of(world, levels, p->bm, p->bl, p->cwl, p->cwn);
  return 0;
}

// Words of the buffers rank `rank` needs (m4ri_amd.h: M4RI_AMD_SHARD_BUF_*), contiguous, stride = width
int64_t m4ri_amd_shard_buffer_words(const m4ri_amd_shard_plan *p, int rank, int which) {
  if (!p || rank < 0 || rank >= p->world) return -1;
  const int64_t sa = cut_of(p->bm, p->world, rank + 1) - cut_of(p->bm, p->world, rank);
  const int64_t sb = cut_of(p->bl, p->world, rank + 1) - cut_of(p->bl, p->world, rank);
  const int64_t no = owned_count(p, rank);
  switch (which) {
    case M4RI_AMD_SHARD_BUF_LOCAL_A: return p->blocks * sa * (p->L / 64);
    case M4RI_AMD_SHARD_BUF_LOCAL_B: return p->blocks * sb * (p->N / 64);
    case M4RI_AMD_SHARD_BUF_LOCAL_C: return p->blocks * sa * (p->N / 64);
    case M4RI_AMD_SHARD_BUF_CHILD_A: return p->nprod * sa * p->cwl;
    case M4RI_AMD_SHARD_BUF_CHILD_B: return p->nprod * sb * p->cwn;
    case M4RI_AMD_SHARD_BUF_SLABS_P: return p->nprod * sa * p->cwn;
    case M4RI_AMD_SHARD_BUF_OPER_A:  return no * p->bm * p->cwl;
    case M4RI_AMD_SHARD_BUF_OPER_B:  return no * p->bl * p->cwn;
    case M4RI_AMD_SHARD_BUF_PROD:    return no * p->bm * p->cwn;
    default: return -1;
  }
}

// Rows of rank `rank`'s slab of one block: which = 0 rows of A / C / a product, 1 rows of B
int64_t m4ri_amd_shard_slab_rows(const m4ri_amd_shard_plan *p, int rank, int which) {
  const int64_t b = which ? p->bl : p->bm;
  return cut_of(b, p->world, rank + 1) - cut_of(b, p->world, rank);
}

int m4ri_amd_shard_piece_of(const m4ri_amd_shard_plan *p, int side, int j, int r, m4ri_amd_shard_piece *out) {
  if (!p || !out || side < 0 || side > 2 || j < 0 || j >= p->nprod || r < 0 || r >= p->world) return -1;
  const int64_t brows = side == 1 ? p->bl : p->bm;          // rows of the whole operand / product
  const int64_t cw    = side == 0 ? p->cwl : p->cwn;        // words per row
  const int64_t c0 = cut_of(brows, p->world, r), c1 = cut_of(brows, p->world, r + 1);
  out->holder     = r;
  out->owner      = j % p->world;
  out->holder_off = (int64_t)j * (c1 - c0) * cw;
  out->owner_off  = (int64_t)(j / p->world) * brows * cw + c0 * cw;
  out->words      = (c1 - c0) * cw;
  return 0;
}

// local down passes: local parents (slab-cyclic rows) -> slab `rank` of every child, children back to back
int m4ri_amd_shard_down_dev(const m4ri_amd_shard_plan *p, int rank, const word *A_local, int64_t a_stride, const word *B_local,
                            int64_t b_stride, word *child_a, word *child_b, void *stream) {
  if (!p || rank < 0 || rank >= p->world) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int64_t sa = m4ri_amd_shard_slab_rows(p, rank, 0), sb = m4ri_amd_shard_slab_rows(p, rank, 1);
  if (p->levels == 1) {
    if (A_local) HIPTRY(gf2_launch_winograd_down(st, 0, A_local, a_stride, 0, child_a, 1, sa, p->cwl));
    if (B_local) HIPTRY(gf2_launch_winograd_down(st, 1, B_local, b_stride, 0, child_b, 1, sb, p->cwn));
  } else if (plan_uses_scheme(p)) {
    if (A_local && sa > 0) HIPTRY(gf2_launch_scheme_down(st, 2, 0, A_local, a_stride, 0, child_a, 1, sa, p->cwl));
    if (B_local && sb > 0) HIPTRY(gf2_launch_scheme_down(st, 2, 1, B_local, b_stride, 0, child_b, 1, sb, p->cwn));
  } else {
    if (A_local) HIPTRY(gf2_launch_winograd_down2(st, 0, A_local, a_stride, 0, child_a, 1, sa, p->cwl));
    if (B_local) HIPTRY(gf2_launch_winograd_down2(st, 1, B_local, b_stride, 0, child_b, 1, sb, p->cwn));
  }
  return 0;
}

// local up pass: slab `rank` of every product -> the local parent of C (add != 0: C_local ^= ...)
int m4ri_amd_shard_up_dev(const m4ri_amd_shard_plan *p, int rank, const word *slabs_p, word *C_local, int64_t c_stride, int add,
                          void *stream) {
  if (!p || rank < 0 || rank >= p->world) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int64_t sa = m4ri_amd_shard_slab_rows(p, rank, 0);
  if (p->levels == 1) HIPTRY(gf2_launch_winograd_up(st, add ? 1 : 0, slabs_p, C_local, c_stride, 0, 1, sa, p->cwn));
  else if (plan_uses_scheme(p)) { if (sa > 0) HIPTRY(gf2_launch_scheme_up(st, 2, add ? 1 : 0, slabs_p, C_local, c_stride, 0, 1, sa, p->cwn)); }
  else HIPTRY(gf2_launch_winograd_up2(st, add ? 1 : 0, slabs_p, C_local, c_stride, 0, 1, sa, p->cwn));
  return 0;
}

}  // extern "C"

// ================================ one process, all devices ==========================================
// The reference's multi-core entry is a C function (mzd_mul_mp, m4ri/mp.c:158-297, mp.h:47,62), so the multi-GPU schedules
// live here, behind the C boundary, on DISTRIBUTED, DEVICE-RESIDENT operands:
//
//   m4ri_amd_dmat      a matrix spread over the configured devices (m4ri_amd_set_devices) in one of three layouts --
//                      ROWS (rank r holds rows [r k, (r+1) k), k = ceil(rows / W)), CYCLIC1 / CYCLIC2 (slab-cyclic over the 2 / 4
//                      row blocks of 1 / 2 Strassen-Winograd levels, see the top of this file), REPLICATED (every rank holds it
//                      all).  Every dimension is zero-padded to a multiple of 256 bits in the local buffers, so all layouts of
//                      one matrix share ONE row stride and every piece that ever crosses a link is a contiguous run of rows.
//   m4ri_amd_dmat_mul  C (+)= A*B on such operands, the result left distributed: products chain with no PCIe traffic.  Two
//                      schedules, chosen by shape and world size like sharding.default_variant (m4ri_amd_multi_default_variant):
//                        * row slabs: C_r = A_r * B with B's row slabs all-gathered by peer copies on a copy stream while the
//                          product with the rank's OWN slab of B runs (no reduction; the reference's own row parallelism,
//                          m4ri/brilliantrussian.c:1121-1123; BASELINE configs[4] takes this at every world size);
//                        * Strassen sub-products: local down pass, slabs of the 7 / 49 sub-product operands pulled by their
//                          owners (hipMemcpyPeerAsync on the owner's inbound stream), the sub-products in row chunks so that
//                          a chunk's operands and the previous chunk's results travel under the multiplications, result slabs
//                          pulled back by their holders on a third stream, local up pass.
//                      One host thread per rank issues that rank's work (a persistent pool: the host posts a step in
//                      parallel, nothing in Python); cross-rank dependencies are HIP events, made safe to wait on by host
//                      flags that say "recorded".  Asynchronous: m4ri_amd_multi_sync() waits for the devices.
//   mzd_mul_mp / m4ri_amd_mul_multi on host matrices = upload (every device its own rows over its own PCIe link) + that +
//   download.
namespace {

constexpr uint8_t FLAG_WINDOW = 0x4;  // mzd.h:150
constexpr int64_t PAD_BITS   = 256;   // every dimension of a distributed matrix is padded to this in the local buffers
constexpr int MAX_RANKS      = 64;

enum Buf { B_CHILD_A = 0, B_CHILD_B, B_SLABS_P, B_OPER_A, B_OPER_B, B_PROD, B_GATHER, B_COUNT };

struct Rank {
  int device         = 0;
  hipStream_t st = nullptr, ci = nullptr, co = nullptr;  // compute; the JOIN streams of the inbound (operand) and outbound (result) copies
  // One copy stream per peer and direction: copies on ONE stream run one after the other, so the pieces a rank pulls from its W - 1
  // peers go on W - 1 streams and all links of the rank carry data at once (what one group of RCCL send/recvs gives the other
  // transport); the join streams wait for the link streams' events and carry the events everybody else waits for
  std::vector<hipStream_t> lin, lout;   // link streams, indexed by the peer rank
  std::vector<hipEvent_t> ev_lin, ev_lout;
  std::vector<char> lin_used, lout_used;  // this link stream has copies the next join has not collected yet
  // All events carry timestamps: they double as the marks of the per-phase timeline (m4ri_amd_multi_timeline)
  hipEvent_t ev_start = nullptr, ev_down = nullptr, ev_gather = nullptr, ev_first = nullptr, ev_back = nullptr;
  // "my part of the operation is complete", double-buffered by the parity of the lane's operation count: operation k records slot
  // k & 1 and waits for slot (k - 1) & 1 of every rank, which nobody re-records before operation k + 1 -- and that is only issued
  // after every thread of operation k has returned (Pool::run).  With ONE event a rank whose thread was late waited for the
  // CURRENT operation of the ranks that had already finished issuing (hipStreamWaitEvent takes the event's latest record): ranks
  // serialised behind each other by thread timing
  hipEvent_t ev_done[2] = {nullptr, nullptr};
  bool done_recorded[2] = {false, false};
  hipEvent_t ev_tl0 = nullptr, ev_tl1 = nullptr;  // first and last mark of the most recent PRODUCT (uploads, downloads, conversions leave them alone)
  std::vector<hipEvent_t> ev_in, ev_prod;  // per unit (round, row chunk) of the Strassen schedule
  word *arena = nullptr;
  size_t cap  = 0;  // words
  word *buf[B_COUNT] = {};
  std::atomic<int64_t> flag_down{0}, flag_prod{0};  // host flags: "ev_down of operation seq is recorded", "ev_prod[u] ... (seq * 4096 + u + 1)"
  // what the last operation recorded (for the timeline)
  int tl_units = 0;
  bool tl_strassen = false, tl_gathered = false, tl_valid = false;
  // pairs without peer access (or with M4RI_AMD_NO_PEER set: the test hook) copy through this pinned bounce buffer, in order on the link stream
  std::vector<word *> bounce;  // per stream slot: 0..W-1 inbound links, W..2W-1 outbound links, 2W the compute stream
};

// Two LANES of ranks over the same devices: each its own streams, events and arenas.  Lane 0 carries everything; lane 1 exists for
// m4ri_amd_dmat_mul_lane, so that two independent products can be in flight -- the operand and result transport of one under the
// multiplications of the other (the products themselves take turns on a device: one engine workspace, engine.hip).
constexpr int NUM_LANES = 2;
struct Lane {
  std::vector<std::unique_ptr<Rank>> ranks;
  int parity = 0;            // slot of ev_done the lane's NEXT operation records
  uint64_t seen_excl = 0;    // the latest exclusive operation (any lane) this lane's work is ordered behind
};

std::mutex g_multi_mu;
std::vector<int> g_devices;   // the devices products are spread over (an id may repeat: "virtual" ranks)
bool g_devices_set = false;
Lane g_lane[NUM_LANES];
std::vector<std::unique_ptr<Rank>> &g_ranks = g_lane[0].ranks;  // lane 0: the ranks every operation but a lane-1 product runs on
uint64_t g_excl_id = 0;       // exclusive operations issued so far (everything but a product: uploads, fills, conversions, downloads)
int g_excl_lane    = 0;       // the lane the latest one ran on
std::vector<char> g_staged;   // W x W: copies dst rank <- src rank go through the host (no peer access between their devices, or the test hook)
int g_pairs_staged = 0;
uint64_t g_config_gen = 1;    // bumped whenever the ranks are rebuilt: distributed matrices of an older configuration are dead
int64_t g_threshold = 16384;  // smallest min(m, l, n) mzd_mul_mp spreads over several devices
int g_variant = 0;            // 0 automatic, 1 row slabs, 2 Strassen sub-products (m4ri_amd_set_multi_variant)
int64_t g_seq = 0;            // operations issued so far
std::atomic<int> g_abort{0};  // a worker failed: the others stop waiting for its flags
m4ri_amd_multi_stats g_mstats = {};

// ---- worker pool: thread i issues the work of rank i -------------------------------------------------
struct Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  const std::function<int(int)> *job = nullptr;
  uint64_t gen = 0;
  int pending  = 0;
  bool stop    = false;
  std::vector<int> rc;

  void worker(int i, uint64_t seen) {  // seen: the generation at the time the pool was (re)started -- jobs before it are not this worker's
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen;
      const std::function<int(int)> *f = job;
      lk.unlock();
      const int r = (*f)(i);
      lk.lock();
      rc[(size_t)i] = r;
      if (--pending == 0) cv_done.notify_all();
    }
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (std::thread &t : th)
      if (t.joinable()) t.join();
    th.clear();
    stop = false;
  }
  int start(int n) {
    shutdown();
    rc.assign((size_t)n, 0);
    try {
      for (int i = 0; i < n; ++i) th.emplace_back([this, i, g = gen] { worker(i, g); });
    } catch (...) {
      shutdown();
      return (int)hipErrorOutOfMemory;
    }
    return 0;
  }
  // every worker i runs f(i); returns the first non-zero result
  int run(const std::function<int(int)> &f) {
    std::unique_lock<std::mutex> lk(mu);
    g_abort.store(0);
    job     = &f;
    pending = (int)th.size();
    ++gen;
    cv.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
    job = nullptr;
    for (int r : rc)
      if (r) return r;
    return 0;
  }
  ~Pool() { shutdown(); }
};
Pool g_pool;

// a failure on one rank must not leave the others spinning on its flags
#define RTRY(expr)                                     \
  do {                                                 \
    const int e_ = (int)(expr);                        \
    if (e_ != 0) { g_abort.store(1); return e_; }      \
  } while (0)

int wait_flag(const std::atomic<int64_t> &f, int64_t want) {
  for (int spins = 0; f.load(std::memory_order_acquire) < want; ++spins) {
    if (g_abort.load(std::memory_order_relaxed)) return (int)hipErrorLaunchFailure;
    if (spins > 64) std::this_thread::yield();
  }
  return 0;
}

void default_devices() {
  if (g_devices_set) return;
  g_devices_set = true;
  g_devices.clear();
  if (const char *env = getenv("M4RI_AMD_DEVICES")) {  // e.g. "0,1,2,3" or, for tests on one GPU, "0,0,0"
    for (const char *q = env; *q;) {
      char *end = nullptr;
      const long v = strtol(q, &end, 10);
      if (end == q) break;
      g_devices.push_back((int)v);
      q = (*end == ',') ? end + 1 : end;
    }
    // the same range m4ri_amd_set_devices accepts; a list with any id outside it is ignored as a whole
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess) have = 0;
    bool ok = g_devices.size() <= (size_t)MAX_RANKS;
    for (int id : g_devices) ok = ok && id >= 0 && id < have && id < 16;
    if (!ok) {
      fprintf(stderr, "m4ri_amd: M4RI_AMD_DEVICES=\"%s\" names a device outside 0..%d: ignored, using every visible device\n", env, (have < 16 ? have : 16) - 1);
      g_devices.clear();
    }
  }
  if (g_devices.empty()) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) n = 1;
    if (n > 16) n = 16;
    for (int d = 0; d < n; ++d) g_devices.push_back(d);
  }
}

void drop_lane(Lane &L) {  // streams, events, arenas and bounce buffers of every rank of a lane
  for (auto &rp : L.ranks) {
    Rank &r = *rp;
    (void)hipSetDevice(r.device);
    if (r.st) (void)hipStreamSynchronize(r.st);
    if (r.ci) (void)hipStreamSynchronize(r.ci);
    if (r.co) (void)hipStreamSynchronize(r.co);
    for (auto *v : {&r.lin, &r.lout}) {  // (capped link streams appear several times: destroy each once)
      std::vector<hipStream_t> seen;
      for (hipStream_t ls : *v) {
        bool dup = false;
        for (hipStream_t x : seen) dup = dup || x == ls;
        if (dup) continue;
        seen.push_back(ls);
        (void)hipStreamSynchronize(ls);
        (void)hipStreamDestroy(ls);
      }
    }
    for (auto *v : {&r.ev_lin, &r.ev_lout}) {
      std::vector<hipEvent_t> seen;
      for (hipEvent_t e : *v) {
        bool dup = false;
        for (hipEvent_t x : seen) dup = dup || x == e;
        if (dup) continue;
        seen.push_back(e);
        (void)hipEventDestroy(e);
      }
    }
    if (r.arena) (void)hipFree(r.arena);
    for (word *b : r.bounce)
      if (b) (void)hipHostFree(b);
    for (hipStream_t s : {r.st, r.ci, r.co})
      if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : {r.ev_start, r.ev_down, r.ev_gather, r.ev_first, r.ev_back, r.ev_done[0], r.ev_done[1], r.ev_tl0, r.ev_tl1})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_in) (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_prod) (void)hipEventDestroy(e);
  }
  L.ranks.clear();
  L.parity = 0;
  L.seen_excl = 0;
}

void drop_ranks() {  // every lane; leaves g_ranks empty
  g_pool.shutdown();
  for (Lane &L : g_lane) drop_lane(L);
  g_staged.clear();
  g_pairs_staged = 0;
  ++g_config_gen;
}

int build_lane(Lane &L) {  // one Rank per configured device: its streams and events
  for (size_t i = 0; i < g_devices.size(); ++i) {
    L.ranks.emplace_back(new Rank());
    Rank &r  = *L.ranks.back();
    r.device = g_devices[i];
    HIPTRY(m4ri_amd_init(r.device));  // binds the device + creates its engine
    HIPTRY(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
    HIPTRY(hipStreamCreateWithFlags(&r.ci, hipStreamNonBlocking));
    HIPTRY(hipStreamCreateWithFlags(&r.co, hipStreamNonBlocking));
    // M4RI_AMD_LINK_STREAMS=n caps the link streams per direction (peer k uses stream k % n; default: one per peer): a knob for
    // boxes whose runtime maps many streams onto few hardware queues
    static const int link_cap = getenv("M4RI_AMD_LINK_STREAMS") ? atoi(getenv("M4RI_AMD_LINK_STREAMS")) : 0;
    for (size_t k = 0; k < g_devices.size(); ++k) {
      hipStream_t a = nullptr, b = nullptr;
      hipEvent_t ea = nullptr, eb = nullptr;
      if (link_cap > 0 && k >= (size_t)link_cap) {  // share the stream (and its event) of peer k % n
        r.lin.push_back(r.lin[k % (size_t)link_cap]);
        r.lout.push_back(r.lout[k % (size_t)link_cap]);
        r.ev_lin.push_back(r.ev_lin[k % (size_t)link_cap]);
        r.ev_lout.push_back(r.ev_lout[k % (size_t)link_cap]);
        continue;
      }
      HIPTRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
      r.lin.push_back(a);
      HIPTRY(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
      r.lout.push_back(b);
      HIPTRY(hipEventCreateWithFlags(&ea, hipEventDisableTiming));
      r.ev_lin.push_back(ea);
      HIPTRY(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
      r.ev_lout.push_back(eb);
    }
    r.lin_used.assign(g_devices.size(), 0);
    r.lout_used.assign(g_devices.size(), 0);
    r.bounce.assign(2 * g_devices.size() + 1, nullptr);
    for (hipEvent_t *e : {&r.ev_start, &r.ev_down, &r.ev_gather, &r.ev_first, &r.ev_back, &r.ev_done[0], &r.ev_done[1], &r.ev_tl0, &r.ev_tl1}) HIPTRY(hipEventCreate(e));
  }
  return 0;
}

// Which ordered pairs of ranks copy through the host: pure arithmetic (m4ri_amd_multi_pair_table exports it for the CPU tests).
// can[d * ndev + e] != 0: device d may map device e's memory (hipDeviceCanAccessPeer).  `spec` is the test hook M4RI_AMD_NO_PEER:
// "all", or a list "i-j,k-l" of RANK pairs (both directions).  Ranks on one device always copy directly.
void pair_table(int W, const int *dev, int ndev, const int *can, const char *spec, char *staged) {
  for (int i = 0; i < W; ++i)
    for (int j = 0; j < W; ++j) {
      const int di = dev[i], dj = dev[j];
      staged[i * W + j] = (di != dj && di >= 0 && dj >= 0 && di < ndev && dj < ndev && !can[di * ndev + dj]) ? 1 : 0;
    }
  if (!spec || !*spec) return;
  if (!strcmp(spec, "all")) {
    for (int i = 0; i < W; ++i)
      for (int j = 0; j < W; ++j) staged[i * W + j] = i != j;
    return;
  }
  for (const char *q = spec; *q;) {
    char *end = nullptr;
    const long a = strtol(q, &end, 10);
    if (end == q || *end != '-') break;
    q = end + 1;
    const long b = strtol(q, &end, 10);
    if (end == q) break;
    if (a >= 0 && a < W && b >= 0 && b < W && a != b) staged[a * W + b] = staged[b * W + a] = 1;
    q = (*end == ',') ? end + 1 : end;
  }
}

int copy_words(Rank &R, int me, int slot, word *dst, const word *src, int src_rank, int src_dev, int64_t words, hipStream_t s);

// First contact with a set of DIFFERENT devices: peer access, then one small copy and one cross-device event wait per ordered pair
// -- the two things ranks sharing one GPU never exercise -- so that a node whose links or runtime refuse them fails HERE, with the
// pair and the HIP error named, before any product is scheduled.  (M4RI_AMD_SELFTEST_PAIRS=1 runs it between ranks of one device too.)
int setup_pairs() {
  const int W = (int)g_ranks.size();
  int ndev = 0;
  HIPTRY(hipGetDeviceCount(&ndev));
  std::vector<int> can((size_t)ndev * (size_t)ndev, 1), dev((size_t)W);
  for (int i = 0; i < W; ++i) dev[(size_t)i] = g_ranks[(size_t)i]->device;
  for (int d = 0; d < ndev; ++d)
    for (int e = 0; e < ndev; ++e) {
      int c = 1;
      if (d != e && hipDeviceCanAccessPeer(&c, d, e) != hipSuccess) c = 0;
      can[(size_t)d * ndev + e] = c;
    }
  g_staged.assign((size_t)W * (size_t)W, 0);
  pair_table(W, dev.data(), ndev, can.data(), getenv("M4RI_AMD_NO_PEER"), g_staged.data());
  g_pairs_staged = 0;
  for (char c : g_staged) g_pairs_staged += c ? 1 : 0;
  bool distinct = false;
  for (int i = 0; i < W; ++i) {
    Rank &r = *g_ranks[(size_t)i];
    HIPTRY(hipSetDevice(r.device));
    for (int k = 0; k < W; ++k) {  // direct xGMI copies between every pair that allows them
      const int other = g_ranks[(size_t)k]->device;
      if (other == r.device) continue;
      distinct = true;
      if (!can[(size_t)r.device * ndev + other]) continue;
      const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
        fprintf(stderr, "m4ri_amd: hipDeviceEnablePeerAccess(device %d -> device %d) failed: %s\n", r.device, other, hipGetErrorString(e));
        return (int)e;
      }
      (void)hipGetLastError();
    }
  }
  const bool forced = getenv("M4RI_AMD_SELFTEST_PAIRS") && atoi(getenv("M4RI_AMD_SELFTEST_PAIRS")) != 0;
  if (!distinct && !forced) return 0;
  constexpr int64_t WORDS = 1 << 17;  // 1 MiB
  std::vector<word *> srcb((size_t)W, nullptr), dstb((size_t)W, nullptr);
  std::vector<word> host((size_t)WORDS);
  int rc = 0, bad_i = -1, bad_j = -1;
  const char *what = "";
  auto fail = [&](int i, int j, const char *w, hipError_t e) { rc = e == hipSuccess ? (int)hipErrorUnknown : (int)e; bad_i = i; bad_j = j; what = w; };
  for (int i = 0; i < W && !rc; ++i) {
    Rank &r = *g_ranks[(size_t)i];
    hipError_t e = hipSetDevice(r.device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&srcb[(size_t)i]), WORDS * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dstb[(size_t)i]), WORDS * 8);
    if (e != hipSuccess) fail(i, i, "hipMalloc", e);
  }
  for (int i = 0; i < W && !rc; ++i)
    for (int j = 0; j < W && !rc; ++j) {
      Rank &D = *g_ranks[(size_t)i], &S = *g_ranks[(size_t)j];
      if (i == j || (D.device == S.device && !forced)) continue;
      const int pattern = 0x11 * ((i * W + j) % 13 + 1);
      hipError_t e = hipSetDevice(S.device);
      if (e == hipSuccess) e = hipMemsetAsync(srcb[(size_t)j], pattern, WORDS * 8, S.st);
      if (e == hipSuccess) e = hipEventRecord(S.ev_start, S.st);
      if (e != hipSuccess) { fail(i, j, "fill on the source device", e); break; }
      e = hipSetDevice(D.device);
      if (e == hipSuccess) e = hipStreamWaitEvent(D.st, S.ev_start, 0);  // the cross-device wait every schedule relies on
      if (e != hipSuccess) { fail(i, j, "hipStreamWaitEvent on another device's event", e); break; }
      if (int c = copy_words(D, i, 2 * W, dstb[(size_t)i], srcb[(size_t)j], j, S.device, WORDS, D.st)) { fail(i, j, g_staged[(size_t)i * W + j] ? "host-staged copy" : "hipMemcpyPeerAsync", (hipError_t)c); break; }
      e = hipMemcpyAsync(host.data(), dstb[(size_t)i], WORDS * 8, hipMemcpyDeviceToHost, D.st);
      if (e == hipSuccess) e = hipStreamSynchronize(D.st);
      if (e != hipSuccess) { fail(i, j, "copy / synchronise", e); break; }
      word want = 0;
      memset(&want, pattern, 8);
      for (int64_t k = 0; k < WORDS; k += 4097)
        if (host[(size_t)k] != want) { fail(i, j, "the copied data is wrong", hipErrorUnknown); break; }
    }
  for (int i = 0; i < W; ++i) {
    (void)hipSetDevice(g_ranks[(size_t)i]->device);
    if (srcb[(size_t)i]) (void)hipFree(srcb[(size_t)i]);
    if (dstb[(size_t)i]) (void)hipFree(dstb[(size_t)i]);
  }
  if (rc)
    fprintf(stderr, "m4ri_amd: multi-device self-test FAILED for rank %d (device %d) <- rank %d (device %d): %s: %s (hipError %d); no product was scheduled\n",
            bad_i, bad_i >= 0 ? g_ranks[(size_t)bad_i]->device : -1, bad_j, bad_j >= 0 ? g_ranks[(size_t)bad_j]->device : -1, what,
            hipGetErrorString((hipError_t)rc), rc);
  return rc;
}

int ensure_ranks() {
  default_devices();
  if (g_ranks.size() == g_devices.size() && !g_ranks.empty()) {
    bool same = true;
    for (size_t i = 0; i < g_ranks.size(); ++i) same = same && g_ranks[i]->device == g_devices[i];
    if (same) return 0;
  }
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  drop_ranks();
  // a failure half way must not leave ranks that LOOK configured nor another current device behind: on any error
  // everything made so far is dropped and the caller's device restored
  int rc = build_lane(g_lane[0]);
  if (!rc) rc = setup_pairs();
  if (!rc) rc = g_pool.start((int)g_ranks.size());
  if (rc) drop_ranks();
  (void)hipSetDevice(cur);
  return rc;
}

int ensure_lane(int lane) {  // lane 1 is made on first use
  if (lane == 0 || g_lane[lane].ranks.size() == g_ranks.size()) return 0;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  drop_lane(g_lane[lane]);
  const int rc = build_lane(g_lane[lane]);
  if (rc) drop_lane(g_lane[lane]);
  (void)hipSetDevice(cur);
  return rc;
}

int unit_events(Rank &r, size_t n) {
  while (r.ev_in.size() < n) { hipEvent_t e; HIPTRY(hipEventCreate(&e)); r.ev_in.push_back(e); }
  while (r.ev_prod.size() < n) { hipEvent_t e; HIPTRY(hipEventCreate(&e)); r.ev_prod.push_back(e); }
  return 0;
}

// device-to-device copy of `words` words into rank `me` (which owns stream `s`; `slot` names that stream: 0..W-1 its inbound link
// streams, W..2W-1 the outbound ones, 2W the compute stream) from rank `src_rank`: inside a device, over the link of the pair, or --
// pairs without peer access -- through the slot's pinned bounce buffer, D2H then H2D in order on the stream
constexpr int64_t BOUNCE_WORDS = 1 << 21;  // 16 MiB
int copy_words(Rank &R, int me, int slot, word *dst, const word *src, int src_rank, int src_dev, int64_t words, hipStream_t s) {
  if (words <= 0) return 0;
  if (R.device == src_dev && !(me != src_rank && !g_staged.empty() && g_staged[(size_t)me * g_ranks.size() + (size_t)src_rank]))
    return (int)hipMemcpyAsync(dst, src, (size_t)words * 8, hipMemcpyDeviceToDevice, s);
  if (g_staged.empty() || !g_staged[(size_t)me * g_ranks.size() + (size_t)src_rank]) return (int)hipMemcpyPeerAsync(dst, R.device, src, src_dev, (size_t)words * 8, s);
  if ((size_t)slot >= R.bounce.size()) return (int)hipErrorInvalidValue;
  if (!R.bounce[(size_t)slot]) HIPTRY(hipHostMalloc(reinterpret_cast<void **>(&R.bounce[(size_t)slot]), (size_t)BOUNCE_WORDS * 8, hipHostMallocDefault));
  word *b = R.bounce[(size_t)slot];
  for (int64_t o = 0; o < words; o += BOUNCE_WORDS) {
    const int64_t k = (words - o) < BOUNCE_WORDS ? (words - o) : BOUNCE_WORDS;
    HIPTRY(hipMemcpyAsync(b, src + o, (size_t)k * 8, hipMemcpyDeviceToHost, s));
    HIPTRY(hipMemcpyAsync(dst + o, b, (size_t)k * 8, hipMemcpyHostToDevice, s));
  }
  return 0;
}

// ---- layouts ---------------------------------------------------------------------------------------
struct Run { int64_t g0, rows, l0; };  // global first row, valid rows, local first row

int64_t padded(int64_t x) { return roundup(x > 0 ? x : 1, PAD_BITS); }
int layout_levels(int layout) { return layout == M4RI_AMD_LAYOUT_CYCLIC1 ? 1 : layout == M4RI_AMD_LAYOUT_CYCLIC2 ? 2 : 0; }

// rows of rank `rank`'s local buffer (padding included)
int64_t local_rows_of(int layout, int world, int rank, int64_t rows) {
  if (layout == M4RI_AMD_LAYOUT_REPLICATED) return rows;
  if (layout == M4RI_AMD_LAYOUT_ROWS) return (rows + world - 1) / world;
  const int64_t S = 1ll << layout_levels(layout), brows = padded(rows) / S;
  return S * (cut_of(brows, world, rank + 1) - cut_of(brows, world, rank));
}

// the runs of VALID global rows rank `rank` holds, in local order
int runs_of(int layout, int world, int rank, int64_t rows, Run *out) {
  int n = 0;
  if (layout == M4RI_AMD_LAYOUT_REPLICATED) {
    if (rows > 0) out[n++] = Run{0, rows, 0};
  } else if (layout == M4RI_AMD_LAYOUT_ROWS) {
    const int64_t k = (rows + world - 1) / world, g0 = k * rank;
    if (g0 < rows) out[n++] = Run{g0, (rows - g0) < k ? (rows - g0) : k, 0};
  } else {
    const int64_t S = 1ll << layout_levels(layout), brows = padded(rows) / S;
    const int64_t c0 = cut_of(brows, world, rank), s = cut_of(brows, world, rank + 1) - c0;
    for (int64_t b = 0; b < S; ++b) {
      const int64_t g0 = b * brows + c0;
      int64_t valid    = rows - g0;
      if (valid > s) valid = s;
      if (valid > 0) out[n++] = Run{g0, valid, b * s};
    }
  }
  return n;
}

}  // namespace

struct m4ri_amd_dmat {
  int64_t rows = 0, ncols = 0, stride = 0;  // stride: words of a local row = padded(ncols) / 64, the same in every layout
  int layout = 0, world = 0;
  uint64_t config_gen = 0;
  std::vector<int> device;
  std::vector<word *> local;
  std::vector<int64_t> lrows;  // rows of the local buffers
};

namespace {

bool alive(const m4ri_amd_dmat *d) { return d && d->config_gen == g_config_gen && d->world == (int)g_ranks.size(); }

m4ri_amd_dmat *dmat_new(int64_t rows, int64_t ncols, int layout) {  // g_multi_mu held, ranks configured
  if (rows < 0 || ncols < 0 || layout < 0 || layout > M4RI_AMD_LAYOUT_REPLICATED) return nullptr;
  const int W = (int)g_ranks.size();
  std::unique_ptr<m4ri_amd_dmat> d(new m4ri_amd_dmat());
  d->rows = rows; d->ncols = ncols; d->layout = layout; d->world = W; d->config_gen = g_config_gen;
  d->stride = padded(ncols) / 64;
  d->device.resize((size_t)W); d->local.assign((size_t)W, nullptr); d->lrows.resize((size_t)W);
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return nullptr;
  bool ok = true;
  for (int r = 0; r < W && ok; ++r) {
    d->device[(size_t)r] = g_ranks[(size_t)r]->device;
    d->lrows[(size_t)r]  = local_rows_of(layout, W, r, rows);
    const size_t bytes   = (size_t)(d->lrows[(size_t)r] > 0 ? d->lrows[(size_t)r] : 1) * (size_t)d->stride * 8;
    // the padding is zero from here on: every operation keeps it so.  The clear runs on the NULL stream, which the ranks' non-blocking
    // streams do not wait for: it has to be complete before the handle exists
    ok = hipSetDevice(d->device[(size_t)r]) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&d->local[(size_t)r]), bytes) == hipSuccess &&
         hipMemset(d->local[(size_t)r], 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  }
  (void)hipSetDevice(cur);
  if (!ok) {
    for (int r = 0; r < W; ++r)
      if (d->local[(size_t)r]) { (void)hipSetDevice(d->device[(size_t)r]); (void)hipFree(d->local[(size_t)r]); }
    (void)hipSetDevice(cur);
    return nullptr;
  }
  return d.release();
}

int sync_all();

void dmat_delete(m4ri_amd_dmat *d) {
  if (!d) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  // other ranks may still be pulling rows out of this matrix's buffers (their part of an operation issued a moment ago, on THEIR
  // devices): everybody's streams first, then the buffers go
  if (d->config_gen == g_config_gen) (void)sync_all();
  for (size_t r = 0; r < d->local.size(); ++r)
    if (d->local[r]) {
      (void)hipSetDevice(d->device[r]);
      (void)hipDeviceSynchronize();
      (void)hipFree(d->local[r]);
    }
  (void)hipSetDevice(cur);
  delete d;
}

// Every operation of a lane starts behind the lane's previous operation on EVERY rank (their reads of what it will overwrite, their
// writes of what it will read) and ends with ev_done on the compute stream; the copy streams start behind ev_start.  Across lanes:
// products do not wait for each other (the caller of m4ri_amd_dmat_mul_lane promises independent results); everything else
// ("exclusive": uploads, fills, conversions, downloads) waits for the last operation of every lane, and the first operation of a lane
// after an exclusive one elsewhere waits for that lane.  begin_op / end_op run on the issuing thread (g_multi_mu held) around
// Pool::run; op_begin / op_end on the ranks' threads in between.
struct OpCtx { int lane; bool cross; };

OpCtx begin_op(int lane, bool exclusive) {
  Lane &L = g_lane[lane];
  OpCtx c{lane, exclusive || (g_excl_id > L.seen_excl && g_excl_lane != lane)};
  if (exclusive) { ++g_excl_id; g_excl_lane = lane; }
  L.seen_excl = g_excl_id;
  return c;
}
int op_begin(Rank &R, const OpCtx &c) {
  for (int x = 0; x < NUM_LANES; ++x) {
    if (x != c.lane && !c.cross) continue;
    const int prev = g_lane[x].parity ^ 1;  // what the lane's latest operation recorded (its threads have all returned)
    for (auto &other : g_lane[x].ranks)
      if (other->done_recorded[prev]) HIPTRY(hipStreamWaitEvent(R.st, other->ev_done[prev], 0));
  }
  HIPTRY(hipEventRecord(R.ev_start, R.st));
  HIPTRY(hipStreamWaitEvent(R.ci, R.ev_start, 0));
  HIPTRY(hipStreamWaitEvent(R.co, R.ev_start, 0));
  for (hipStream_t ls : R.lin) HIPTRY(hipStreamWaitEvent(ls, R.ev_start, 0));
  for (hipStream_t ls : R.lout) HIPTRY(hipStreamWaitEvent(ls, R.ev_start, 0));
  return 0;
}
// the join stream collects what the link streams have been given since the last join: afterwards an event recorded on it means
// "all of those copies are done"
int join_links(Rank &R, bool inbound) {
  std::vector<hipStream_t> &ls = inbound ? R.lin : R.lout;
  std::vector<hipEvent_t> &ev  = inbound ? R.ev_lin : R.ev_lout;
  std::vector<char> &used      = inbound ? R.lin_used : R.lout_used;
  for (size_t k = 0; k < ls.size(); ++k)
    if (used[k]) {
      HIPTRY(hipEventRecord(ev[k], ls[k]));
      HIPTRY(hipStreamWaitEvent(inbound ? R.ci : R.co, ev[k], 0));
      used[k] = 0;
    }
  return 0;
}
int op_end(Rank &R, const OpCtx &c) {
  HIPTRY(hipEventRecord(R.ev_done[g_lane[c.lane].parity], R.st));
  return 0;
}
// after the pool has joined: the slot the operation recorded is now every rank's "previous operation"
void end_op(const OpCtx &c) {
  Lane &L = g_lane[c.lane];
  for (auto &r : L.ranks) r->done_recorded[L.parity] = true;
  L.parity ^= 1;
}
// one operation: every rank's thread runs f(rank)
int run_op(int lane, bool exclusive, const std::function<int(int, const OpCtx &)> &f) {
  const OpCtx c = begin_op(lane, exclusive);
  const int rc  = g_pool.run([&](int me) { return f(me, c); });
  end_op(c);
  return rc;
}

int carve(Rank &r, const int64_t *words) {
  size_t need = 0;
  for (int k = 0; k < B_COUNT; ++k) need += (size_t)pad32(words[k] > 0 ? words[k] : 0);
  if (need > r.cap) {
    // the old arena goes away: OTHER ranks may still be pulling slabs out of it (their part of the previous operation, on their
    // devices), so everybody's previous operation has to be over, not just this device's
    for (Lane &L : g_lane)
      for (auto &other : L.ranks)
        for (int k = 0; k < 2; ++k)
          if (other->done_recorded[k]) HIPTRY(hipEventSynchronize(other->ev_done[k]));
    HIPTRY(hipDeviceSynchronize());
    if (r.arena) HIPTRY(hipFree(r.arena));
    r.arena = nullptr; r.cap = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&r.arena), (need > 0 ? need : 32) * 8));
    r.cap = need;
  }
  word *q = r.arena;
  for (int k = 0; k < B_COUNT; ++k) { r.buf[k] = q; q += pad32(words[k] > 0 ? words[k] : 0); }
  return 0;
}

// ---- redistribution: dst <- src, any two layouts of one shape (also the all-gather: ROWS -> REPLICATED) --------------
int redistribute_rank(const m4ri_amd_dmat *dst, const m4ri_amd_dmat *src, int me, const OpCtx &ctx) {
  Rank &R = *g_lane[ctx.lane].ranks[(size_t)me];
  RTRY(hipSetDevice(R.device));
  RTRY(op_begin(R, ctx));
  const int W = dst->world;
  Run dr[4], sr[4];
  const int nd = runs_of(dst->layout, W, me, dst->rows, dr);
  for (int i = 0; i < nd; ++i) {
    // walk the source holders starting at this rank: a replicated source is read locally, slabs spread over the links
    for (int k = 0; k < W; ++k) {
      const int s  = (me + k) % W;
      const int ns = runs_of(src->layout, W, s, src->rows, sr);
      for (int j = 0; j < ns; ++j) {
        const int64_t lo = dr[i].g0 > sr[j].g0 ? dr[i].g0 : sr[j].g0;
        const int64_t hi = (dr[i].g0 + dr[i].rows) < (sr[j].g0 + sr[j].rows) ? (dr[i].g0 + dr[i].rows) : (sr[j].g0 + sr[j].rows);
        if (hi <= lo) continue;
        RTRY(copy_words(R, me, 2 * W, dst->local[(size_t)me] + (dr[i].l0 + lo - dr[i].g0) * dst->stride,
                        src->local[(size_t)s] + (sr[j].l0 + lo - sr[j].g0) * src->stride, s, src->device[(size_t)s], (hi - lo) * dst->stride, R.st));
      }
      if (src->layout == M4RI_AMD_LAYOUT_REPLICATED) break;  // the first holder had everything
    }
  }
  return op_end(R, ctx);
}

int redistribute(m4ri_amd_dmat *dst, const m4ri_amd_dmat *src) {
  if (!alive(dst) || !alive(src) || dst->rows != src->rows || dst->ncols != src->ncols) return (int)hipErrorInvalidValue;
  if (dst == src) return 0;
  return run_op(0, true, [&](int me, const OpCtx &ctx) { return redistribute_rank(dst, src, me, ctx); });
}

// ---- schedule 1: row slabs -----------------------------------------------------------------------------
// C_r (+)= A_r * B.  A and C in ROWS layout; B in ROWS layout (gathered here, the gather under the product with the own
// slab) or REPLICATED (nothing to move).
struct SlabOp {
  m4ri_amd_dmat *C; const m4ri_amd_dmat *A, *B;
  int add, cutoff;
  bool overlap;
};

int slabs_rank(const SlabOp &op, int me, const OpCtx &ctx) {
  Rank &R = *g_lane[ctx.lane].ranks[(size_t)me];
  RTRY(hipSetDevice(R.device));
  const int W      = op.A->world;
  const int64_t l  = op.A->ncols, n = op.B->ncols, m = op.A->rows;
  const int64_t ka = (m + W - 1) / W, kb = (l + W - 1) / W;
  const int64_t r0 = ka * me < m ? ka * me : m, mr = (m - r0) < ka ? (m - r0) : ka;  // my rows of A and C
  const int64_t sa = op.A->stride, sbw = op.B->stride, sc = op.C->stride;
  int64_t words[B_COUNT] = {};
  const bool gathered = op.B->layout == M4RI_AMD_LAYOUT_ROWS;
  if (gathered) words[B_GATHER] = (int64_t)W * kb * sbw;
  RTRY(carve(R, words));
  RTRY(op_begin(R, ctx));
  RTRY(hipEventRecord(R.ev_tl0, R.st));
  R.tl_valid = false;
  word *Cme = op.C->local[(size_t)me];
  const word *Ame = op.A->local[(size_t)me];
  R.tl_strassen = false; R.tl_gathered = gathered; R.tl_units = 0;
  if (!gathered) {
    if (mr > 0) RTRY(m4ri_amd_mul_dev(Cme, sc, Ame, sa, op.B->local[(size_t)me], sbw, mr, l, n, op.add, op.cutoff, R.st));
    RTRY(hipEventRecord(R.ev_first, R.st));
    RTRY(hipEventRecord(R.ev_gather, R.ci));
    RTRY(hipEventRecord(R.ev_tl1, R.st));
    R.tl_valid = true;
    return op_end(R, ctx);
  }
  word *Bfull      = R.buf[B_GATHER];
  const int64_t k0 = kb * me < l ? kb * me : l, k1 = (k0 + kb) < l ? (k0 + kb) : l;  // my rows of B
  // the ONE exchange of the schedule: the other ranks' slabs of B, pulled over the link of each pair on the inbound stream,
  // nearest neighbour first so that the W - 1 links of a rank are all in use
  for (int k = op.overlap ? 1 : 0; k < W; ++k) {
    const int s      = (me + k) % W;
    const int64_t g0 = kb * s < l ? kb * s : l, g1 = (g0 + kb) < l ? (g0 + kb) : l;
    if (g1 > g0) {
      RTRY(copy_words(R, me, s, Bfull + g0 * sbw, op.B->local[(size_t)s], s, op.B->device[(size_t)s], (g1 - g0) * sbw, R.lin[(size_t)s]));
      R.lin_used[(size_t)s] = 1;
    }
  }
  RTRY(join_links(R, true));
  RTRY(hipEventRecord(R.ev_gather, R.ci));
  if (op.overlap) {
    // own slab first (resident: nothing to wait for), then what lies before and after it in the gathered B
    bool first = true;
    if (mr > 0 && k1 > k0) {
      RTRY(m4ri_amd_mul_dev(Cme, sc, Ame + k0 / 64, sa, op.B->local[(size_t)me], sbw, mr, k1 - k0, n, op.add, op.cutoff, R.st));
      first = false;
    }
    RTRY(hipEventRecord(R.ev_first, R.st));
    RTRY(hipStreamWaitEvent(R.st, R.ev_gather, 0));
    if (mr > 0 && k0 > 0) {
      RTRY(m4ri_amd_mul_dev(Cme, sc, Ame, sa, Bfull, sbw, mr, k0, n, (op.add || !first) ? 1 : 0, op.cutoff, R.st));
      first = false;
    }
    if (mr > 0 && l > k1) {
      RTRY(m4ri_amd_mul_dev(Cme, sc, Ame + k1 / 64, sa, Bfull + k1 * sbw, sbw, mr, l - k1, n, (op.add || !first) ? 1 : 0, op.cutoff, R.st));
      first = false;
    }
    if (mr > 0 && first && !op.add) RTRY(hipMemsetAsync(Cme, 0, (size_t)mr * (size_t)sc * 8, R.st));  // l == 0
  } else {
    RTRY(hipEventRecord(R.ev_first, R.st));
    RTRY(hipStreamWaitEvent(R.st, R.ev_gather, 0));
    if (mr > 0) RTRY(m4ri_amd_mul_dev(Cme, sc, Ame, sa, Bfull, sbw, mr, l, n, op.add, op.cutoff, R.st));
  }
  RTRY(hipEventRecord(R.ev_tl1, R.st));
  R.tl_valid = true;
  return op_end(R, ctx);
}

// ---- schedule 2: the sub-products of the top Strassen-Winograd level(s) ------------------------------------------
struct StrassenOp {
  m4ri_amd_dmat *C; const m4ri_amd_dmat *A, *B;
  m4ri_amd_shard_plan p;
  int add, cutoff, chunks;
  int64_t seq;
  int group;  // rounds per batched product: the sub-products a rank owns are multiplied `group` at a time (m4ri_amd_mul_batch_dev)
};

// row chunk c of a sub-product = the slabs of ranks [lo, hi): rows [cut(lo), cut(hi)) of its A operand and of its result
void chunk_of(const StrassenOp &op, int c, int *lo, int *hi) {
  *lo = op.p.world * c / op.chunks;
  *hi = op.p.world * (c + 1) / op.chunks;
}

int strassen_rank(const StrassenOp &op, int me, const OpCtx &ctx) {
  std::vector<std::unique_ptr<Rank>> &ranks = g_lane[ctx.lane].ranks;
  Rank &R = *ranks[(size_t)me];
  RTRY(hipSetDevice(R.device));
  const m4ri_amd_shard_plan &p = op.p;
  const int W = p.world, nch = op.chunks, rounds = (p.nprod + W - 1) / W;
  int64_t words[B_COUNT] = {};
  words[B_CHILD_A] = m4ri_amd_shard_buffer_words(&p, me, M4RI_AMD_SHARD_BUF_CHILD_A);
  words[B_CHILD_B] = m4ri_amd_shard_buffer_words(&p, me, M4RI_AMD_SHARD_BUF_CHILD_B);
  words[B_SLABS_P] = m4ri_amd_shard_buffer_words(&p, me, M4RI_AMD_SHARD_BUF_SLABS_P);
  words[B_OPER_A]  = m4ri_amd_shard_buffer_words(&p, me, M4RI_AMD_SHARD_BUF_OPER_A);
  words[B_OPER_B]  = m4ri_amd_shard_buffer_words(&p, me, M4RI_AMD_SHARD_BUF_OPER_B);
  words[B_PROD]    = m4ri_amd_shard_buffer_words(&p, me, M4RI_AMD_SHARD_BUF_PROD);
  RTRY(carve(R, words));
  RTRY(unit_events(R, (size_t)rounds * (size_t)nch));
  RTRY(op_begin(R, ctx));
  RTRY(hipEventRecord(R.ev_tl0, R.st));
  R.tl_valid = false;
  R.tl_strassen = true; R.tl_gathered = false; R.tl_units = 0;
  // 1. the level's operand additions on my slabs: no communication
  RTRY(m4ri_amd_shard_down_dev(&p, me, op.A->local[(size_t)me], op.A->stride, op.B->local[(size_t)me], op.B->stride, R.buf[B_CHILD_A], R.buf[B_CHILD_B], R.st));
  RTRY(hipEventRecord(R.ev_down, R.st));
  R.flag_down.store(op.seq, std::memory_order_release);
  // 2. my sub-products, round by round, row chunk by row chunk: the chunk's operand slabs are pulled on the inbound stream
  //    (every holder's down pass is a HIP event; its host flag says the event is recorded), so chunk c + 1 travels while chunk
  //    c is multiplied
  bool waited[MAX_RANKS] = {};
  auto pull_operand = [&](int side, int j, int r) -> int {
    m4ri_amd_shard_piece pc;
    m4ri_amd_shard_piece_of(&p, side, j, r, &pc);
    if (pc.words == 0) return 0;
    Rank &H = *ranks[(size_t)r];
    if (!waited[r]) {
      if (int e = wait_flag(H.flag_down, op.seq)) return e;
      HIPTRY(hipStreamWaitEvent(R.lin[(size_t)r], H.ev_down, 0));
      waited[r] = true;
    }
    R.lin_used[(size_t)r] = 1;
    return copy_words(R, me, r, R.buf[side ? B_OPER_B : B_OPER_A] + pc.owner_off, H.buf[side ? B_CHILD_B : B_CHILD_A] + pc.holder_off, r, H.device, pc.words,
                      R.lin[(size_t)r]);
  };
  int units = 0;
  const int G = (nch == 1 && op.group > 1) ? op.group : 1;
  for (int q = 0; q < rounds && G > 1; q += G) {
    // `G` of my sub-products at a time, as ONE batched product: every pass and the leaf launch shared, so that sub-products whose own
    // leaves leave the chip's last round of tiles half empty fill it together (6 x 16384^3 on a rank of 8: 9 rounds of tiles, not 12).
    // The operands of group g + 1 travel while group g is multiplied.
    int cnt = 0;
    for (int k = 0; k < G && q + k < rounds; ++k)
      if ((q + k) * W + me < p.nprod) ++cnt;
    if (cnt == 0) break;
    for (int k = 0; k < cnt; ++k) {
      const int j = (q + k) * W + me;
      for (int x = 0; x < W; ++x) RTRY(pull_operand(1, j, (me + x) % W));  // own slab first
      for (int x = 0; x < W; ++x) RTRY(pull_operand(0, j, (me + x) % W));
    }
    RTRY(join_links(R, true));
    RTRY(hipEventRecord(R.ev_in[(size_t)units], R.ci));
    RTRY(hipStreamWaitEvent(R.st, R.ev_in[(size_t)units], 0));
    RTRY(m4ri_amd_mul_batch_dev(R.buf[B_PROD] + (int64_t)q * p.bm * p.cwn, p.cwn, p.bm * p.cwn, R.buf[B_OPER_A] + (int64_t)q * p.bm * p.cwl, p.cwl, p.bm * p.cwl,
                                R.buf[B_OPER_B] + (int64_t)q * p.bl * p.cwn, p.cwn, p.bl * p.cwn, p.bm, p.bl, p.cwn * 64, cnt, 0, op.cutoff, R.st));
    RTRY(hipEventRecord(R.ev_prod[(size_t)units], R.st));
    R.flag_prod.store(op.seq * 4096 + units + 1, std::memory_order_release);
    ++units;
  }
  for (int q = 0; q < rounds && G == 1; ++q) {
    const int j = q * W + me;
    if (j >= p.nprod) break;
    for (int c = 0; c < nch; ++c, ++units) {
      int lo, hi;
      chunk_of(op, c, &lo, &hi);
      if (c == 0)
        for (int k = 0; k < W; ++k) RTRY(pull_operand(1, j, (me + k) % W));  // the whole B operand, own slab first
      for (int r = lo; r < hi; ++r) RTRY(pull_operand(0, j, r));
      RTRY(join_links(R, true));
      RTRY(hipEventRecord(R.ev_in[(size_t)units], R.ci));
      RTRY(hipStreamWaitEvent(R.st, R.ev_in[(size_t)units], 0));
      const int64_t row0 = cut_of(p.bm, W, lo), rows = cut_of(p.bm, W, hi) - row0;
      if (rows > 0)
        RTRY(m4ri_amd_mul_dev(R.buf[B_PROD] + ((int64_t)q * p.bm + row0) * p.cwn, p.cwn, R.buf[B_OPER_A] + ((int64_t)q * p.bm + row0) * p.cwl, p.cwl,
                              R.buf[B_OPER_B] + (int64_t)q * p.bl * p.cwn, p.cwn, rows, p.bl, p.cwn * 64, 0, op.cutoff, R.st));
      RTRY(hipEventRecord(R.ev_prod[(size_t)units], R.st));
      R.flag_prod.store(op.seq * 4096 + units + 1, std::memory_order_release);
    }
  }
  R.tl_units = units;
  // 3. my slab of every product, pulled from its owner on the outbound stream as soon as the chunk that holds it is done,
  //    then the level's result additions on my slabs
  int myc = 0;
  for (int c = 0; c < nch; ++c) {
    int lo, hi;
    chunk_of(op, c, &lo, &hi);
    if (me >= lo && me < hi) myc = c;
  }
  for (int j = 0; j < p.nprod; ++j) {
    m4ri_amd_shard_piece pc;
    m4ri_amd_shard_piece_of(&p, 2, j, me, &pc);
    if (pc.words == 0) continue;
    Rank &O     = *ranks[(size_t)pc.owner];
    const int u = G > 1 ? (j / W) / G : (j / W) * nch + myc;  // the owner's unit that holds it
    RTRY(wait_flag(O.flag_prod, op.seq * 4096 + u + 1));
    RTRY(hipStreamWaitEvent(R.lout[(size_t)pc.owner], O.ev_prod[(size_t)u], 0));
    RTRY(copy_words(R, me, W + pc.owner, R.buf[B_SLABS_P] + pc.holder_off, O.buf[B_PROD] + pc.owner_off, pc.owner, O.device, pc.words, R.lout[(size_t)pc.owner]));
    R.lout_used[(size_t)pc.owner] = 1;
  }
  RTRY(join_links(R, false));
  RTRY(hipEventRecord(R.ev_back, R.co));
  RTRY(hipStreamWaitEvent(R.st, R.ev_back, 0));
  RTRY(m4ri_amd_shard_up_dev(&p, me, R.buf[B_SLABS_P], op.C->local[(size_t)me], op.C->stride, op.add, R.st));
  RTRY(hipEventRecord(R.ev_tl1, R.st));
  R.tl_valid = true;
  return op_end(R, ctx);
}

// How many of a rank's sub-products go into one batched product.  The engine's time model prices a batch of b sub-products
// (m4ri_amd_model_seconds_batch: tiles in rounds of 256, so two half-filled last rounds become one full one); of the group sizes whose
// total is within 5 % of the best the SMALLEST wins -- more groups = more of the operand and result transport under multiplications,
// and the model's own noise is a few per cent (6 x 16384^3 on one MI355X: one at a time 3.54 ms, 2 + 2 + 2 3.31, 3 + 3 3.36, all six 3.33;
// the single 32768^3 of the 7-way split 3.81 -- profiles/r06_rank_batch_timing.log).
// One sub-product per rank (7 on 8 ranks) or a caller's cutoff: no grouping.
int pick_group(const m4ri_amd_shard_plan &p, int cutoff) {
  const int rounds = (p.nprod + p.world - 1) / p.world;
  if (rounds < 2 || cutoff != 0) return 1;
  double cost[65], best = 1e300;
  const int gmax = rounds < 64 ? rounds : 64;
  for (int g = 1; g <= gmax; ++g) {
    const int full = rounds / g, rest = rounds % g;
    cost[g] = full * m4ri_amd_model_seconds_batch(p.bm, p.bl, p.cwn * 64, -1, g) + (rest ? m4ri_amd_model_seconds_batch(p.bm, p.bl, p.cwn * 64, -1, rest) : 0.0);
    if (cost[g] < best) best = cost[g];
  }
  for (int g = 1; g <= gmax; ++g)
    if (cost[g] <= 1.05 * best) return g;
  return 1;
}

// the plan of a product on CYCLIC-v operands: every dimension padded to 256 bits (the layouts' own padding)
void plan_for(m4ri_amd_shard_plan *p, int world, int64_t m, int64_t l, int64_t n, int levels) {
  memset(p, 0, sizeof *p);
  p->world = world; p->levels = levels; p->blocks = 1 << levels;
  p->m = m; p->l = l; p->n = n;
  p->M = padded(m); p->L = padded(l); p->N = padded(n);
  p->bm = p->M / p->blocks; p->bl = p->L / p->blocks;
  p->cwl = p->L / p->blocks / 64; p->cwn = p->N / p->blocks / 64;
  p->nprod = nprod_of(world, levels, p->bm, p->bl, p->cwl, p->cwn);
}

// sharding.default_variant: row slabs on 2 ranks (they share ONE link, which the Strassen exchange would saturate) and for every
// product whose halves a sharded Strassen level would leave too thin to pay for the exchange of operands and results (m/2 < 4096,
// l/2 < 8192 or n/2 < 4096; BASELINE.json configs[4], 131072 x 8192 x 131072, has l/2 < 8192: row slabs of A and C with B replicated,
// no reduction); the Strassen sub-products from 5 ranks on -- and on 3 and 4 ranks where the two sharded levels are the
// scheme's 47 sub-products of at least 16384 on every side (65536^3 on 4 ranks: 12 sub-products of 16384^3 per rank in batched products
// 6.58 ms against the row slab's 16384 x 65536 x 65536 in 8.00 ms, one MI355X, profiles/r06_rank_batch_timing.log; below that size
// nothing was measured and the slabs stay)
int default_variant(int world, int64_t m, int64_t l, int64_t n) {
  if (world <= 2) return M4RI_AMD_VARIANT_SLABS;
  if (m / 2 < 4096 || l / 2 < 8192 || n / 2 < 4096) return M4RI_AMD_VARIANT_SLABS;
  if (world <= 4) {
    m4ri_amd_shard_plan p;
    const bool big = m / 4 >= 16384 && l / 4 >= 16384 && n / 4 >= 16384;
    return (big && m4ri_amd_shard_plan_make(&p, world, m, l, n, 2) == 0 && p.nprod != 49) ? M4RI_AMD_VARIANT_STRASSEN : M4RI_AMD_VARIANT_SLABS;
  }
  return M4RI_AMD_VARIANT_STRASSEN;
}

int auto_levels(int world, int64_t m, int64_t l, int64_t n) {
  m4ri_amd_shard_plan p;
  return m4ri_amd_shard_plan_make(&p, world, m, l, n, 0) ? 1 : p.levels;
}

// C (+)= A*B on distributed operands (g_multi_mu held).  Operands in another layout than the schedule's are converted
// through temporaries (correct, not fast: keep matrices in the layout m4ri_amd_multi_layout_for names).
int dmat_mul(m4ri_amd_dmat *C, const m4ri_amd_dmat *A, const m4ri_amd_dmat *B, int add, int cutoff, int variant, int lane = 0) {
  if (!alive(C) || !alive(A) || !alive(B) || A->ncols != B->rows || C->rows != A->rows || C->ncols != B->ncols || cutoff < 0 || C == A || C == B)
    return (int)hipErrorInvalidValue;
  const int W = (int)g_ranks.size();
  const int64_t m = A->rows, l = A->ncols, n = B->ncols;
  if (variant == 0) {
    // the operands' own layout decides when it fits a schedule; otherwise the shape and the world size
    const int v = layout_levels(A->layout);
    if (v > 0 && B->layout == A->layout && C->layout == A->layout) variant = M4RI_AMD_VARIANT_STRASSEN;
    else if (A->layout == M4RI_AMD_LAYOUT_ROWS && C->layout == M4RI_AMD_LAYOUT_ROWS && (B->layout == M4RI_AMD_LAYOUT_ROWS || B->layout == M4RI_AMD_LAYOUT_REPLICATED))
      variant = M4RI_AMD_VARIANT_SLABS;
    else variant = g_variant ? g_variant : default_variant(W, m, l, n);
  }
  int want = M4RI_AMD_LAYOUT_ROWS;
  if (variant == M4RI_AMD_VARIANT_STRASSEN) {
    const int v = layout_levels(A->layout) ? layout_levels(A->layout) : layout_levels(C->layout) ? layout_levels(C->layout) : auto_levels(W, m, l, n);
    want        = v == 2 ? M4RI_AMD_LAYOUT_CYCLIC2 : M4RI_AMD_LAYOUT_CYCLIC1;
  }
  // conversions
  std::unique_ptr<m4ri_amd_dmat, void (*)(m4ri_amd_dmat *)> tA(nullptr, dmat_delete), tB(nullptr, dmat_delete), tC(nullptr, dmat_delete);
  const m4ri_amd_dmat *a = A, *b = B;
  m4ri_amd_dmat *c = C;
  if (A->layout != want) {
    tA.reset(dmat_new(m, l, want));
    if (!tA) return (int)hipErrorOutOfMemory;
    if (int rc = redistribute(tA.get(), A)) return rc;
    a = tA.get();
  }
  if (B->layout != want && !(want == M4RI_AMD_LAYOUT_ROWS && B->layout == M4RI_AMD_LAYOUT_REPLICATED)) {
    tB.reset(dmat_new(l, n, want));
    if (!tB) return (int)hipErrorOutOfMemory;
    if (int rc = redistribute(tB.get(), B)) return rc;
    b = tB.get();
  }
  if (C->layout != want) {
    tC.reset(dmat_new(m, n, want));
    if (!tC) return (int)hipErrorOutOfMemory;
    if (add)
      if (int rc = redistribute(tC.get(), C)) return rc;
    c = tC.get();
  }
  g_mstats = m4ri_amd_multi_stats{};
  g_mstats.world = W; g_mstats.variant = variant; g_mstats.m = m; g_mstats.l = l; g_mstats.n = n;
  g_mstats.converted = (tA ? 1 : 0) + (tB ? 1 : 0) + (tC ? 1 : 0);
  g_mstats.pairs_staged = g_pairs_staged;
  int rc = 0;
  ++g_seq;
  if (m == 0 || n == 0) {
    rc = 0;
  } else if (variant == M4RI_AMD_VARIANT_SLABS) {
    SlabOp op{c, a, b, add, cutoff, false};
    const int64_t kb = (l + W - 1) / W;
    // the gather under the product with the own slab: needs every slab boundary of B on a word of A's rows, and pays when a rank's
    // own slab is a large part of the inner dimension (few ranks)
    op.overlap = b->layout == M4RI_AMD_LAYOUT_ROWS && kb % 64 == 0 && W <= 4 && W > 1 && l > 0;
    g_mstats.overlap    = op.overlap ? 1 : 0;
    g_mstats.link_bytes = b->layout == M4RI_AMD_LAYOUT_ROWS ? 8.0 * (double)b->stride * (double)l * (double)(W - 1) : 0.0;
    rc = run_op(lane, false, [&](int me, const OpCtx &ctx) { return slabs_rank(op, me, ctx); });
  } else {
    StrassenOp op{c, a, b, {}, add, cutoff, 1, g_seq, 1};
    plan_for(&op.p, W, m, l, n, layout_levels(want));
    // two ROW chunks when one product per rank is all there is to hide transfers behind and its halves keep the engine's Strassen depth
    // (measured: two 16384 x 32768 x 32768 halves cost 1.00 - 1.03 of the whole, profiles/r03_rank_shapes_timing.log)
    op.chunks = (op.p.levels == 1 && op.p.bm >= 4 * 4096 && W >= 2) ? 2 : 1;
    if (const char *env = getenv("M4RI_AMD_MULTI_CHUNKS")) { const int v = atoi(env); if (v >= 1 && v <= W && v <= 16) op.chunks = v; }
    op.group = op.chunks == 1 ? pick_group(op.p, cutoff) : 1;
    if (const char *env = getenv("M4RI_AMD_MULTI_GROUP")) { const int v = atoi(env); if (v >= 1 && v <= 64 && op.chunks == 1) op.group = v; }
    g_mstats.levels = op.p.levels; g_mstats.sub_products = op.p.nprod; g_mstats.chunks = op.chunks; g_mstats.group = op.group;
    double moved = 0;
    for (int side = 0; side < 3; ++side)
      for (int j = 0; j < op.p.nprod; ++j)
        for (int r = 0; r < W; ++r) {
          m4ri_amd_shard_piece pc;
          m4ri_amd_shard_piece_of(&op.p, side, j, r, &pc);
          if (pc.holder != pc.owner) moved += 8.0 * (double)pc.words;
        }
    g_mstats.link_bytes = moved;
    rc = run_op(lane, false, [&](int me, const OpCtx &ctx) { return strassen_rank(op, me, ctx); });
  }
  if (rc) return rc;
  if (tC) return redistribute(C, tC.get());  // (the temporaries are freed after a device synchronisation: dmat_delete)
  return 0;
}

int sync_all() {
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  int rc = 0;
  for (Lane &L : g_lane)
  for (auto &r : L.ranks) {
    if (hipSetDevice(r->device) != hipSuccess) { rc = (int)hipErrorInvalidDevice; continue; }
    std::vector<hipStream_t> all = r->lin;
    all.insert(all.end(), r->lout.begin(), r->lout.end());
    all.insert(all.end(), {r->ci, r->co, r->st});
    for (hipStream_t s : all) {
      const hipError_t e = hipStreamSynchronize(s);
      if (e != hipSuccess && !rc) rc = (int)e;
    }
  }
  (void)hipSetDevice(cur);
  return rc;
}

// ---- host matrices in and out ----------------------------------------------------------------------------
// host rows of M -> the zero-padded local buffers: every device its own rows over its own PCIe link
int upload_rank(m4ri_amd_dmat *d, const mzd_t *M, int me, const OpCtx &ctx) {
  Rank &R = *g_lane[ctx.lane].ranks[(size_t)me];
  RTRY(hipSetDevice(R.device));
  RTRY(op_begin(R, ctx));
  word *local = d->local[(size_t)me];
  if (d->lrows[(size_t)me] > 0) RTRY(hipMemsetAsync(local, 0, (size_t)d->lrows[(size_t)me] * (size_t)d->stride * 8, R.st));
  Run runs[4];
  const int nr = runs_of(d->layout, d->world, me, d->rows, runs);
  for (int i = 0; i < nr && M->width > 0; ++i) {
    RTRY(hipMemcpy2DAsync(local + runs[i].l0 * d->stride, (size_t)d->stride * 8, M->data + runs[i].g0 * M->rowstride, (size_t)M->rowstride * 8,
                          (size_t)M->width * 8, (size_t)runs[i].rows, hipMemcpyHostToDevice, R.st));
    // a window's last word carries its parent's neighbouring columns (mzd.h:117-123): zero them
    if (M->ncols % 64) RTRY(gf2_launch_mask_tail(R.st, local + runs[i].l0 * d->stride, d->stride, runs[i].rows, M->ncols));
  }
  RTRY(op_end(R, ctx));
  RTRY(hipStreamSynchronize(R.st));  // the caller may free or change M as soon as this returns
  return 0;
}

// the local buffers -> host C, touching only the words and bits the reference would (mzd.h:117-123)
int download_rank(const m4ri_amd_dmat *d, mzd_t *C, int me, bool all_ranks, const OpCtx &ctx) {
  Rank &R = *g_lane[ctx.lane].ranks[(size_t)me];
  RTRY(hipSetDevice(R.device));
  RTRY(op_begin(R, ctx));
  RTRY(op_end(R, ctx));
  if (C->width == 0) return 0;
  if (d->layout == M4RI_AMD_LAYOUT_REPLICATED && !all_ranks && me != 0) return 0;
  const bool dangerous = (C->flags & FLAG_WINDOW) && (C->ncols % 64 != 0);
  const int64_t wfull  = dangerous ? C->width - 1 : C->width;
  std::vector<word> last;
  Run runs[4];
  int nr = runs_of(d->layout, d->world, me, d->rows, runs);
  if (d->layout == M4RI_AMD_LAYOUT_REPLICATED && all_ranks) {  // every rank downloads a share of the rows
    const int64_t k = (d->rows + d->world - 1) / d->world, g0 = k * me;
    nr = g0 < d->rows ? 1 : 0;
    runs[0] = Run{g0, (d->rows - g0) < k ? (d->rows - g0) : k, g0};
  }
  for (int i = 0; i < nr; ++i) {
    const word *src = d->local[(size_t)me] + runs[i].l0 * d->stride;
    if (wfull > 0)
      RTRY(hipMemcpy2DAsync(C->data + runs[i].g0 * C->rowstride, (size_t)C->rowstride * 8, src, (size_t)d->stride * 8, (size_t)wfull * 8, (size_t)runs[i].rows,
                            hipMemcpyDeviceToHost, R.st));
    if (dangerous) {
      last.resize((size_t)runs[i].rows);
      RTRY(hipMemcpy2DAsync(last.data(), 8, src + (C->width - 1), (size_t)d->stride * 8, 8, (size_t)runs[i].rows, hipMemcpyDeviceToHost, R.st));
      RTRY(hipStreamSynchronize(R.st));
      const word mask = C->high_bitmask;
      for (int64_t k = 0; k < runs[i].rows; ++k) {
        word *w = C->data + (runs[i].g0 + k) * C->rowstride + (C->width - 1);
        *w      = (*w & ~mask) | (last[(size_t)k] & mask);
      }
    }
  }
  RTRY(hipStreamSynchronize(R.st));
  return 0;
}

// the three distributed temporaries of the host entry point, kept between calls (grow-only in spirit: re-made when the
// shape or the layout changes)
m4ri_amd_dmat *g_tmp[3] = {nullptr, nullptr, nullptr};

m4ri_amd_dmat *tmp_dmat(int slot, int64_t rows, int64_t ncols, int layout) {
  m4ri_amd_dmat *&d = g_tmp[slot];
  if (d && alive(d) && d->rows == rows && d->ncols == ncols && d->layout == layout) return d;
  if (d) dmat_delete(d);
  d = dmat_new(rows, ncols, layout);
  return d;
}

void drop_tmp() {
  for (m4ri_amd_dmat *&d : g_tmp) {
    if (d) dmat_delete(d);
    d = nullptr;
  }
}

int mul_multi(mzd_t *C, const mzd_t *A, const mzd_t *B, int add, int cutoff, int levels) {
  if (int rc = ensure_ranks()) return rc;
  const int W = (int)g_ranks.size();
  const int64_t m = A->nrows, l = A->ncols, n = B->ncols;
  int variant = levels > 0 ? M4RI_AMD_VARIANT_STRASSEN : (g_variant ? g_variant : default_variant(W, m, l, n));
  if (levels == 0 && variant == M4RI_AMD_VARIANT_STRASSEN) levels = auto_levels(W, m, l, n);
  const int layout = variant == M4RI_AMD_VARIANT_STRASSEN ? (levels == 2 ? M4RI_AMD_LAYOUT_CYCLIC2 : M4RI_AMD_LAYOUT_CYCLIC1) : M4RI_AMD_LAYOUT_ROWS;
  m4ri_amd_dmat *dA = tmp_dmat(0, m, l, layout), *dB = tmp_dmat(1, l, n, layout), *dC = tmp_dmat(2, m, n, layout);
  if (!dA || !dB || !dC) return (int)hipErrorOutOfMemory;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  // (three operations, not one with three parts: an operation records ev_done once, and the next part's op_begin has to see it)
  int rc = run_op(0, true, [&](int me, const OpCtx &ctx) { return upload_rank(dA, A, me, ctx); });
  if (!rc) rc = run_op(0, true, [&](int me, const OpCtx &ctx) { return upload_rank(dB, B, me, ctx); });
  if (!rc && add) rc = run_op(0, true, [&](int me, const OpCtx &ctx) { return upload_rank(dC, C, me, ctx); });
  if (!rc) rc = dmat_mul(dC, dA, dB, add, cutoff, variant);
  if (!rc) rc = run_op(0, true, [&](int me, const OpCtx &ctx) { return download_rank(dC, C, me, true, ctx); });
  (void)hipSetDevice(cur);
  return rc;
}

}  // namespace

extern "C" {

int m4ri_amd_set_devices(int n, const int *ids) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (n < 0 || (n > 0 && !ids) || n > MAX_RANKS) return -1;  // up to 64 RANKS (ids may repeat) ...
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess) have = 0;
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= have || ids[i] >= 16) return -1;  // ... on device ids the per-device tables (arena, scratch: 16 entries) hold
  g_devices.assign(ids, ids + n);
  g_devices_set = n > 0;  // n == 0: back to the default (M4RI_AMD_DEVICES or every visible device)
  return 0;
}

int m4ri_amd_get_device_list(int *ids, int cap) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  default_devices();
  for (int i = 0; i < (int)g_devices.size() && i < cap; ++i) ids[i] = g_devices[(size_t)i];
  return (int)g_devices.size();
}

int64_t m4ri_amd_set_multi_threshold(int64_t min_dim) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  const int64_t old = g_threshold;
  if (min_dim >= 0) g_threshold = min_dim;
  return old;
}

int m4ri_amd_set_multi_variant(int variant) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  const int old = g_variant;
  if (variant >= 0 && variant <= M4RI_AMD_VARIANT_STRASSEN) g_variant = variant;
  return old;
}

int m4ri_amd_multi_default_variant(int world, int64_t m, int64_t l, int64_t n) { return default_variant(world, m, l, n); }

int m4ri_amd_shard_group(const m4ri_amd_shard_plan *p, int cutoff) { return p && p->world >= 1 && p->nprod >= 1 ? pick_group(*p, cutoff) : 1; }

int m4ri_amd_multi_layout_for(int variant, int world, int64_t m, int64_t l, int64_t n) {
  if (variant == 0) variant = default_variant(world, m, l, n);
  if (variant != M4RI_AMD_VARIANT_STRASSEN) return M4RI_AMD_LAYOUT_ROWS;
  return auto_levels(world, m, l, n) == 2 ? M4RI_AMD_LAYOUT_CYCLIC2 : M4RI_AMD_LAYOUT_CYCLIC1;
}

int64_t m4ri_amd_layout_local_rows(int layout, int world, int rank, int64_t rows) {
  if (layout < 0 || layout > M4RI_AMD_LAYOUT_REPLICATED || world < 1 || rank < 0 || rank >= world || rows < 0) return -1;
  return local_rows_of(layout, world, rank, rows);
}

int m4ri_amd_layout_runs(int layout, int world, int rank, int64_t rows, int64_t *g0, int64_t *nrows, int64_t *l0, int cap) {
  if (layout < 0 || layout > M4RI_AMD_LAYOUT_REPLICATED || world < 1 || rank < 0 || rank >= world || rows < 0) return -1;
  Run runs[4];
  const int n = runs_of(layout, world, rank, rows, runs);
  for (int i = 0; i < n && i < cap; ++i) { g0[i] = runs[i].g0; nrows[i] = runs[i].rows; l0[i] = runs[i].l0; }
  return n;
}

// would mzd_mul_mp spread this product over several devices?  (gf2_multi_run's own test)
int gf2_multi_wanted(int64_t m, int64_t l, int64_t n) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  default_devices();
  const int64_t mn = m < l ? (m < n ? m : n) : (l < n ? l : n);
  return g_devices.size() > 1 && mn >= g_threshold && mn >= 1;
}

m4ri_amd_dmat *m4ri_amd_dmat_create(int64_t rows, int64_t ncols, int layout) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (ensure_ranks()) return nullptr;
  return dmat_new(rows, ncols, layout);
}

void m4ri_amd_dmat_free(m4ri_amd_dmat *d) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  dmat_delete(d);
}

int m4ri_amd_dmat_info(const m4ri_amd_dmat *d, m4ri_amd_dmat_info_t *out) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (!d || !out) return -1;
  out->rows = d->rows; out->ncols = d->ncols; out->stride = d->stride; out->layout = d->layout; out->world = d->world;
  out->alive = alive(d) ? 1 : 0;
  return 0;
}

int m4ri_amd_dmat_local(const m4ri_amd_dmat *d, int rank, word **data, int64_t *local_rows, int *device) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (!d || rank < 0 || rank >= d->world) return -1;
  if (data) *data = d->local[(size_t)rank];
  if (local_rows) *local_rows = d->lrows[(size_t)rank];
  if (device) *device = d->device[(size_t)rank];
  return 0;
}

int m4ri_amd_dmat_fill(m4ri_amd_dmat *d, uint64_t seed) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (!alive(d)) return (int)hipErrorInvalidValue;
  return run_op(0, true, [&](int me, const OpCtx &ctx) -> int {
    Rank &R = *g_ranks[(size_t)me];
    RTRY(hipSetDevice(R.device));
    RTRY(op_begin(R, ctx));
    Run runs[4];
    const int nr = runs_of(d->layout, d->world, me, d->rows, runs);
    for (int i = 0; i < nr; ++i)
      RTRY(m4ri_amd_fill_rows_dev(d->local[(size_t)me] + runs[i].l0 * d->stride, d->stride, runs[i].g0, runs[i].rows, d->ncols, seed, R.st));
    return op_end(R, ctx);
  });
}

int m4ri_amd_dmat_upload(m4ri_amd_dmat *d, const mzd_t *M) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (!alive(d) || !M || M->nrows != d->rows || M->ncols != d->ncols) return (int)hipErrorInvalidValue;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  const int rc = run_op(0, true, [&](int me, const OpCtx &ctx) { return upload_rank(d, M, me, ctx); });
  (void)hipSetDevice(cur);
  return rc;
}

int m4ri_amd_dmat_download(const m4ri_amd_dmat *d, mzd_t *M) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (!alive(d) || !M || M->nrows != d->rows || M->ncols != d->ncols) return (int)hipErrorInvalidValue;
  return run_op(0, true, [&](int me, const OpCtx &ctx) { return download_rank(d, M, me, true, ctx); });
}

int m4ri_amd_dmat_convert(m4ri_amd_dmat *dst, const m4ri_amd_dmat *src) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  return redistribute(dst, src);
}

int m4ri_amd_dmat_mul(m4ri_amd_dmat *C, const m4ri_amd_dmat *A, const m4ri_amd_dmat *B, int add, int cutoff, int variant) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (variant < 0 || variant > M4RI_AMD_VARIANT_STRASSEN) return (int)hipErrorInvalidValue;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  const int rc = dmat_mul(C, A, B, add, cutoff, variant);
  (void)hipSetDevice(cur);
  return rc;
}

int m4ri_amd_dmat_mul_lane(m4ri_amd_dmat *C, const m4ri_amd_dmat *A, const m4ri_amd_dmat *B, int add, int cutoff, int variant, int lane) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (variant < 0 || variant > M4RI_AMD_VARIANT_STRASSEN || lane < 0 || lane >= NUM_LANES) return (int)hipErrorInvalidValue;
  if (!alive(C)) return (int)hipErrorInvalidValue;
  if (int rc = ensure_lane(lane)) return rc;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  const int rc = dmat_mul(C, A, B, add, cutoff, variant, lane);
  (void)hipSetDevice(cur);
  return rc;
}

int m4ri_amd_multi_sync(void) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  return sync_all();
}

void m4ri_amd_multi_pair_table(int world, const int *devices, int ndev, const int *can_access, const char *no_peer_spec, char *staged) {
  if (world <= 0 || !devices || !staged || (ndev > 0 && !can_access)) return;
  pair_table(world, devices, ndev, can_access, no_peer_spec, staged);
}

// What the links between the configured ranks give, measured: `bytes` copied dst <- src for every ordered pair of ranks, one pair at a
// time (pair_gbs[dst * W + src], GB/s, 0 on the diagonal) and then all pairs at once (*all_gbs = the sum of all bytes over the wall
// time); staged[dst * W + src] = 1 where the pair has no peer access and copies go through the host.  Ranks sharing a device report
// the on-device copy rate (*same_device = 1 if any pair does).  Uses the ranks' own link streams: exactly the path the schedules use.
int m4ri_amd_multi_link_probe(int64_t bytes, double *pair_gbs, double *all_gbs, char *staged, int *same_device) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (bytes < 8 || !pair_gbs) return (int)hipErrorInvalidValue;
  if (int rc = ensure_ranks()) return rc;
  if (int rc = sync_all()) return rc;
  const int W = (int)g_ranks.size();
  const int64_t words = bytes / 8;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  std::vector<word *> src((size_t)W, nullptr), dst((size_t)W * (size_t)W, nullptr);
  int rc = 0;
  auto cleanup = [&]() {
    for (int i = 0; i < W; ++i) {
      (void)hipSetDevice(g_ranks[(size_t)i]->device);
      if (src[(size_t)i]) (void)hipFree(src[(size_t)i]);
      for (int j = 0; j < W; ++j)
        if (dst[(size_t)i * W + j]) (void)hipFree(dst[(size_t)i * W + j]);
    }
    (void)hipSetDevice(cur);
  };
  bool shared = false;
  for (int i = 0; i < W && !rc; ++i) {
    Rank &R = *g_ranks[(size_t)i];
    rc = (int)hipSetDevice(R.device);
    if (!rc) rc = (int)hipMalloc(reinterpret_cast<void **>(&src[(size_t)i]), (size_t)words * 8);
    if (!rc) rc = (int)hipMemset(src[(size_t)i], 0x3c, (size_t)words * 8);
    for (int j = 0; j < W && !rc; ++j) {
      if (j == i) continue;
      shared = shared || g_ranks[(size_t)j]->device == R.device;
      rc = (int)hipMalloc(reinterpret_cast<void **>(&dst[(size_t)i * W + j]), (size_t)words * 8);
    }
    if (!rc) rc = (int)hipDeviceSynchronize();
  }
  if (rc) { cleanup(); return rc; }
  for (int i = 0; i < W * W; ++i) pair_gbs[i] = 0.0;
  for (auto &r : g_ranks) r->tl_valid = false;  // (the probe borrows the timeline's first and last mark)
  auto one = [&](int i, int j) -> int {  // rank i pulls from rank j on its inbound link stream
    Rank &R = *g_ranks[(size_t)i];
    HIPTRY(hipSetDevice(R.device));
    return copy_words(R, i, j, dst[(size_t)i * W + j], src[(size_t)j], j, g_ranks[(size_t)j]->device, words, R.lin[(size_t)j]);
  };
  for (int i = 0; i < W && !rc; ++i)
    for (int j = 0; j < W && !rc; ++j) {
      if (i == j) continue;
      Rank &R = *g_ranks[(size_t)i];
      rc = one(i, j);  // warm-up: first touch of the pair (mappings, the bounce buffer)
      if (!rc) rc = (int)hipStreamSynchronize(R.lin[(size_t)j]);
      if (!rc) rc = (int)hipEventRecord(R.ev_tl0, R.lin[(size_t)j]);
      if (!rc) rc = one(i, j);
      if (!rc) rc = (int)hipEventRecord(R.ev_tl1, R.lin[(size_t)j]);
      if (!rc) rc = (int)hipEventSynchronize(R.ev_tl1);
      float ms = 0;
      if (!rc) rc = (int)hipEventElapsedTime(&ms, R.ev_tl0, R.ev_tl1);
      if (!rc && ms > 0) pair_gbs[(size_t)i * W + j] = (double)bytes / ((double)ms * 1e-3) / 1e9;
    }
  if (!rc && all_gbs) {
    *all_gbs = 0.0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    for (int i = 0; i < W && !rc; ++i)
      for (int k = 1; k < W && !rc; ++k) rc = one(i, (i + k) % W);
    if (!rc) rc = sync_all();
    const double dt = now() - t0;
    if (!rc && dt > 0) *all_gbs = (double)bytes * (double)W * (double)(W - 1) / dt / 1e9;
  }
  if (staged)
    for (int i = 0; i < W * W; ++i) staged[i] = g_staged.empty() ? 0 : g_staged[(size_t)i];
  if (same_device) *same_device = shared ? 1 : 0;
  cleanup();
  return rc;
}

int m4ri_amd_multi_get_stats(m4ri_amd_multi_stats *out) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (!out) return -1;
  *out = g_mstats;
  return 0;
}

// The marks of rank `rank`'s part of the most recent m4ri_amd_dmat_mul, in ms after the rank's compute stream entered the
// operation (the devices are synchronised first).  Order: Strassen schedule -- down pass done; then per unit (round, row
// chunk) operands in, product done; result slabs in; up pass done.  Row slabs -- gather done; first product done (the one
// with the own slab of B when the gather is overlapped); all done.  Returns the number of marks written, < 0 on error.
int m4ri_amd_multi_timeline(int rank, double *ms, int cap) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (rank < 0 || rank >= (int)g_ranks.size() || !ms) return -1;
  if (sync_all()) return -1;
  Rank &R = *g_ranks[(size_t)rank];
  if (!R.tl_valid) return 0;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(R.device);
  std::vector<hipEvent_t> marks;
  if (R.tl_strassen) {
    marks.push_back(R.ev_down);
    for (int u = 0; u < R.tl_units; ++u) { marks.push_back(R.ev_in[(size_t)u]); marks.push_back(R.ev_prod[(size_t)u]); }
    marks.push_back(R.ev_back);
  } else {
    marks.push_back(R.ev_gather);
    marks.push_back(R.ev_first);
  }
  marks.push_back(R.ev_tl1);
  int n = 0;
  for (hipEvent_t e : marks) {
    float t = 0;
    if (n >= cap) break;
    if (hipEventElapsedTime(&t, R.ev_tl0, e) != hipSuccess) t = -1.0f;
    ms[n++] = (double)t;
  }
  (void)hipSetDevice(cur);
  return n;
}

int m4ri_amd_mul_multi(mzd_t *C, const mzd_t *A, const mzd_t *B, int add, int cutoff, int levels) {
  if (!C || !A || !B || A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols || cutoff < 0 || levels < 0 || levels > 2) return (int)hipErrorInvalidValue;
  if (C->nrows == 0 || C->ncols == 0) return 0;
  if (A->ncols == 0) {  // empty inner dimension: C = 0 / C unchanged
    if (!add)
      for (rci_t i = 0; i < C->nrows; ++i) {
        word *row = C->data + (int64_t)i * C->rowstride;
        for (wi_t k = 0; k + 1 < C->width; ++k) row[k] = 0;
        row[C->width - 1] &= ~C->high_bitmask;
      }
    return 0;
  }
  std::lock_guard<std::mutex> lk(g_multi_mu);
  return mul_multi(C, A, B, add, cutoff, levels);
}

void gf2_release_multi(void) {  // called by m4ri_amd_release_workspace
  std::lock_guard<std::mutex> lk(g_multi_mu);
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return;
  drop_tmp();
  for (auto &rp : g_ranks) {
    Rank &r = *rp;
    (void)hipSetDevice(r.device);
    (void)hipDeviceSynchronize();
    if (r.arena) (void)hipFree(r.arena);
    r.arena = nullptr; r.cap = 0;
  }
  (void)hipSetDevice(cur);
}

}  // extern "C"
