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
//   * m4ri_amd_mul_multi / mzd_mul_mp: one process, all devices, host mzd_t in and out; every device
//     uploads and downloads its own slabs (W PCIe links in parallel) and pieces move by
//     hipMemcpyPeerAsync.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    // fewest "rounds" of sub-products on the busiest rank, as a fraction of the whole: ceil(7^v / W) / 7^v;
    // the second level only when it wins and its sub-products stay large (>= 2048 on every side)
    const double f1 = (double)((7 + world - 1) / world) / 7.0, f2 = (double)((49 + world - 1) / world) / 49.0;
    levels = (f2 < f1 - 1e-9 && m / 4 >= 2048 && l / 4 >= 2048 && n / 4 >= 2048) ? 2 : 1;
  }
  memset(p, 0, sizeof *p);
  p->world  = world;
  p->levels = levels;
  p->blocks = 1 << levels;
  p->nprod  = levels == 2 ? 49 : 7;
  p->m = m; p->l = l; p->n = n;
  p->M = roundup(m, p->blocks);
  p->L = roundup(l, 64ll * p->blocks);
  p->N = roundup(n, 64ll * p->blocks);
  p->bm  = p->M / p->blocks;
  p->bl  = p->L / p->blocks;
  p->cwl = p->L / p->blocks / 64;
  p->cwn = p->N / p->blocks / 64;
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
  else HIPTRY(gf2_launch_winograd_up2(st, add ? 1 : 0, slabs_p, C_local, c_stride, 0, 1, sa, p->cwn));
  return 0;
}

}  // extern "C"

// ================================ one process, all devices, host mzd_t ==============================
namespace {

constexpr uint8_t FLAG_WINDOW = 0x4;  // mzd.h:150

struct Rank {
  int device         = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_down = nullptr, ev_prod = nullptr;
  word *arena        = nullptr;
  size_t cap         = 0;  // words
  word *buf[9]       = {};
  int rc             = 0;
};

std::mutex g_multi_mu;
std::vector<int> g_devices;   // the devices products are spread over (an id may repeat: "virtual" ranks)
bool g_devices_set = false;
std::vector<Rank> g_ranks;
int64_t g_threshold = 16384;  // smallest min(m, l, n) mzd_mul_mp spreads over several devices

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
  }
  if (g_devices.empty()) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) n = 1;
    for (int d = 0; d < n; ++d) g_devices.push_back(d);
  }
}

int ensure_ranks() {
  default_devices();
  if (g_ranks.size() == g_devices.size()) {
    bool same = true;
    for (size_t i = 0; i < g_ranks.size(); ++i) same = same && g_ranks[i].device == g_devices[i];
    if (same) return 0;
  }
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  auto drop_all = [&]() {  // streams, events and arenas of every rank that has them; leaves g_ranks empty
    for (Rank &r : g_ranks) {
      (void)hipSetDevice(r.device);
      if (r.arena) (void)hipFree(r.arena);
      if (r.stream) (void)hipStreamDestroy(r.stream);
      if (r.ev_down) (void)hipEventDestroy(r.ev_down);
      if (r.ev_prod) (void)hipEventDestroy(r.ev_prod);
    }
    g_ranks.clear();
  };
  drop_all();
  // a failure half way must not leave ranks that LOOK configured (same size, same ids, null streams) nor another current
  // device behind: on any error everything made so far is dropped and the caller's device restored
  auto build = [&]() -> int {
    g_ranks.assign(g_devices.size(), Rank{});
    for (size_t i = 0; i < g_ranks.size(); ++i) {
      Rank &r  = g_ranks[i];
      r.device = g_devices[i];
      HIPTRY(m4ri_amd_init(r.device));  // binds the device + creates its engine
      HIPTRY(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
      HIPTRY(hipEventCreateWithFlags(&r.ev_down, hipEventDisableTiming));
      HIPTRY(hipEventCreateWithFlags(&r.ev_prod, hipEventDisableTiming));
      for (size_t k = 0; k < g_ranks.size(); ++k) {  // direct xGMI copies between every pair
        const int other = g_devices[k];
        int can         = 0;
        if (other != r.device && hipDeviceCanAccessPeer(&can, r.device, other) == hipSuccess && can) {
          const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
          if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return (int)e;
          (void)hipGetLastError();
        }
      }
    }
    return 0;
  };
  const int rc = build();
  if (rc) drop_all();
  (void)hipSetDevice(cur);
  return rc;
}

int carve(Rank &r, const m4ri_amd_shard_plan &p, int rank) {
  size_t need = 0;
  int64_t w[9];
  for (int k = 0; k < 9; ++k) { w[k] = pad32(m4ri_amd_shard_buffer_words(&p, rank, k)); need += (size_t)w[k]; }
  if (need > r.cap) {
    HIPTRY(hipStreamSynchronize(r.stream));
    if (r.arena) HIPTRY(hipFree(r.arena));
    r.arena = nullptr; r.cap = 0;
    HIPTRY(hipMalloc(reinterpret_cast<void **>(&r.arena), need * 8));
    r.cap = need;
  }
  word *q = r.arena;
  for (int k = 0; k < 9; ++k) { r.buf[k] = q; q += w[k]; }
  return 0;
}

// host rows of M (slab-cyclic selection for `rank`) -> the zero-padded local parent on the device
int upload_local(Rank &r, const m4ri_amd_shard_plan &p, int rank, const mzd_t *M, word *local, int64_t brows, int64_t lwidth) {
  const int64_t c0 = cut_of(brows, p.world, rank), s = cut_of(brows, p.world, rank + 1) - c0;
  if (s == 0) return 0;
  HIPTRY(hipMemsetAsync(local, 0, (size_t)(p.blocks * s * lwidth) * 8, r.stream));
  for (int b = 0; b < p.blocks; ++b) {
    const int64_t g0 = (int64_t)b * brows + c0;
    int64_t rows     = (int64_t)M->nrows - g0;
    if (rows > s) rows = s;
    if (rows <= 0 || M->width == 0) continue;
    HIPTRY(hipMemcpy2DAsync(local + (int64_t)b * s * lwidth, (size_t)lwidth * 8, M->data + g0 * M->rowstride, (size_t)M->rowstride * 8,
                            (size_t)M->width * 8, (size_t)rows, hipMemcpyHostToDevice, r.stream));
  }
  // a window's last word carries its parent's neighbouring columns (mzd.h:117-123): zero them
  HIPTRY(gf2_launch_mask_tail(r.stream, local, lwidth, p.blocks * s, M->ncols));
  return 0;
}

// the local parent of C -> host C, touching only the words and bits the reference would (mzd.h:117-123)
int download_local(Rank &r, const m4ri_amd_shard_plan &p, int rank, mzd_t *C, const word *local) {
  const int64_t c0 = cut_of(p.bm, p.world, rank), s = cut_of(p.bm, p.world, rank + 1) - c0, lw = p.N / 64;
  if (s == 0 || C->width == 0) return 0;
  const bool dangerous = (C->flags & FLAG_WINDOW) && (C->ncols % 64 != 0);
  const int64_t wfull  = dangerous ? C->width - 1 : C->width;
  std::vector<word> last;
  for (int b = 0; b < p.blocks; ++b) {
    const int64_t g0 = (int64_t)b * p.bm + c0;
    int64_t rows     = (int64_t)C->nrows - g0;
    if (rows > s) rows = s;
    if (rows <= 0) continue;
    const word *src = local + (int64_t)b * s * lw;
    if (wfull > 0)
      HIPTRY(hipMemcpy2DAsync(C->data + g0 * C->rowstride, (size_t)C->rowstride * 8, src, (size_t)lw * 8, (size_t)wfull * 8, (size_t)rows,
                              hipMemcpyDeviceToHost, r.stream));
    if (dangerous) {
      last.resize((size_t)rows);
      HIPTRY(hipMemcpy2DAsync(last.data(), 8, src + (C->width - 1), (size_t)lw * 8, 8, (size_t)rows, hipMemcpyDeviceToHost, r.stream));
      HIPTRY(hipStreamSynchronize(r.stream));
      const word mask = C->high_bitmask;
      for (int64_t i = 0; i < rows; ++i) {
        word *w = C->data + (g0 + i) * C->rowstride + (C->width - 1);
        *w      = (*w & ~mask) | (last[(size_t)i] & mask);
      }
    }
  }
  return 0;
}

template <typename F>
int on_every_rank(F f) {
  std::vector<std::thread> th;
  for (size_t i = 0; i < g_ranks.size(); ++i)
    th.emplace_back([&, i] {
      Rank &r = g_ranks[i];
      r.rc    = (int)hipSetDevice(r.device);
      if (r.rc == 0) r.rc = f(r, (int)i);
    });
  for (auto &t : th) t.join();
  for (Rank &r : g_ranks)
    if (r.rc) return r.rc;
  return 0;
}

int mul_multi(mzd_t *C, const mzd_t *A, const mzd_t *B, int add, int cutoff, int levels) {
  if (int rc = ensure_ranks()) return rc;
  const int W = (int)g_ranks.size();
  m4ri_amd_shard_plan p;
  if (m4ri_amd_shard_plan_make(&p, W, A->nrows, A->ncols, B->ncols, levels)) return (int)hipErrorInvalidValue;
  int cur = 0;
  HIPTRY(hipGetDevice(&cur));
  // 1. every device: its slabs of A, B (and C when accumulating) over its own PCIe link, local down passes
  int rc = on_every_rank([&](Rank &r, int i) -> int {
    if (int e = carve(r, p, i)) return e;
    if (int e = upload_local(r, p, i, A, r.buf[M4RI_AMD_SHARD_BUF_LOCAL_A], p.bm, p.L / 64)) return e;
    if (int e = upload_local(r, p, i, B, r.buf[M4RI_AMD_SHARD_BUF_LOCAL_B], p.bl, p.N / 64)) return e;
    if (add) { if (int e = upload_local(r, p, i, C, r.buf[M4RI_AMD_SHARD_BUF_LOCAL_C], p.bm, p.N / 64)) return e; }
    if (int e = m4ri_amd_shard_down_dev(&p, i, r.buf[M4RI_AMD_SHARD_BUF_LOCAL_A], p.L / 64, r.buf[M4RI_AMD_SHARD_BUF_LOCAL_B], p.N / 64,
                                        r.buf[M4RI_AMD_SHARD_BUF_CHILD_A], r.buf[M4RI_AMD_SHARD_BUF_CHILD_B], r.stream)) return e;
    return (int)hipEventRecord(r.ev_down, r.stream);
  });
  if (rc) { (void)hipSetDevice(cur); return rc; }
  // 2. slab r of child j: r -> owner(j), pulled on the owner's stream (one peer copy per piece: on a real
  //    node W*(W-1) directed xGMI links work at once); then the owner's sub-products
  for (int o = 0; o < W; ++o) {
    Rank &ro = g_ranks[o];
    HIPTRY(hipSetDevice(ro.device));
    bool any = false;
    for (int j = o; j < p.nprod; j += W) {
      any = true;
      for (int side = 0; side < 2; ++side)
        for (int r = 0; r < W; ++r) {
          m4ri_amd_shard_piece pc;
          m4ri_amd_shard_piece_of(&p, side, j, r, &pc);
          if (pc.words == 0) continue;
          Rank &rh = g_ranks[r];
          if (j == o) HIPTRY(hipStreamWaitEvent(ro.stream, rh.ev_down, 0));
          HIPTRY(hipMemcpyPeerAsync(ro.buf[side ? M4RI_AMD_SHARD_BUF_OPER_B : M4RI_AMD_SHARD_BUF_OPER_A] + pc.owner_off, ro.device,
                                    rh.buf[side ? M4RI_AMD_SHARD_BUF_CHILD_B : M4RI_AMD_SHARD_BUF_CHILD_A] + pc.holder_off, rh.device,
                                    (size_t)pc.words * 8, ro.stream));
        }
    }
    if (any) {
      for (int j = o, jl = 0; j < p.nprod; j += W, ++jl)
        HIPTRY(m4ri_amd_mul_dev(ro.buf[M4RI_AMD_SHARD_BUF_PROD] + (int64_t)jl * p.bm * p.cwn, p.cwn,
                                ro.buf[M4RI_AMD_SHARD_BUF_OPER_A] + (int64_t)jl * p.bm * p.cwl, p.cwl,
                                ro.buf[M4RI_AMD_SHARD_BUF_OPER_B] + (int64_t)jl * p.bl * p.cwn, p.cwn, p.bm, p.bl, p.cwn * 64, 0, cutoff,
                                ro.stream));
    }
    HIPTRY(hipEventRecord(ro.ev_prod, ro.stream));
  }
  // 3. slab r of product j: owner(j) -> r, pulled on r's stream; local up pass; 4. download
  for (int r = 0; r < W; ++r) {
    Rank &rh = g_ranks[r];
    HIPTRY(hipSetDevice(rh.device));
    for (int o = 0; o < W && o < p.nprod; ++o) HIPTRY(hipStreamWaitEvent(rh.stream, g_ranks[o].ev_prod, 0));
    for (int j = 0; j < p.nprod; ++j) {
      m4ri_amd_shard_piece pc;
      m4ri_amd_shard_piece_of(&p, 2, j, r, &pc);
      if (pc.words == 0) continue;
      Rank &ro = g_ranks[pc.owner];
      HIPTRY(hipMemcpyPeerAsync(rh.buf[M4RI_AMD_SHARD_BUF_SLABS_P] + pc.holder_off, rh.device, ro.buf[M4RI_AMD_SHARD_BUF_PROD] + pc.owner_off,
                                ro.device, (size_t)pc.words * 8, rh.stream));
    }
    HIPTRY(m4ri_amd_shard_up_dev(&p, r, rh.buf[M4RI_AMD_SHARD_BUF_SLABS_P], rh.buf[M4RI_AMD_SHARD_BUF_LOCAL_C], p.N / 64, add, rh.stream));
  }
  rc = on_every_rank([&](Rank &r, int i) -> int {
    if (int e = download_local(r, p, i, C, r.buf[M4RI_AMD_SHARD_BUF_LOCAL_C])) return e;
    return (int)hipStreamSynchronize(r.stream);
  });
  (void)hipSetDevice(cur);
  return rc;
}

}  // namespace

extern "C" {

int m4ri_amd_set_devices(int n, const int *ids) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  if (n < 0 || (n > 0 && !ids) || n > 64) return -1;  // up to 64 RANKS (ids may repeat) ...
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
  for (int i = 0; i < (int)g_devices.size() && i < cap; ++i) ids[i] = g_devices[i];
  return (int)g_devices.size();
}

int64_t m4ri_amd_set_multi_threshold(int64_t min_dim) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  const int64_t old = g_threshold;
  if (min_dim >= 0) g_threshold = min_dim;
  return old;
}

// would mzd_mul_mp spread this product over several devices?  (gf2_multi_run's own test)
int gf2_multi_wanted(int64_t m, int64_t l, int64_t n) {
  std::lock_guard<std::mutex> lk(g_multi_mu);
  default_devices();
  const int64_t mn = m < l ? (m < n ? m : n) : (l < n ? l : n);
  return g_devices.size() > 1 && mn >= g_threshold && mn >= 1;
}

int m4ri_amd_mul_multi(mzd_t *C, const mzd_t *A, const mzd_t *B, int add, int cutoff, int levels) {
  if (!C || !A || !B || A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols || cutoff < 0) return (int)hipErrorInvalidValue;
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
  for (Rank &r : g_ranks) {
    (void)hipSetDevice(r.device);
    if (r.stream) (void)hipStreamSynchronize(r.stream);
    if (r.arena) (void)hipFree(r.arena);
    r.arena = nullptr; r.cap = 0;
  }
  (void)hipSetDevice(cur);
}

}  // extern "C"
