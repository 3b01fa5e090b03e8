// mzd_api.hip -- the drop-in boundary (include/m4ri_amd.h, part 1): M4RI's own entry points for the
// multiply path, taking and returning host mzd_t matrices, implemented on the device engine.
//
// Reference interfaces replaced (same names, argument meaning and fatal-error behaviour):
//   mzd_mul, mzd_addmul, _mzd_mul_even, _mzd_addmul_even, _mzd_addmul   m4ri/strassen.h:52-126
//   _mzd_sqr_even, _mzd_addsqr_even                                      m4ri/strassen.c:210,528
//   mzd_mul_m4rm, mzd_addmul_m4rm, _mzd_mul_m4rm                         m4ri/brilliantrussian.h:274-317
//   mzd_mul_mp, mzd_addmul_mp                                            m4ri/mp.h:47,62
//
// Each call: upload A and B (hipMemcpy2D straight out of the caller's rows, so windows cost
// nothing extra) into a grow-only staging arena, zero the excess bits on the device, run the engine,
// and copy C back touching only the words and bits the reference would touch (mzd.h:117-123).
// Matrices pinned with m4ri_amd_pin (part 3 of the header) are not moved at all: operands are read
// from, and results left in, their device copy -- windows into a pinned parent included.  There is NO CPU fallback: a HIP
// failure is fatal, like every other error on this path (misc.c:36-42).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sys/mman.h>
#include <sys/sysinfo.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <list>
#include <mutex>
#include <thread>
#include <vector>
#include "gf2_common.h"
#include "../../include/m4ri_amd.h"

namespace {

constexpr uint8_t FLAG_EXCESS = 0x2;  // mzd.h:144
constexpr uint8_t FLAG_WINDOW = 0x4;  // mzd.h:150

// Locks of the host entry points.  One per DEVICE for everything that works on that device's staging arena, solver scratch and
// engine (ApiLock: host threads that drive different GPUs run side by side, threads on one GPU take turns -- round 3 had one lock
// for all devices), one short lock for the table of pinned matrices, one for the statistics.
constexpr int ARENA_DEVICES = 16;
std::mutex g_dev_mu[ARENA_DEVICES];
std::mutex g_pin_mu, g_stats_mu;
struct ApiLock {
  std::unique_lock<std::mutex> lk;
  ApiLock() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ARENA_DEVICES) dev = 0;
    lk = std::unique_lock<std::mutex>(g_dev_mu[dev]);
  }
};

// M4RI_AMD_STATS=1: at exit, print how many products went through the host entry points, the time
// spent in them (transfers included) and the bytes moved over PCIe -- for judging what an LD_PRELOAD
// run of a larger program (PLE, TRSM) spends where
struct ApiStats {
  double seconds = 0, h2d = 0, d2h = 0;
  long calls = 0;
  int populate_fallback = 0;  // a fresh result's pages were faulted by touching them: no MADV_POPULATE_WRITE here
  ~ApiStats() {
    if (calls && getenv("M4RI_AMD_STATS"))
      fprintf(stderr, "m4ri_amd: %ld products through the host entry points, %.3f s inside them, %.2f GiB up, %.2f GiB down\n",
              calls, seconds, h2d / 1073741824.0, d2h / 1073741824.0);
    if (populate_fallback && getenv("M4RI_AMD_STATS")) fprintf(stderr, "m4ri_amd: MADV_POPULATE_WRITE unavailable, fresh results were pre-faulted by touching their pages\n");
  }
} g_api_stats;

[[noreturn]] void die(const char *fmt, ...) {  // misc.c:36-42
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  abort();
}

#define HIPDIE(expr)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (hipError_t)(expr);                                                         \
    if (e_ != hipSuccess) die("m4ri_amd: HIP failure '%s' in %s (%s:%d)\n", hipGetErrorString(e_), #expr, __FILE__, __LINE__); \
  } while (0)

typedef mzd_t *(*mzd_init_fn)(rci_t, rci_t);
mzd_init_fn host_mzd_init() {
  // the caller will mzd_free() the result, so it has to come from the host program's libm4ri
  // allocator when there is one
  static mzd_init_fn fn = reinterpret_cast<mzd_init_fn>(dlsym(RTLD_DEFAULT, "mzd_init"));
  return fn;
}

mzd_t *result_init(rci_t r, rci_t c) {
  mzd_init_fn host_init = host_mzd_init();
  return host_init ? host_init(r, c) : m4ri_amd_mzd_init(r, c);
}

// Blocks of this size and more come straight from mmap in m4ri_amd_mzd_init: whole zero pages from the kernel, nothing to clear
constexpr size_t BIG_BLOCK = (size_t)8 << 20;
size_t big_len(size_t bytes) { return (bytes + 4095) & ~(size_t)4095; }

// The allocator's cache of large blocks (the counterpart of the reference's m4ri_mmc cache, mmc.c:44-116, which keeps the blocks
// BELOW the L3 size; this one keeps the few huge ones): m4ri_amd_mzd_free parks up to two mappings, at most 4 GiB together, and the
// next allocation of exactly that size takes one back -- its pages are already faulted in, which is the whole cost of a fresh
// 512 MiB result (33 ... 87 ms).  m4ri_amd_release_workspace() returns them to the system.
struct BigCache {
  std::mutex mu;
  struct Slot { void *p = nullptr; size_t len = 0; } slot[2];
  void *take(size_t len) {
    std::lock_guard<std::mutex> lk(mu);
    for (Slot &s : slot)
      if (s.p && s.len == len) { void *p = s.p; s = Slot{}; return p; }
    return nullptr;
  }
  bool park(void *p, size_t len) {
    static const bool enabled = !(getenv("M4RI_AMD_RESULT_CACHE") && atoi(getenv("M4RI_AMD_RESULT_CACHE")) == 0);
    std::lock_guard<std::mutex> lk(mu);
    size_t held = 0;
    for (Slot &s : slot) held += s.len;
    if (!enabled || held + len > ((size_t)4 << 30)) return false;
    for (Slot &s : slot)
      if (!s.p) { s.p = p; s.len = len; return true; }
    return false;
  }
  void drop() {
    std::lock_guard<std::mutex> lk(mu);
    for (Slot &s : slot) {
      if (s.p) munmap(s.p, s.len);
      s = Slot{};
    }
  }
} g_big_cache;

void *big_map(size_t len) {  // a fresh 2 MiB-aligned mapping: zero pages on demand
  const size_t al = (size_t)2 << 20;
  char *raw = static_cast<char *>(mmap(nullptr, len + al, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (raw == MAP_FAILED) return nullptr;
  char *q = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + al - 1) & ~(uintptr_t)(al - 1));
  if (q > raw) munmap(raw, (size_t)(q - raw));
  if (raw + len + al > q + len) munmap(q + len, (size_t)((raw + len + al) - (q + len)));
  return q;
}

void zero_with_threads(char *p, size_t bytes) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = hw >= 256 ? 32 : hw >= 64 ? 16 : 8, per = ((bytes / nt) + 4095) & ~(size_t)4095;
  std::vector<std::thread> th;
  size_t at = 0;
  try {
    for (; at < bytes; at += per) {
      const size_t len = bytes - at < per ? bytes - at : per;
      th.emplace_back([=] { memset(p + at, 0, len); });
    }
  } catch (...) {  // no more threads to be had: the rest here
    if (at < bytes) memset(p + at, 0, bytes - at);
  }
  for (std::thread &t : th) t.join();
}

mzd_t *descriptor(rci_t r, rci_t c) {  // mzd.c:142-150
  mzd_t *A = static_cast<mzd_t *>(calloc(1, sizeof(mzd_t)));
  if (!A) return nullptr;
  A->nrows        = r;
  A->ncols        = c;
  A->width        = c > 0 ? (c - 1) / 64 + 1 : 0;
  A->rowstride    = (A->width & 1) ? A->width + 1 : A->width;
  A->high_bitmask = (~(word)0) >> ((64 - c % 64) % 64);
  A->flags        = (A->high_bitmask != ~(word)0) ? 0x2 : 0;  // mzd.h:144: non-zero excess
  return A;
}

// A result that is still on its way.  mzd_mul(NULL, A, B, cutoff) is the region the reference's own bench times
// (bench/bench_multiplication.c:85-107), and a fresh 512 MiB matrix costs 33 ... 87 ms of page faults and clearing -- more than
// the 65536^3 product it receives -- when that happens in front of the upload.  Nothing needs C before the first block of the
// result is downloaded, so:
//   * with the host program's allocator (its mzd_init: malloc + ONE memset, misc.h:614-634) the call runs on a thread of its own
//     beside the upload and the first products; the first download waits for it (get());
//   * with this library's allocator the block is mmap'ed (instant, zero pages on demand) and worker threads fault its pages in
//     (MADV_POPULATE_WRITE: never changes a byte, so it may run beside the downloads) while the GPU works.
struct LateC {
  rci_t nrows = 0, ncols = 0;
  mzd_t *C = nullptr;
  std::thread maker;
  std::vector<std::thread> populate;
  std::once_flag once;
  mzd_t *get() {
    std::call_once(once, [this] { if (maker.joinable()) maker.join(); });
    return C;
  }
  mzd_t *finish() {
    get();
    for (std::thread &t : populate) t.join();
    populate.clear();
    return C;
  }
};

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23  // Linux 5.14
#endif

void late_begin(LateC &lc, rci_t r, rci_t c) {
  lc.nrows = r; lc.ncols = c;
  if (mzd_init_fn host_init = host_mzd_init()) {
    try {
      lc.maker = std::thread([&lc, host_init, r, c] { lc.C = host_init(r, c); });
    } catch (...) {  // no thread to be had (RLIMIT_NPROC, a pids cgroup): allocate here and now
      lc.C = host_init(r, c);
    }
    return;
  }
  // this library's allocator.  The product overwrites every valid word of the result, so a parked block of the same size is taken
  // as it is (its padding words cleared when rows have one); a block the cache does not have is a fresh mapping
  mzd_t *C = descriptor(r, c);
  if (!C) die("m4ri_amd: out of memory\n");
  lc.C = C;
  const size_t bytes = (size_t)r * (size_t)C->rowstride * 8, len = big_len(bytes);
  if (void *p = g_big_cache.take(len)) {
    C->data = static_cast<word *>(p);
    // M4RI_AMD_POISON_RESULT=1 (tests/test_gpu_host_pipeline.py): the parked block is filled with a pattern first, so that a run path
    // that failed to write some valid word of the result would show -- the check behind "taken as it is" 
    static const bool poison = getenv("M4RI_AMD_POISON_RESULT") && atoi(getenv("M4RI_AMD_POISON_RESULT")) != 0;
    if (poison) {
      std::memset(p, 0xA5, bytes);
      if (C->rowstride != C->width)
        for (rci_t i = 0; i < r; ++i) C->data[(int64_t)i * C->rowstride + C->width] = 0;
    } else if (C->rowstride != C->width) zero_with_threads(static_cast<char *>(p), bytes);
    return;
  }
  char *base = static_cast<char *>(big_map(len));
  if (!base) die("m4ri_amd: out of memory\n");
  C->data = reinterpret_cast<word *>(base);
  // How the pages of a NEW block come into being (M4RI_AMD_FRESH, a developer switch; measured in profiles/r04_fresh_result/):
  //   populate   worker threads fault them in beside the upload and the products (MADV_POPULATE_WRITE never changes a byte, so it
  //              may run beside the downloads; where the kernel lacks it, an atomic OR of zero per page)
  //   huge       the same after MADV_HUGEPAGE;  lazy: nothing, the downloads fault them;  zero: cleared by threads here and now
  //   auto (default): huge when memory is plentiful, else populate
  // Measured on the GPU box, 65536^3, every result a new block (profiles/r04_fresh_result.log; C given: 42.6 ms): huge 58 ... 61 ms,
  // populate 75 ... 86, lazy 73 ... 78, zero 68 ... 82 -- and 43.8 ms when the cache above has the block.  Huge pages are asked for only
  // when memory is plentiful (8 x the block available): on a host short of memory MADV_HUGEPAGE makes every fault compact memory
  // synchronously (a first 512 MiB block took 0.7 ... 1.7 s in the 8 GiB build container).
  static const char *mode_env = getenv("M4RI_AMD_FRESH");
  char mode = mode_env ? mode_env[0] : 'a';
  if (mode == 'a') {
    struct sysinfo si;
    mode = (sysinfo(&si) == 0 && (double)si.freeram * (double)si.mem_unit >= 8.0 * (double)bytes) ? 'h' : 'p';
  }
  if (mode == 'l') return;
  if (mode == 'z') { zero_with_threads(base, bytes); return; }
  if (mode == 'h') (void)madvise(base, len, 14 /* MADV_HUGEPAGE */);
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = hw >= 64 ? 16 : hw >= 16 ? 8 : 2, chunk = (size_t)16 << 20;
  // the workers walk the block front to back in 16 MiB chunks, round robin: the rows the first download will write come first
  try {
    for (size_t k = 0; k < nt; ++k)
      lc.populate.emplace_back([base, bytes, k, nt, chunk] {
        bool advise = true;
        for (size_t at = k * chunk; at < bytes; at += nt * chunk) {
          const size_t n = bytes - at < chunk ? bytes - at : chunk;
          if (advise && madvise(base + at, big_len(n), MADV_POPULATE_WRITE) == 0) continue;
          advise = false;
          g_api_stats.populate_fallback = 1;
          for (size_t off = 0; off < n; off += 4096) __atomic_fetch_or(reinterpret_cast<unsigned char *>(base + at + off), (unsigned char)0, __ATOMIC_RELAXED);
        }
      });
  } catch (...) {  // fewer (or no) helpers: the downloads fault what is left
  }
}

extern "C" hipError_t gf2_launch_copy_masked(hipStream_t s, word *C, int64_t cs, const word *A, int64_t as, int64_t rows, int64_t ncols);

// ---- staging arena ------------------------------------------------------------------------------
// Device copies of host operands live in ONE grow-only buffer that is re-carved on every call: no
// hipMalloc/hipFree per product (they cost more than a small product itself).
struct Arena {
  word *base  = nullptr;
  size_t cap  = 0;  // words
  size_t used = 0;
};
Arena g_arenas[ARENA_DEVICES];  // one per HIP device: run() works on whatever device is current
thread_local int g_arena_dev = 0;  // the device whose arena this thread is carving (set by arena_reserve, under that device's lock)
#define g_arena g_arenas[g_arena_dev]

void arena_reserve(size_t words) {
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  if (dev < 0 || dev >= ARENA_DEVICES) die("m4ri_amd: HIP device %d out of range\n", dev);
  g_arena_dev = dev;
  g_arena.used = 0;
  if (words <= g_arena.cap) return;
  HIPDIE(hipDeviceSynchronize());
  if (g_arena.base) HIPDIE(hipFree(g_arena.base));
  g_arena.base = nullptr; g_arena.cap = 0;
  HIPDIE(hipMalloc(reinterpret_cast<void **>(&g_arena.base), words * 8));
  g_arena.cap = words;
}

struct DevMat {
  word *p        = nullptr;
  int64_t stride = 0;
};

size_t dev_words(int64_t rows, int64_t ncols) {
  const int64_t w = words_of(ncols), st = (w + 1) & ~(int64_t)1;
  return (((size_t)(rows > 0 ? rows : 1) * (size_t)(st > 0 ? st : 2)) + 31) & ~(size_t)31;  // 256-byte granules
}

void dev_alloc(DevMat &d, int64_t rows, int64_t ncols) {
  const int64_t w = words_of(ncols);
  d.stride        = (w + 1) & ~(int64_t)1;  // even: rows stay 16-byte aligned
  d.p             = g_arena.base + g_arena.used;
  g_arena.used += dev_words(rows, ncols);
  if (g_arena.used > g_arena.cap) die("m4ri_amd: staging arena overrun (internal error)\n");
}

// host rows -> device matrix with zero excess
void upload(DevMat &d, const mzd_t *M) {
  dev_alloc(d, M->nrows, M->ncols);
  if (M->nrows == 0 || M->width == 0) return;
  if (d.stride != M->width)  // padding word must not stay uninitialised (it is never read as data,
    HIPDIE(hipMemsetAsync(d.p, 0, (size_t)M->nrows * d.stride * 8, 0));  // but keep it deterministic)
  HIPDIE(hipMemcpy2D(d.p, (size_t)d.stride * 8, M->data, (size_t)M->rowstride * 8, (size_t)M->width * 8,
                     (size_t)M->nrows, hipMemcpyHostToDevice));
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.h2d += (double)M->width * 8.0 * (double)M->nrows; }
  HIPDIE(m4ri_amd_mask_tail_dev(d.p, d.stride, M->nrows, M->ncols, nullptr));
}

// device result -> host C, writing words [0, width) only and, in the last word, only the bits
// inside high_bitmask when C is a window with excess
void download(const DevMat &d, mzd_t *C) {
  if (C->nrows == 0 || C->width == 0) return;
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.d2h += (double)C->width * 8.0 * (double)C->nrows; }
  const bool dangerous = (C->flags & FLAG_WINDOW) && (C->ncols % 64 != 0);
  if (!dangerous) {
    HIPDIE(hipMemcpy2D(C->data, (size_t)C->rowstride * 8, d.p, (size_t)d.stride * 8, (size_t)C->width * 8,
                       (size_t)C->nrows, hipMemcpyDeviceToHost));
    return;
  }
  if (C->width > 1)
    HIPDIE(hipMemcpy2D(C->data, (size_t)C->rowstride * 8, d.p, (size_t)d.stride * 8, (size_t)(C->width - 1) * 8,
                       (size_t)C->nrows, hipMemcpyDeviceToHost));
  std::vector<word> last((size_t)C->nrows);
  HIPDIE(hipMemcpy2D(last.data(), 8, d.p + (C->width - 1), (size_t)d.stride * 8, 8, (size_t)C->nrows, hipMemcpyDeviceToHost));
  const word mask = C->high_bitmask;
  for (rci_t i = 0; i < C->nrows; ++i) {
    word *w = C->data + (int64_t)i * C->rowstride + (C->width - 1);
    *w      = (*w & ~mask) | (last[(size_t)i] & mask);
  }
}

// ---- one transfer each way for products of a few MiB ----------------------------------------------------
// A product from host memory used to cost two blocking 2-D uploads (each with its own clear and tail-mask launch), the
// kernels, a blocking 2-D download and a device synchronisation: 85 ... 130 us whatever the size.  Operands of a few MiB
// are instead packed by the CPU into ONE pinned buffer per device (rows at the device stride, padding words cleared, the bits
// beyond the last column masked on the way -- a window's neighbours never reach the device), sent with one asynchronous copy,
// and the result comes back the same way and is merged into C's rows under the column mask by the CPU.
struct HostStage {
  word *p    = nullptr;
  size_t cap = 0;  // words
};
HostStage g_host_stage[ARENA_DEVICES];
constexpr size_t SMALL_STAGE_BYTES = (size_t)2 << 20;  // A + B + C above this: the 2-D copies straight from / to the caller's rows (4096^3 = 6 MiB: 0.32 ms packed against 0.25 ms direct)

word *host_stage(int dev, size_t words) {
  HostStage &h = g_host_stage[dev];
  if (words > h.cap) {
    if (h.p) { HIPDIE(hipDeviceSynchronize()); HIPDIE(hipHostFree(h.p)); }
    h.p = nullptr; h.cap = 0;
    HIPDIE(hipHostMalloc(reinterpret_cast<void **>(&h.p), words * 8, hipHostMallocDefault));
    h.cap = words;
  }
  return h.p;
}

// rows of M -> `dst` at `stride` words per row: valid words copied, the last one masked, padding words cleared
void pack_rows(word *dst, int64_t stride, const mzd_t *M) {
  const int64_t w = M->width;
  for (rci_t i = 0; i < M->nrows; ++i) {
    word *d       = dst + (int64_t)i * stride;
    const word *r = M->data + (int64_t)i * M->rowstride;
    if (w > 0) {
      memcpy(d, r, (size_t)w * 8);
      d[w - 1] &= M->high_bitmask;
    }
    for (int64_t k = w; k < stride; ++k) d[k] = 0;
  }
}

// `src` (rows at `stride`) -> C's rows: whole words, the last one under the column mask when C is a window
void unpack_rows(mzd_t *C, const word *src, int64_t stride) {
  const int64_t w = C->width;
  if (w == 0) return;
  const bool window = (C->flags & FLAG_WINDOW) != 0;
  const word mask   = C->high_bitmask;
  for (rci_t i = 0; i < C->nrows; ++i) {
    word *c       = C->data + (int64_t)i * C->rowstride;
    const word *r = src + (int64_t)i * stride;
    if (w > 1) memcpy(c, r, (size_t)(w - 1) * 8);
    c[w - 1] = window ? ((c[w - 1] & ~mask) | (r[w - 1] & mask)) : (r[w - 1] & mask);
  }
}

// ---- residency table (include/m4ri_amd.h part 3) ------------------------------------
// A pinned matrix keeps a device copy with the host layout (same rowstride), so any window into it
// is the same offset into the device copy.  Products read pinned operands where they are and leave a
// pinned result on the device (the host copy is stale until m4ri_amd_sync / m4ri_amd_unpin).
struct Pin {
  const word *hbase;  // host block
  size_t words;       // nrows * rowstride
  int64_t rowstride;
  rci_t nrows, ncols;
  word *dbase;
  bool dev_newer;  // written under the pin's device lock (set_newer), read there or by the cheap status query (is_newer): atomic accesses
  mzd_t *owner;
  int device;  // HIP device the copy lives on
};
inline void set_newer(Pin &p, bool v) { __atomic_store_n(&p.dev_newer, v, __ATOMIC_RELEASE); }
inline bool is_newer(const Pin &p) { return __atomic_load_n(&p.dev_newer, __ATOMIC_ACQUIRE); }
std::list<Pin> g_pins;  // a list: entries stay where they are while other threads pin and unpin (g_pin_mu guards the walk and the edits)

// Locking rule of the table: g_pin_mu guards the walk and the edits of the LIST; the lock of the device a pin lives on
// (g_dev_mu[pin.device]) guards the ENTRY -- its device copy, its dev_newer flag, its removal.  A product holds that lock for its
// whole duration (it runs on the pin's device or dies, operand()), so whoever changes or removes a pin takes the PIN'S device lock,
// not the lock of whatever device its own thread happens to have current (PinLock).  A Pin* from find_pin is therefore valid for as
// long as the caller holds the lock of the device the pin is on.
Pin *find_pin(const mzd_t *M) {
  if (!M || !M->data) return nullptr;
  std::lock_guard<std::mutex> pl(g_pin_mu);
  for (Pin &p : g_pins)
    if (M->data >= p.hbase && M->data < p.hbase + p.words && M->rowstride == p.rowstride) return &p;
  return nullptr;
}

int pin_device(const mzd_t *M) {  // the device M's pinned parent lives on, -1 when there is none (read under the list's lock)
  if (!M || !M->data) return -1;
  std::lock_guard<std::mutex> pl(g_pin_mu);
  for (Pin &p : g_pins)
    if (M->data >= p.hbase && M->data < p.hbase + p.words && M->rowstride == p.rowstride) return p.device;
  return -1;
}

// the pin of M with the lock of ITS device held and that device current (restored on the way out); p == nullptr: not pinned
struct PinLock {
  std::unique_lock<std::mutex> lk;
  Pin *p   = nullptr;
  int prev = -1, dev = -1;  // the caller's current device, the pin's (kept here: unpin erases *p before this object goes)
  explicit PinLock(const mzd_t *M) {
    for (;;) {  // until the look and the lock agree: giving up would report a pinned matrix as "not pinned" (a stale host copy unnoticed)
      const int d = pin_device(M);
      if (d < 0 || d >= ARENA_DEVICES) return;
      lk = std::unique_lock<std::mutex>(g_dev_mu[d]);
      Pin *q = find_pin(M);
      if (q && q->device == d) {
        p   = q;
        dev = d;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d) HIPDIE(hipSetDevice(d));
        return;
      }
      lk.unlock();  // unpinned, or re-pinned elsewhere, between the look and the lock: look again
    }
  }
  ~PinLock() {
    if (p && prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

// device view of a host operand: inside its pinned parent, or a staged upload (copy == false: space only)
DevMat operand(const mzd_t *M, bool copy, Pin **pin_out = nullptr) {
  DevMat d;
  Pin *p = find_pin(M);
  if (pin_out) *pin_out = p;
  if (p) {
    int dev = 0;
    HIPDIE(hipGetDevice(&dev));
    if (p->device != dev) die("m4ri_amd: matrix pinned on device %d used while device %d is current\n", p->device, dev);
    d.p      = p->dbase + (M->data - p->hbase);
    d.stride = p->rowstride;
    return d;
  }
  if (copy) upload(d, M);
  else dev_alloc(d, M->nrows, M->ncols);
  return d;
}

int norm_cutoff(int cutoff, const char *who) {  // strassen.c:348-354
  if (cutoff < 0) die("%s: cutoff must be >= 0.\n", who);
  return cutoff;  // 0 = engine default; >0 normalised inside m4ri_amd_mul_dev
}

// Large products from host memory, pipelined over blocks:  C_ij (+)= A_ik * B_kj  on a gi x gj x gk grid (2 x 2 x 1 when
// both m and n allow it, 2 x 2 x 2 from 65536^3 on, four row slabs of A and C for narrow products).  The uploads are
// blocking calls of this thread, the downloads of a second one (PCIe is full duplex), the products run on a non-blocking
// stream:  upload A_00, B_00 | P || upload A_01, B_10 | P -> C_00 complete: download || upload B_01 | P || ...
// so of the 1.5 GiB over PCIe at 65536^3 only the first pair of blocks (256 MiB) and the last block of C (128 MiB) stay
// exposed.  Same bits as the one-shot schedule (every step is an ordinary product or addmul).  Returns false when the
// product is too small to pay for it.
// m * l * n at or below which a product from host memory is computed on the host (m4ri_amd_set_small_product_threshold; the measured
// crossover against the GPU path on the GPU box, with the small leaf in the engine: 512^3 31 us on the host, 40 through the GPU;
// 576^3 49 against 47 -- profiles/r06_small_products_host_routine.log).  M4RI_AMD_SMALL_THRESHOLD overrides the default.
std::atomic<int64_t> g_small_threshold{getenv("M4RI_AMD_SMALL_THRESHOLD") ? atoll(getenv("M4RI_AMD_SMALL_THRESHOLD")) : ((int64_t)1 << 27)};
std::atomic<int64_t> g_small_count{0};  // products that took the host path

// Does a product of these dimensions go to the host routine?  m * l * n at most the threshold AND the routine's own cost
// (gf2_small_host_cost of small_host.cpp: word operations of ITS algorithm) at most threshold / 240 -- 2^27 / 240 = 559240, 39 us at the
// 0.07 ns per word operation it runs at: below the 25 ... 60 us a call through the GPU costs whatever its size.  The second bound is
// what keeps degenerate shapes -- 1 x 1 x 2^26, 2^26 x 1 x 1 -- away from a single-threaded loop.
extern "C" double gf2_small_host_cost(int64_t m, int64_t l, int64_t n);
bool small_product_wanted(int64_t m, int64_t l, int64_t n, int64_t threshold) {
  if (threshold <= 0 || m <= 0 || n <= 0) return false;
  if ((double)m * (double)l * (double)n > (double)threshold) return false;
  return gf2_small_host_cost(m, l, n) <= (double)threshold / 240.0;
}
std::atomic<size_t> g_pipeline_min_bytes{(size_t)64 << 20};  // A + B + C bytes from which blocks are used (16384^3: 2.62 -> 2.42 ms, 24576^3: 6.3 -> 5.3 ms); 0 disables (m4ri_amd_set_host_pipeline)
hipStream_t g_compute_stream[ARENA_DEVICES];

bool run_pipelined(mzd_t *C, const mzd_t *A, const mzd_t *B, bool add, int cutoff, LateC *late) {
  const int64_t m = A->nrows, l = A->ncols, n = B->ncols;
  const size_t bytes = ((size_t)m * A->width + (size_t)B->nrows * B->width + (size_t)m * (size_t)words_of(n)) * 8;
  auto Cget = [&]() -> mzd_t * { return late ? late->get() : C; };  // a result still being allocated: first needed by the first download
  if (g_pipeline_min_bytes == 0 || bytes < g_pipeline_min_bytes || m < 4 * 4096) return false;
  // cuts: rows of A / C on whole 4096-row tiles, columns and the inner dimension on whole words
  std::vector<int64_t> rcut, ccut, kcut;
  static const char *grid_env = getenv("M4RI_AMD_PIPE_GRID");  // "gi,gj[,gk]": developer override of the block grid
  int egi = 0, egj = 0, egk = 1;
  const int nenv = grid_env ? sscanf(grid_env, "%d,%d,%d", &egi, &egj, &egk) : 0;
  if (nenv >= 2 && egi >= 1 && egj >= 1 && egk >= 1 && m >= (int64_t)egi * 4096 && n >= (int64_t)egj * 64 && l >= (int64_t)egk * 64) {
    if (nenv < 3) egk = 1;
  } else if (n >= 16384) {
    egi = 2; egj = 2;
    // from 65536^3 on also two slices of the inner dimension: the blocks stay cubes (full Strassen depth) and the first
    // product waits for half as many bytes (43.1 instead of 46.0 ms; at 32768^3 the plain 2 x 2 is 3 % faster)
    egk = (l >= 65536 && m >= 65536 && n >= 65536) ? 2 : 1;
  } else {
    egi = (int)((m + ((m / 4 + 4095) / 4096) * 4096 - 1) / (((m / 4 + 4095) / 4096) * 4096)); egj = 1; egk = 1;
  }
  // cut number i of `parts` through `total`: on the coarsest grid of unit * 2^j (j <= 5) that moves it by at most 8 % of a part, else
  // on whole tiles of rows / whole words
  auto coarse_cut = [](int64_t total, int parts, int i, int64_t unit0, int64_t prev, bool rows) -> int64_t {
    const int64_t target = total * i / parts, fine = rows ? ((target + 4095) / 4096) * 4096 : (target / 64) * 64;
    for (int j = 5; j >= 1; --j) {
      const int64_t unit = unit0 << j, c = ((target + unit / 2) / unit) * unit;
      const int64_t off = c > target ? c - target : target - c;
      if (c > prev && c < total && off * 100 <= 8 * (total / parts)) return c;
    }
    return fine;
  };
  if (n >= 16384 || nenv >= 2) {
    // rows are cut on the coarsest grid of 4096 * 2^j rows that moves the cut by at most 8 % of a block: the engine gives a block of
    // k * 4096 * 2^L rows its full Strassen depth (engine.hip plan_row_blocks) -- 65664 rows cut 32768 + 32896, not 36864 + 28800
    rcut.push_back(0);
    for (int i = 1; i < egi; ++i) rcut.push_back(coarse_cut(m, egi, i, 4096, rcut.back(), true));
    rcut.push_back(m);
  } else {
    const int64_t srows = ((m / 4 + 4095) / 4096) * 4096;
    for (int64_t r = 0; r < m; r += srows) rcut.push_back(r);
    rcut.push_back(m);
  }
  // columns and inner bits likewise, on 1024 * 2^j bits (whole words at every Strassen level the block will use): no strips in the
  // blocks before the last
  ccut.push_back(0);
  for (int j = 1; j < egj; ++j) ccut.push_back(coarse_cut(n, egj, j, 1024, ccut.back(), false));
  ccut.push_back(n);
  kcut.push_back(0);
  for (int k = 1; k < egk; ++k) kcut.push_back(coarse_cut(l, egk, k, 1024, kcut.back(), false));
  kcut.push_back(l);
  const int gi = (int)rcut.size() - 1, gj = (int)ccut.size() - 1, gk = (int)kcut.size() - 1;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  hipStream_t &cs = g_compute_stream[dev];
  if (!cs) HIPDIE(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  auto R = [&](int i) { return rcut[(size_t)i + 1] - rcut[(size_t)i]; };
  auto Cc = [&](int j) { return ccut[(size_t)j + 1] - ccut[(size_t)j]; };
  auto K = [&](int k) { return kcut[(size_t)k + 1] - kcut[(size_t)k]; };
  size_t need = 0;
  for (int i = 0; i < gi; ++i)
    for (int k = 0; k < gk; ++k) need += dev_words(R(i), K(k));
  for (int k = 0; k < gk; ++k)
    for (int j = 0; j < gj; ++j) need += dev_words(K(k), Cc(j));
  for (int i = 0; i < gi; ++i)
    for (int j = 0; j < gj; ++j) need += dev_words(R(i), Cc(j));
  // M4RI_AMD_PIPE_W7=1: the top level as Strassen-Winograd over the 2 x 2 x 2 grid (7 block products instead of 8) when the halves are
  // equal and whole -- the schedule below.  Built on the round-4 review's suggestion, bit-exact, and measured SLOWER than the 8 classical
  // block products (65536^3: 44.6 against 41.7 ms, same box, alternating; profiles/r05_host_pipeline_timeline_65536.log): its third
  // product needs seven of the eight quadrants, which the PCIe link delivers at 16.8 ms at the earliest, while the classical order
  // never waits.  It stays an option for hosts with a faster link, not the default.
  static const bool w7_on = getenv("M4RI_AMD_PIPE_W7") && atoi(getenv("M4RI_AMD_PIPE_W7")) == 1;
  const bool w7 = w7_on && !add && gi == 2 && gj == 2 && gk == 2 && R(0) == R(1) && K(0) == K(1) && Cc(0) == Cc(1) && K(0) % 64 == 0 && Cc(0) % 64 == 0;
  if (w7) need += dev_words(R(0), K(0)) + dev_words(K(0), Cc(0)) + dev_words(R(0), Cc(0));
  arena_reserve(need);
  auto block_of = [&](const mzd_t *M, int64_t r0, int64_t r1, int64_t c0, int64_t c1) {  // mzd_init_window, mzd.c:159-177
    mzd_t S = *M;
    S.data  = M->data + r0 * M->rowstride + c0 / 64;
    S.nrows = (rci_t)(r1 - r0);
    S.ncols = (rci_t)(c1 - c0);
    S.width = (S.ncols + 63) / 64;
    S.high_bitmask = (~(word)0) >> ((64 - S.ncols % 64) % 64);
    // the block is a window unless it reaches M's own last column of a non-window M (then whole last words may be written)
    if ((M->flags & FLAG_WINDOW) || c1 != M->ncols) S.flags |= FLAG_WINDOW;
    return S;
  };
  const int nt = gi * gj;
  // M4RI_AMD_PIPE_TRACE=1: the timeline of this call on stderr (host clock for the copies, HIP events for the products)
  static const bool trace = getenv("M4RI_AMD_PIPE_TRACE") != nullptr;
  struct Mark { char what; int a, b; double t0, t1; };
  std::vector<Mark> marks;
  std::mutex marks_mu;
  timespec tl0;
  clock_gettime(CLOCK_MONOTONIC, &tl0);
  auto now_ms = [&]() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return 1e3 * (double)(t.tv_sec - tl0.tv_sec) + 1e-6 * (double)(t.tv_nsec - tl0.tv_nsec); };
  auto mark = [&](char what, int a, int b, double t0) { if (trace) { std::lock_guard<std::mutex> g(marks_mu); marks.push_back({what, a, b, t0, now_ms()}); } };
  hipEvent_t tr_base = nullptr;
  std::vector<hipEvent_t> tr_p0, tr_p1;
  double tr_base_ms = 0;
  std::vector<DevMat> dA((size_t)gi * gk), dB((size_t)gk * gj), dC((size_t)nt);
  std::vector<hipEvent_t> upA((size_t)gi * gk, nullptr), upB((size_t)gk * gj, nullptr), upC((size_t)nt, nullptr), done((size_t)nt, nullptr);
  auto ev = [](hipEvent_t &e) { HIPDIE(hipEventCreateWithFlags(&e, hipEventDisableTiming)); };
  auto upload_a = [&](int i, int k) {  // host rows -> device (blocking), tail masks on the null stream, then the event the products wait for
    hipEvent_t &e = upA[(size_t)i * gk + k];
    if (e) return;
    const mzd_t S = block_of(A, rcut[(size_t)i], rcut[(size_t)i + 1], kcut[(size_t)k], kcut[(size_t)k + 1]);
    const double t0 = trace ? now_ms() : 0;
    upload(dA[(size_t)i * gk + k], &S);
    mark('A', i, k, t0);
    ev(e);
    HIPDIE(hipEventRecord(e, nullptr));
  };
  auto upload_b = [&](int k, int j) {
    hipEvent_t &e = upB[(size_t)k * gj + j];
    if (e) return;
    const mzd_t S = block_of(B, kcut[(size_t)k], kcut[(size_t)k + 1], ccut[(size_t)j], ccut[(size_t)j + 1]);
    const double t0 = trace ? now_ms() : 0;
    upload(dB[(size_t)k * gj + j], &S);
    mark('B', k, j, t0);
    ev(e);
    HIPDIE(hipEventRecord(e, nullptr));
  };
  auto c_block = [&](int i, int j) { return block_of(Cget(), rcut[(size_t)i], rcut[(size_t)i + 1], ccut[(size_t)j], ccut[(size_t)j + 1]); };
  auto prepare_c = [&](int t) {
    if (upC[(size_t)t]) return;
    if (add) { const mzd_t S = c_block(t / gj, t % gj); upload(dC[(size_t)t], &S); }
    else dev_alloc(dC[(size_t)t], R(t / gj), Cc(t % gj));
    ev(upC[(size_t)t]);
    HIPDIE(hipEventRecord(upC[(size_t)t], nullptr));
  };
  // The downloads run on a second host thread: a blocking copy holds its thread, and the blocks of C can travel down
  // while the next operands travel up (PCIe is full duplex: 53 GiB/s each way on this box, tools/pcie_probe_2d.py).
  std::mutex dl_mu;
  std::condition_variable dl_cv;
  int issued = 0;  // entries of dl_order whose `done` event exists
  std::vector<int> dl_order((size_t)nt);  // the blocks of C in the order they are computed (and complete)
  // M4RI_AMD_PIPE_ORDER=c (developer): block columns of C outermost -- the first gi blocks need only the first block column of B
  static const bool col_major = getenv("M4RI_AMD_PIPE_ORDER") && getenv("M4RI_AMD_PIPE_ORDER")[0] == 'c';
  for (int q = 0; q < nt; ++q) dl_order[(size_t)q] = col_major ? (q % gi) * gj + (q / gi) : q;
  if (w7) { dl_order[0] = 0; dl_order[1] = 3; dl_order[2] = 1; dl_order[3] = 2; }  // C11, C22, C12, C21
  std::thread downloader([&]() {
    HIPDIE(hipSetDevice(dev));
    for (int q = 0; q < nt; ++q) {
      {
        std::unique_lock<std::mutex> lk(dl_mu);
        dl_cv.wait(lk, [&] { return issued > q; });
      }
      const int t = dl_order[(size_t)q];
      HIPDIE(hipEventSynchronize(done[(size_t)t]));
      mzd_t S = c_block(t / gj, t % gj);
      const double t0 = trace ? now_ms() : 0;
      download(dC[(size_t)t], &S);
      mark('D', t / gj, t % gj, t0);
    }
  });
  if (trace) {
    HIPDIE(hipEventCreate(&tr_base));
    HIPDIE(hipEventRecord(tr_base, cs));
    HIPDIE(hipEventSynchronize(tr_base));
    tr_base_ms = now_ms();
  }
  auto block_done = [&](int t) {  // block t of C is complete once everything issued on the compute stream so far has run: tell the downloader
    ev(done[(size_t)t]);
    HIPDIE(hipEventRecord(done[(size_t)t], cs));
    {
      std::lock_guard<std::mutex> lk(dl_mu);
      issued += 1;
    }
    dl_cv.notify_one();
  };
  // products in the order (block of C, inner slice); the uploads of step s + 1 are issued right after product s
  const int steps = w7 ? 0 : nt * gk;
  auto need_for = [&](int s) {
    const int t = dl_order[(size_t)(s / gk)], k = s % gk;
    upload_a(t / gj, k);
    upload_b(k, t % gj);
    prepare_c(t);
  };
  if (w7) {
    // Strassen-Winograd at the top (strassen.c:111-150 in the product form of engine.hip: A-side [A11, A12, S4, A22, S1, S2, S3], B-side
    // [B11, B21, B22, T4, T1, T2, T3]; P0 goes to all four quadrants, P1 to C11, P2 to C12, P3 to C21, P4 to C12 and C22, P5 to all but
    // C11, P6 to C21 and C22), in the order the quadrants ARRIVE: P0 and P1 need one quadrant of each operand, P6 seven of the eight,
    // the rest all.  The quadrants of C accumulate on the device (single-target products as addmul straight into their quadrant, the
    // others through one temporary), and each is downloaded when its last term has landed: C11 after P1, C22 after P5, C12 after P2,
    // C21 after P3.  The operand sums are chains in two temporaries: X = S3, S1, S2 (+= A11), S4 (+= A12); Y = T3, T1, T2 (+= B22), T4 (+= B21).
    const int64_t hm = R(0), hl = K(0), hn = Cc(0);
    DevMat X, Y, P;
    dev_alloc(X, hm, hl);
    dev_alloc(Y, hl, hn);
    dev_alloc(P, hm, hn);
    for (int t = 0; t < 4; ++t) prepare_c(t);
    auto qa = [&](int i, int k) -> const DevMat & { return dA[(size_t)i * gk + k]; };
    auto qb = [&](int k, int j) -> const DevMat & { return dB[(size_t)k * gj + j]; };
    auto wait_a = [&](int i, int k) { HIPDIE(hipStreamWaitEvent(cs, upA[(size_t)i * gk + k], 0)); };
    auto wait_b = [&](int k, int j) { HIPDIE(hipStreamWaitEvent(cs, upB[(size_t)k * gj + j], 0)); };
    auto xor3 = [&](const DevMat &D, const DevMat &U, const DevMat &V, int64_t rows, int64_t ncols) {
      HIPDIE(m4ri_amd_xor_dev(D.p, D.stride, U.p, U.stride, V.p, V.stride, rows, ncols, cs));
    };
    int pno = 0;
    auto product = [&](const DevMat &c, const DevMat &a, const DevMat &b, bool acc) {
      if (trace) { hipEvent_t e; HIPDIE(hipEventCreate(&e)); HIPDIE(hipEventRecord(e, cs)); tr_p0.push_back(e); }
      HIPDIE(m4ri_amd_mul_dev(c.p, c.stride, a.p, a.stride, b.p, b.stride, hm, hl, hn, acc ? 1 : 0, cutoff, cs));
      if (trace) { hipEvent_t e; HIPDIE(hipEventCreate(&e)); HIPDIE(hipEventRecord(e, cs)); tr_p1.push_back(e); }
      ++pno;
    };
    const size_t cq_bytes = (size_t)hm * (size_t)dC[0].stride * 8;
    for (int t = 0; t < 4; ++t) HIPDIE(hipStreamWaitEvent(cs, upC[(size_t)t], 0));
    upload_a(0, 0); upload_b(0, 0);                       // A11, B11
    wait_a(0, 0); wait_b(0, 0);
    product(dC[0], qa(0, 0), qb(0, 0), false);            // P0 -> C11, and the start of the other three
    for (int t = 1; t < 4; ++t) HIPDIE(hipMemcpyAsync(dC[(size_t)t].p, dC[0].p, cq_bytes, hipMemcpyDeviceToDevice, cs));
    upload_a(0, 1); upload_b(1, 0);                       // A12, B21
    wait_a(0, 1); wait_b(1, 0);
    product(dC[0], qa(0, 1), qb(1, 0), true);             // P1: C11 += A12 * B21
    block_done(0);
    upload_a(1, 0); upload_b(0, 1); upload_b(1, 1);       // A21, B12, B22
    wait_a(1, 0); wait_b(0, 1); wait_b(1, 1);
    xor3(X, qa(0, 0), qa(1, 0), hm, hl);                  // S3 = A11 + A21
    xor3(Y, qb(1, 1), qb(0, 1), hl, hn);                  // T3 = B22 + B12
    product(P, X, Y, false);                              // P6 -> C21, C22
    xor3(dC[3], dC[3], P, hm, hn);
    xor3(dC[2], dC[2], P, hm, hn);
    upload_a(1, 1);                                       // A22
    wait_a(1, 1);
    xor3(X, qa(1, 0), qa(1, 1), hm, hl);                  // S1 = A21 + A22
    xor3(Y, qb(0, 1), qb(0, 0), hl, hn);                  // T1 = B12 + B11
    product(P, X, Y, false);                              // P4 -> C12, C22
    xor3(dC[3], dC[3], P, hm, hn);
    xor3(dC[1], dC[1], P, hm, hn);
    xor3(X, X, qa(0, 0), hm, hl);                         // S2 = S1 + A11
    xor3(Y, Y, qb(1, 1), hl, hn);                         // T2 = T1 + B22
    product(P, X, Y, false);                              // P5 -> C12, C21, C22
    xor3(dC[3], dC[3], P, hm, hn);
    block_done(3);
    xor3(dC[2], dC[2], P, hm, hn);
    xor3(dC[1], dC[1], P, hm, hn);
    xor3(X, X, qa(0, 1), hm, hl);                         // S4 = S2 + A12
    product(dC[1], X, qb(1, 1), true);                    // P2: C12 += S4 * B22
    block_done(1);
    xor3(Y, Y, qb(1, 0), hl, hn);                         // T4 = T2 + B21
    product(dC[2], qa(1, 1), Y, true);                    // P3: C21 += A22 * T4
    block_done(2);
  } else need_for(0);
  for (int s = 0; s < steps; ++s) {
    const int t = dl_order[(size_t)(s / gk)], k = s % gk, i = t / gj, j = t % gj;
    HIPDIE(hipStreamWaitEvent(cs, upA[(size_t)i * gk + k], 0));
    HIPDIE(hipStreamWaitEvent(cs, upB[(size_t)k * gj + j], 0));
    HIPDIE(hipStreamWaitEvent(cs, upC[(size_t)t], 0));
    const DevMat &a = dA[(size_t)i * gk + k], &b = dB[(size_t)k * gj + j];
    if (trace) { hipEvent_t e; HIPDIE(hipEventCreate(&e)); HIPDIE(hipEventRecord(e, cs)); tr_p0.push_back(e); }
    HIPDIE(m4ri_amd_mul_dev(dC[(size_t)t].p, dC[(size_t)t].stride, a.p, a.stride, b.p, b.stride, R(i), K(k), Cc(j), (add || k > 0) ? 1 : 0, cutoff, cs));
    if (trace) { hipEvent_t e; HIPDIE(hipEventCreate(&e)); HIPDIE(hipEventRecord(e, cs)); tr_p1.push_back(e); }
    if (k == gk - 1) block_done(t);
    if (s + 1 < steps) need_for(s + 1);  // overlaps product s
  }
  downloader.join();
  HIPDIE(hipDeviceSynchronize());
  if (trace) {
    const double t_end = now_ms();
    for (size_t s = 0; s < tr_p0.size(); ++s) {
      float a = 0, b = 0;
      HIPDIE(hipEventElapsedTime(&a, tr_base, tr_p0[s]));
      HIPDIE(hipEventElapsedTime(&b, tr_base, tr_p1[s]));
      marks.push_back({'P', w7 ? (int)s : dl_order[s / (size_t)gk], w7 ? 0 : (int)s % gk, tr_base_ms + a, tr_base_ms + b});
      (void)hipEventDestroy(tr_p0[s]);
      (void)hipEventDestroy(tr_p1[s]);
    }
    (void)hipEventDestroy(tr_base);
    std::sort(marks.begin(), marks.end(), [](const Mark &x, const Mark &y) { return x.t0 < y.t0; });
    fprintf(stderr, "m4ri_amd pipeline %lld x %lld x %lld, grid %d x %d x %d%s: %.2f ms (A/B = upload of block (i,k)/(k,j), P = product (block of C, slice)%s, D = download)\n",
            (long long)m, (long long)l, (long long)n, gi, gj, gk, w7 ? ", Strassen-Winograd at the top" : "", t_end,
            w7 ? " -- here the seven in the order P0 P1 P6 P4 P5 P2 P3" : "");
    for (const Mark &k : marks) fprintf(stderr, "  %c(%d,%d) %8.2f .. %8.2f  (%6.2f ms)\n", k.what, k.a, k.b, k.t0, k.t1, k.t1 - k.t0);
  }
  for (auto *v : {&upA, &upB, &upC, &done})
    for (hipEvent_t e : *v)
      if (e) (void)hipEventDestroy(e);
  return true;
}

// the whole product: strassen == true runs the Strassen-Winograd engine, false a single leaf
// `late`: C == nullptr and the result is still being allocated (LateC; never with add, never empty)
mzd_t *run(mzd_t *C, const mzd_t *A, const mzd_t *B, bool add, bool strassen, int cutoff, LateC *late = nullptr) {
  ApiLock lk;
  const rci_t cm = A->nrows, cn = B->ncols;
  if (cm == 0 || cn == 0) return late ? late->get() : C;  // strassen.c:44
  struct Timer {
    timespec t0;
    Timer() { clock_gettime(CLOCK_MONOTONIC, &t0); }
    ~Timer() { timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.seconds += (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec); g_api_stats.calls += 1; }
  } timer;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  const bool same = (A == B);
  Pin *pinC = late ? nullptr : find_pin(C);
  // products the launch / PCIe floor of a call would dominate: this library's own host Four Russians (small_host.cpp), on an
  // initialised device and only for matrices that live in host memory (a pinned operand is already on the GPU)
  if (!late && !pinC && small_product_wanted(cm, A->ncols, cn, g_small_threshold.load()) && !find_pin(A) && !find_pin(B)) {
    lk.lk.unlock();  // the routine needs no device state: small products of many threads run side by side
    if (m4ri_amd_small_mul_host(C, A, B, add ? 1 : 0)) die("m4ri_amd: small product failed (internal error)\n");
    g_small_count += 1;
    return C;
  }
  if (strassen && !same && !pinC && !find_pin(A) && !find_pin(B) && run_pipelined(C, A, B, add, cutoff, late)) return late ? late->get() : C;
  // nothing pinned and a few MiB in all: one asynchronous transfer each way through the device's pinned buffer
  if (!late && !pinC && !find_pin(A) && !find_pin(B)) {
    const size_t wa = dev_words(A->nrows, A->ncols), wb = same ? 0 : dev_words(B->nrows, B->ncols), wc = dev_words(cm, cn);
    if ((wa + wb + wc) * 8 <= SMALL_STAGE_BYTES) {
      arena_reserve(wa + wb + wc);
      word *hs = host_stage(dev, wa + wb + wc);
      DevMat dA, dB, dC;
      dev_alloc(dA, A->nrows, A->ncols);
      if (same) dB = dA; else dev_alloc(dB, B->nrows, B->ncols);
      dev_alloc(dC, cm, cn);
      pack_rows(hs, dA.stride, A);
      if (!same) pack_rows(hs + wa, dB.stride, B);
      if (add) pack_rows(hs + wa + wb, dC.stride, C);
      const size_t up = (wa + wb + (add ? wc : 0)) * 8;
      HIPDIE(hipMemcpyAsync(dA.p, hs, up, hipMemcpyHostToDevice, nullptr));   // dA, dB, dC are consecutive in the arena, like hs
      if (strassen)
        HIPDIE(m4ri_amd_mul_dev(dC.p, dC.stride, dA.p, dA.stride, dB.p, dB.stride, A->nrows, A->ncols, B->ncols, add, cutoff, nullptr));
      else
        HIPDIE(m4ri_amd_m4rm_dev(dC.p, dC.stride, dA.p, dA.stride, dB.p, dB.stride, A->nrows, A->ncols, B->ncols, add, 0, nullptr));
      HIPDIE(hipMemcpyAsync(hs + wa + wb, dC.p, wc * 8, hipMemcpyDeviceToHost, nullptr));
      HIPDIE(hipStreamSynchronize(nullptr));
      unpack_rows(C, hs + wa + wb, dC.stride);
      { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.h2d += (double)up; g_api_stats.d2h += (double)wc * 8.0; }
      return C;
    }
  }
  // a pinned C whose last word is shared with other columns of its parent is computed in staging and
  // merged under the column mask; otherwise the engine writes straight into the parent
  const bool c_staged = !pinC || (cn % 64 != 0 && cn != pinC->ncols);
  arena_reserve((find_pin(A) ? 0 : dev_words(A->nrows, A->ncols)) + ((same || find_pin(B)) ? 0 : dev_words(B->nrows, B->ncols)) +
                (c_staged ? dev_words(cm, cn) : 0));
  const DevMat dA = operand(A, true);
  const DevMat dB = same ? dA : operand(B, true);
  DevMat dC;
  if (!c_staged) dC = operand(C, false);
  else if (!pinC) { if (add) upload(dC, C); else dev_alloc(dC, cm, cn); }
  else {
    dev_alloc(dC, cm, cn);
    if (add) {
      const DevMat src = operand(C, false);
      HIPDIE(hipMemcpy2DAsync(dC.p, (size_t)dC.stride * 8, src.p, (size_t)src.stride * 8, (size_t)C->width * 8, (size_t)C->nrows,
                              hipMemcpyDeviceToDevice, nullptr));
      HIPDIE(m4ri_amd_mask_tail_dev(dC.p, dC.stride, C->nrows, C->ncols, nullptr));
    }
  }
  if (strassen)
    HIPDIE(m4ri_amd_mul_dev(dC.p, dC.stride, dA.p, dA.stride, dB.p, dB.stride, A->nrows, A->ncols, B->ncols, add, cutoff, nullptr));
  else
    HIPDIE(m4ri_amd_m4rm_dev(dC.p, dC.stride, dA.p, dA.stride, dB.p, dB.stride, A->nrows, A->ncols, B->ncols, add, 0, nullptr));
  // a B that is a window of a pinned parent carries its neighbours' bits in the last word, and they
  // land in C's excess columns: clear them before the result leaves the staging buffer
  // (an unstaged pinned C spans its parent's full width, so its excess bits must be zero anyway:
  // mzd.h:115-121 -- mask there as well, or sync/unpin would carry the neighbours' bits to the host)
  if (c_staged || cn % 64 != 0) HIPDIE(m4ri_amd_mask_tail_dev(dC.p, dC.stride, cm, cn, nullptr));
  if (pinC) {
    if (c_staged) {
      const DevMat dst = operand(C, false);
      HIPDIE(gf2_launch_copy_masked(nullptr, dst.p, dst.stride, dC.p, dC.stride, C->nrows, C->ncols));
    }
    set_newer(*pinC, true);  // the host copy is stale until m4ri_amd_sync / m4ri_amd_unpin
  } else {
    if (late) C = late->get();
    download(dC, C);
  }
  HIPDIE(hipDeviceSynchronize());
  return C;
}

// ---- in/out matrices of the solvers (B of a TRSM, A of a PLE) --------------------------------------
// device view of a matrix that is read AND written: inside its pinned parent when the engine may write
// whole last words there, otherwise a staged copy (uploaded, or copied out of the pinned parent)
struct InOut {
  DevMat d;
  bool staged = true;
  Pin *pin    = nullptr;
};

size_t inout_words(const mzd_t *M) {
  Pin *p = find_pin(M);
  const bool staged = !p || (M->ncols % 64 != 0 && M->ncols != p->ncols);
  return staged ? dev_words(M->nrows, M->ncols) : 0;
}

InOut inout_begin(mzd_t *M) {
  InOut io;
  io.pin    = find_pin(M);
  io.staged = !io.pin || (M->ncols % 64 != 0 && M->ncols != io.pin->ncols);
  if (!io.staged) { io.d = operand(M, false); return io; }
  if (!io.pin) { upload(io.d, M); return io; }
  dev_alloc(io.d, M->nrows, M->ncols);
  const DevMat src = operand(M, false);
  HIPDIE(hipMemcpy2DAsync(io.d.p, (size_t)io.d.stride * 8, src.p, (size_t)src.stride * 8, (size_t)M->width * 8, (size_t)M->nrows,
                          hipMemcpyDeviceToDevice, nullptr));
  HIPDIE(m4ri_amd_mask_tail_dev(io.d.p, io.d.stride, M->nrows, M->ncols, nullptr));
  return io;
}

void inout_end(InOut &io, mzd_t *M) {
  if (io.pin) {
    if (io.staged) {
      const DevMat dst = operand(M, false);
      HIPDIE(gf2_launch_copy_masked(nullptr, dst.p, dst.stride, io.d.p, io.d.stride, M->nrows, M->ncols));
    }
    set_newer(*io.pin, true);
  } else {
    download(io.d, M);
  }
}

// B <- T^-1 B (left) or B <- B T^-1 (right) for a unit triangular T (triangular.c:41-514); T's other triangle and
// diagonal are never read
void run_trsm(bool upper, const mzd_t *T, mzd_t *B, int cutoff, bool right = false) {
  ApiLock lk;
  if (B->nrows == 0 || B->ncols == 0 || (!right && B->nrows <= 1) || (right && B->ncols <= 1)) return;  // one unknown per system: X = B
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  // T as the solver wants it: bits beyond its last column clean (trsm.hip: solve()).  A ragged window into a pinned parent
  // carries the parent's neighbouring columns in its last word: such a T is solved from a masked copy, like an in/out matrix
  Pin *pinT             = find_pin(T);
  const bool t_staged   = pinT && T->ncols % 64 != 0 && T->ncols != pinT->ncols;
  arena_reserve(((pinT && !t_staged) ? 0 : dev_words(T->nrows, T->ncols)) + inout_words(B));
  DevMat dT;
  if (t_staged) {
    dev_alloc(dT, T->nrows, T->ncols);
    const DevMat src = operand(T, false);
    HIPDIE(hipMemcpy2DAsync(dT.p, (size_t)dT.stride * 8, src.p, (size_t)src.stride * 8, (size_t)T->width * 8, (size_t)T->nrows, hipMemcpyDeviceToDevice, nullptr));
    HIPDIE(m4ri_amd_mask_tail_dev(dT.p, dT.stride, T->nrows, T->ncols, nullptr));
  } else {
    dT = operand(T, true);
  }
  InOut io        = inout_begin(B);
  if (right && upper) HIPDIE(m4ri_amd_trsm_upper_right_dev(dT.p, dT.stride, io.d.p, io.d.stride, B->nrows, B->ncols, cutoff, nullptr));
  else if (right)     HIPDIE(m4ri_amd_trsm_lower_right_dev(dT.p, dT.stride, io.d.p, io.d.stride, B->nrows, B->ncols, cutoff, nullptr));
  else if (upper)     HIPDIE(m4ri_amd_trsm_upper_left_dev(dT.p, dT.stride, io.d.p, io.d.stride, B->nrows, B->ncols, cutoff, nullptr));
  else                HIPDIE(m4ri_amd_trsm_lower_left_dev(dT.p, dT.stride, io.d.p, io.d.stride, B->nrows, B->ncols, cutoff, nullptr));
  inout_end(io, B);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
}

void pin_download(Pin &p) {
  if (!is_newer(p)) return;
  const int64_t width = words_of(p.ncols);
  if (p.nrows && width)
    HIPDIE(hipMemcpy2D(const_cast<word *>(p.hbase), (size_t)p.rowstride * 8, p.dbase, (size_t)p.rowstride * 8, (size_t)width * 8,
                       (size_t)p.nrows, hipMemcpyDeviceToHost));
  set_newer(p, false);
}

void pin_upload(Pin &p) {
  const int64_t width = words_of(p.ncols);
  if (p.nrows && width) {
    HIPDIE(hipMemcpy(p.dbase, p.hbase, p.words * 8, hipMemcpyHostToDevice));  // padding words included: same layout
    HIPDIE(m4ri_amd_mask_tail_dev(p.dbase, p.rowstride, p.nrows, p.ncols, nullptr));
    HIPDIE(hipDeviceSynchronize());
  }
  set_newer(p, false);
}

}  // namespace

extern "C" {

mzd_t *m4ri_amd_mzd_init(rci_t r, rci_t c) {  // mzd.c:142-157
  mzd_t *A = descriptor(r, c);
  if (!A) die("m4ri_amd_mzd_init: out of memory\n");
  if (r && c) {
    void *p = nullptr;
    const size_t bytes = (size_t)r * (size_t)A->rowstride * 8;
    if (bytes >= BIG_BLOCK) {
      // large blocks: a mapping of their own, 2 MiB-aligned (whole transparent huge pages where the system serves them) -- the kernel
      // hands out zero pages, so nothing is cleared and no page is touched before somebody needs it (a fresh 512 MiB block used to
      // cost 87 ms of page faults under one memset, 33 ms under 32 threads); a parked block of that size is cleared and reused.
      // (No MADV_HUGEPAGE: with the usual defrag = madvise setting it makes every fault compact memory synchronously.)
      const size_t len = big_len(bytes);
      if ((p = g_big_cache.take(len)) != nullptr) zero_with_threads(static_cast<char *>(p), bytes);
      else if ((p = big_map(len)) == nullptr) die("m4ri_amd_mzd_init: out of memory\n");
    } else {
      if (posix_memalign(&p, 64, bytes)) die("m4ri_amd_mzd_init: out of memory\n");
      memset(p, 0, bytes);
    }
    A->data = static_cast<word *>(p);
  }
  return A;
}

void m4ri_amd_mzd_free(mzd_t *A) {  // mzd.c:179-185
  if (!A) return;
  if (!(A->flags & FLAG_WINDOW) && A->data) {
    const size_t bytes = (size_t)A->nrows * (size_t)A->rowstride * 8;
    if (bytes >= BIG_BLOCK) {  // m4ri_amd_mzd_init: large blocks are mappings of their own; the cache may keep this one
      if (!g_big_cache.park(A->data, big_len(bytes))) munmap(A->data, big_len(bytes));
    } else free(A->data);
  }
  free(A);
}

// C == NULL at a product entry point: the product into a fresh matrix -- allocated beside the upload and the first products
// when it is large (LateC), in front of them when it is small or the product is empty (the zero matrix)
static mzd_t *run_fresh(mzd_t const *A, mzd_t const *B, bool strassen, int cutoff) {
  const rci_t r = A->nrows, c = B->ncols;
  const size_t bytes = (size_t)r * (size_t)((c + 63) / 64) * 8;
  if (bytes < BIG_BLOCK || A->ncols == 0) {
    mzd_t *C = result_init(r, c);
    return (r == 0 || c == 0 || A->ncols == 0) ? C : run(C, A, B, false, strassen, cutoff);
  }
  LateC late;
  late_begin(late, r, c);
  run(nullptr, A, B, false, strassen, cutoff, &late);
  return late.finish();
}

mzd_t *mzd_mul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) {  // strassen.c:345-365
  if (A->ncols != B->nrows) die("mzd_mul: A ncols (%d) need to match B nrows (%d).\n", A->ncols, B->nrows);
  cutoff = norm_cutoff(cutoff, "mzd_mul");
  if (C == NULL) return run_fresh(A, B, true, cutoff);
  else if (C->nrows != A->nrows || C->ncols != B->ncols)
    die("mzd_mul: C (%d x %d) has wrong dimensions, expected (%d x %d)\n", C->nrows, C->ncols, A->nrows, B->ncols);
  return run(C, A, B, false, true, cutoff);
}

mzd_t *mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) {  // strassen.c:675-700
  if (A->ncols != B->nrows) die("mzd_addmul: A ncols (%d) need to match B nrows (%d).\n", A->ncols, B->nrows);
  cutoff = norm_cutoff(cutoff, "mzd_addmul");
  if (C == NULL) return run_fresh(A, B, true, cutoff);  // 0 + A*B
  else if (C->nrows != A->nrows || C->ncols != B->ncols)
    die("mzd_addmul: C (%d x %d) has wrong dimensions, expected (%d x %d)\n", C->nrows, C->ncols, A->nrows, B->ncols);
  if (A->nrows == 0 || A->ncols == 0 || B->ncols == 0) return C;
  return run(C, A, B, true, true, cutoff);
}

mzd_t *_mzd_mul_even(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return run(C, A, B, false, true, cutoff); }
mzd_t *_mzd_addmul_even(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return run(C, A, B, true, true, cutoff); }
mzd_t *_mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return run(C, A, B, true, true, cutoff); }
mzd_t *_mzd_sqr_even(mzd_t *C, mzd_t const *A, int cutoff) { return run(C, A, A, false, true, cutoff); }
mzd_t *_mzd_addsqr_even(mzd_t *C, mzd_t const *A, int cutoff) { return run(C, A, A, true, true, cutoff); }

mzd_t *mzd_mul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k) {  // brilliantrussian.c:999-1012
  (void)k;
  if (A->ncols != B->nrows) die("mzd_mul_m4rm: A ncols (%d) need to match B nrows (%d).\n", A->ncols, B->nrows);
  if (C == NULL) return run_fresh(A, B, false, 0);
  else if (C->nrows != A->nrows || C->ncols != B->ncols)
    die("mzd_mul_m4rm: C (%d x %d) has wrong dimensions.\n", C->nrows, C->ncols);
  return run(C, A, B, false, false, 0);
}

mzd_t *mzd_addmul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k) {  // brilliantrussian.c:1014-1028
  (void)k;
  if (C == NULL) die("mzd_addmul_m4rm: C must not be NULL.\n");  // the reference dereferences C first (:1018)
  if (C->ncols == 0 || C->nrows == 0) return C;
  if (A->ncols != B->nrows) die("mzd_mul_m4rm A ncols (%d) need to match B nrows (%d) .\n", A->ncols, B->nrows);
  if (C->nrows != A->nrows || C->ncols != B->ncols) die("mzd_mul_m4rm: C has wrong dimensions.\n");
  return run(C, A, B, true, false, 0);
}

mzd_t *_mzd_mul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k, int clear) {  // brilliantrussian.c:1032
  (void)k;
  return run(C, A, B, clear == 0, false, 0);
}

int64_t m4ri_amd_set_small_product_threshold(int64_t ops) {
  const int64_t old = g_small_threshold;
  if (ops >= 0) g_small_threshold = ops;
  return old;
}

int64_t m4ri_amd_small_product_count(void) { return g_small_count; }

int m4ri_amd_small_product_wanted(int64_t m, int64_t l, int64_t n) { return small_product_wanted(m, l, n, g_small_threshold.load()) ? 1 : 0; }

// A + B + C bytes from which the host entry points pipeline a product over row slabs (0: never); returns the previous value
int64_t m4ri_amd_set_host_pipeline(int64_t min_bytes) {
  ApiLock lk;
  const int64_t old = (int64_t)g_pipeline_min_bytes;
  if (min_bytes >= 0) g_pipeline_min_bytes = (size_t)min_bytes;
  return old;
}

// ---- triangular solves: the reference's names, host mzd_t in / out -------------
void mzd_trsm_lower_left(mzd_t const *L, mzd_t *B, const int cutoff) {  // triangular.c:396-404
  if (L->ncols != B->nrows) die("mzd_trsm_lower_left: L ncols (%d) need to match B nrows (%d).\n", L->ncols, B->nrows);
  if (L->nrows != L->ncols) die("mzd_trsm_lower_left: L must be square and is found to be (%d) x (%d).\n", L->nrows, L->ncols);
  run_trsm(false, L, B, cutoff < 0 ? 0 : cutoff);
}
void _mzd_trsm_lower_left(mzd_t const *L, mzd_t *B, const int cutoff) { run_trsm(false, L, B, cutoff < 0 ? 0 : cutoff); }  // triangular.c:406
void _mzd_trsm_lower_left_russian(mzd_t const *L, mzd_t *B, int k) { (void)k; run_trsm(false, L, B, 0); }  // triangular_russian.c:206

void mzd_trsm_upper_left(mzd_t const *U, mzd_t *B, const int cutoff) {  // triangular.c:457-465
  if (U->ncols != B->nrows) die("mzd_trsm_upper_left: U ncols (%d) need to match B nrows (%d).\n", U->ncols, B->nrows);
  if (U->nrows != U->ncols) die("mzd_trsm_upper_left: U must be square and is found to be (%d) x (%d).\n", U->nrows, U->ncols);
  run_trsm(true, U, B, cutoff < 0 ? 0 : cutoff);
}
void _mzd_trsm_upper_left(mzd_t const *U, mzd_t *B, const int cutoff) { run_trsm(true, U, B, cutoff < 0 ? 0 : cutoff); }  // triangular.c:467
void _mzd_trsm_upper_left_russian(mzd_t const *U, mzd_t *B, int k) { (void)k; run_trsm(true, U, B, 0); }  // triangular_russian.c:50

void mzd_trsm_upper_right(mzd_t const *U, mzd_t *B, const int cutoff) {  // triangular.c:41-50
  if (U->nrows != B->ncols) die("mzd_trsm_upper_right: U nrows (%d) need to match B ncols (%d).\n", U->nrows, B->ncols);
  if (U->nrows != U->ncols) die("mzd_trsm_upper_right: U must be square and is found to be (%d) x (%d).\n", U->nrows, U->ncols);
  run_trsm(true, U, B, cutoff < 0 ? 0 : cutoff, true);
}
void _mzd_trsm_upper_right(mzd_t const *U, mzd_t *B, const int cutoff) { run_trsm(true, U, B, cutoff < 0 ? 0 : cutoff, true); }  // triangular.c:61
void mzd_trsm_lower_right(mzd_t const *L, mzd_t *B, const int cutoff) {  // triangular.c:301-310
  if (L->nrows != B->ncols) die("mzd_trsm_lower_right: L nrows (%d) need to match B ncols (%d).\n", L->nrows, B->ncols);
  if (L->nrows != L->ncols) die("mzd_trsm_lower_right: L must be square and is found to be (%d) x (%d).\n", L->nrows, L->ncols);
  run_trsm(false, L, B, cutoff < 0 ? 0 : cutoff, true);
}
void _mzd_trsm_lower_right(mzd_t const *L, mzd_t *B, const int cutoff) { run_trsm(false, L, B, cutoff < 0 ? 0 : cutoff, true); }  // triangular.c:312

// ---- PLE decomposition: the reference's names, host mzd_t / mzp_t in and out ---------
static rci_t run_ple(mzd_t *A, mzp_t *P, mzp_t *Q, bool pluq, bool russian) {
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  if (A->nrows == 0 || A->ncols == 0) {
    for (rci_t i = 0; i < A->nrows; ++i) P->values[i] = i;
    for (rci_t j = 0; j < A->ncols; ++j) Q->values[j] = j;
    return 0;
  }
  arena_reserve(inout_words(A));
  InOut io = inout_begin(A);
  int32_t rank = 0;
  HIPDIE((pluq ? m4ri_amd_pluq_dev : m4ri_amd_ple_dev)(io.d.p, io.d.stride, A->nrows, A->ncols, P->values, Q->values, &rank,
                                                       russian ? 0 : M4RI_AMD_PLE_CUTOFF, nullptr));
  inout_end(io, A);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return rank;
}

rci_t mzd_ple(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff) {  // ple.c:33-39
  (void)cutoff;
  if (P->length != A->nrows) die("mzd_ple: Permutation P length (%d) must match A nrows (%d)\n", P->length, A->nrows);
  if (Q->length != A->ncols) die("mzd_ple: Permutation Q length (%d) must match A ncols (%d)\n", Q->length, A->ncols);
  return run_ple(A, P, Q, false, false);
}
rci_t _mzd_ple(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff) { (void)cutoff; return run_ple(A, P, Q, false, false); }  // ple.c:62-171
rci_t _mzd_ple_russian(mzd_t *A, mzp_t *P, mzp_t *Q, int k) { (void)k; return run_ple(A, P, Q, false, true); }           // ple_russian.c:380-617

rci_t mzd_pluq(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff) {  // ple.c:41-48
  (void)cutoff;
  if (P->length != A->nrows) die("mzd_pluq: Permutation P length (%d) must match A nrows (%d)\n", P->length, A->nrows);
  if (Q->length != A->ncols) die("mzd_pluq: Permutation Q length (%d) must match A ncols (%d)\n", Q->length, A->ncols);
  return run_ple(A, P, Q, true, false);
}
rci_t _mzd_pluq(mzd_t *A, mzp_t *P, mzp_t *Q, const int cutoff) { (void)cutoff; return run_ple(A, P, Q, true, false); }  // ple.c:50-60
rci_t _mzd_pluq_russian(mzd_t *A, mzp_t *P, mzp_t *Q, int k) { (void)k; return run_ple(A, P, Q, true, true); }         // ple_russian.c:625-629

void mzd_apply_p_right_trans_tri(mzd_t *A, mzp_t const *Q) {  // mzp.c:279-293
  if (Q->length != A->ncols) die("mzd_apply_p_right_trans_tri: Permutation length (%d) must match A ncols (%d)\n", Q->length, A->ncols);
  if (A->nrows == 0 || A->ncols == 0) return;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  arena_reserve(inout_words(A));
  InOut io = inout_begin(A);
  HIPDIE(m4ri_amd_apply_p_right_trans_tri_dev(io.d.p, io.d.stride, A->nrows, A->ncols, Q->values, nullptr));
  inout_end(io, A);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
}

// ---- echelon forms and the column permutations (echelon.hip) -----------------------------------------------------
static rci_t run_echelonize(mzd_t *A, int full) {
  if (A->nrows == 0 || A->ncols == 0) return 0;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  arena_reserve(inout_words(A));
  InOut io = inout_begin(A);
  int32_t rank = 0;
  HIPDIE(m4ri_amd_echelonize_dev(io.d.p, io.d.stride, A->nrows, A->ncols, full, &rank, nullptr));
  inout_end(io, A);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return rank;
}
rci_t mzd_echelonize(mzd_t *A, int full) { return run_echelonize(A, full); }                  // echelonform.c:29-31
rci_t mzd_echelonize_m4ri(mzd_t *A, int full, int k) { (void)k; return run_echelonize(A, full); }  // echelonform.c:33-35
rci_t mzd_echelonize_pluq(mzd_t *A, int full) { return run_echelonize(A, full); }             // echelonform.c:37-139
rci_t mzd_echelonize_naive(mzd_t *A, int full) { return run_echelonize(A, full); }            // mzd.c:208-233: plain Gauss(-Jordan), the same pivoting rule, the same matrix
rci_t _mzd_echelonize_m4ri(mzd_t *A, const int full, int k, int heuristic, const double threshold) {  // brilliantrussian.c:603-841
  (void)k; (void)heuristic; (void)threshold;
  return run_echelonize(A, full);
}

static void run_apply_p_right(mzd_t *A, mzp_t const *P, int trans) {  // mzp.c:193-260
  if (A->nrows == 0 || A->ncols == 0) return;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  arena_reserve(inout_words(A));
  InOut io = inout_begin(A);
  HIPDIE(m4ri_amd_apply_p_right_dev(io.d.p, io.d.stride, A->nrows, A->ncols, P->values, P->length, trans, nullptr));
  inout_end(io, A);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
}
void mzd_apply_p_right(mzd_t *A, mzp_t const *P) { run_apply_p_right(A, P, 0); }
void mzd_apply_p_right_trans(mzd_t *A, mzp_t const *P) { run_apply_p_right(A, P, 1); }

// ---- the drivers over PLUQ: systems, kernels, inverses, row permutations (solve.hip) ---------------------------------
static void run_apply_p_left(mzd_t *A, mzp_t const *P, int trans) {  // mzp.c:65-81
  if (A->nrows == 0 || A->ncols == 0) return;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  arena_reserve(inout_words(A));
  InOut io = inout_begin(A);
  HIPDIE(m4ri_amd_apply_p_left_dev(io.d.p, io.d.stride, A->nrows, A->ncols, P->values, P->length, trans, nullptr));
  inout_end(io, A);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
}
void mzd_apply_p_left(mzd_t *A, mzp_t const *P) { run_apply_p_left(A, P, 0); }
void mzd_apply_p_left_trans(mzd_t *A, mzp_t const *P) { run_apply_p_left(A, P, 1); }

// A == nullptr-decomposition variant: rank/P/Q given (mzd_pluq_solve_left); otherwise A is decomposed in place
static int run_solve_left(mzd_t *A, mzd_t const *Adec, rci_t rank, mzp_t const *P, mzp_t const *Q, mzd_t *B, int cutoff, int check) {
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  int retval = 0;
  if (cutoff < 0) cutoff = 0;
  if (A) {
    arena_reserve(inout_words(A) + inout_words(B));
    InOut ia = inout_begin(A), ib = inout_begin(B);
    HIPDIE(m4ri_amd_solve_left_dev(ia.d.p, ia.d.stride, A->nrows, A->ncols, ib.d.p, ib.d.stride, B->nrows, B->ncols, cutoff, check, &retval, nullptr));
    inout_end(ia, A);
    inout_end(ib, B);
  } else {
    arena_reserve((find_pin(Adec) ? 0 : dev_words(Adec->nrows, Adec->ncols)) + inout_words(B));
    const DevMat dA = operand(Adec, true);
    InOut ib        = inout_begin(B);
    HIPDIE(m4ri_amd_pluq_solve_left_dev(dA.p, dA.stride, Adec->nrows, Adec->ncols, rank, P->values, Q->values, ib.d.p, ib.d.stride, B->nrows, B->ncols,
                                        cutoff, check, &retval, nullptr));
    inout_end(ib, B);
  }
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return retval;
}

int mzd_solve_left(mzd_t *A, mzd_t *B, int const cutoff, int const inconsistency_check) {  // solve.c:30-40
  if (A->ncols > B->nrows) die("mzd_solve_left: A ncols (%d) must be smaller than B nrows (%d).\n", A->ncols, B->nrows);
  if (B->nrows != (A->ncols > A->nrows ? A->ncols : A->nrows))
    die("mzd_solve_left: B nrows (%d) must be equal to max of A nrows (%d) and A ncols (%d).\n", B->nrows, A->nrows, A->ncols);
  return run_solve_left(A, nullptr, 0, nullptr, nullptr, B, cutoff, inconsistency_check);
}
int _mzd_solve_left(mzd_t *A, mzd_t *B, int const cutoff, int const inconsistency_check) {  // solve.c:123-152
  return run_solve_left(A, nullptr, 0, nullptr, nullptr, B, cutoff, inconsistency_check);
}
int mzd_pluq_solve_left(mzd_t const *A, rci_t rank, mzp_t const *P, mzp_t const *Q, mzd_t *B, int const cutoff, int const inconsistency_check) {  // solve.c:42-55
  if (A->ncols > B->nrows) die("mzd_pluq_solve_left: A ncols (%d) need to be lower than B nrows (%d).\n", A->ncols, B->nrows);
  if (P->length != A->nrows) die("mzd_pluq_solve_left: A nrows (%d) need to match P size (%d).\n", A->nrows, P->length);
  if (Q->length != A->ncols) die("mzd_pluq_solve_left: A ncols (%d) need to match Q size (%d).\n", A->ncols, P->length);
  return run_solve_left(nullptr, A, rank, P, Q, B, cutoff, inconsistency_check);
}
int _mzd_pluq_solve_left(mzd_t const *A, rci_t rank, mzp_t const *P, mzp_t const *Q, mzd_t *B, int const cutoff, int const inconsistency_check) {  // solve.c:57-121
  return run_solve_left(nullptr, A, rank, P, Q, B, cutoff, inconsistency_check);
}

mzd_t *mzd_kernel_left_pluq(mzd_t *A, int const cutoff) {  // solve.c:154-191
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  if (A->ncols == 0) return nullptr;  // rank 0 == ncols
  if (A->nrows == 0) {                // rank 0: every vector is in the kernel (solve.c:168-180 leaves the identity)
    mzd_t *I = result_init(A->ncols, A->ncols);
    for (rci_t i = 0; i < A->ncols; ++i) I->data[(int64_t)i * I->rowstride + i / 64] |= (word)1 << (i % 64);
    return I;
  }
  arena_reserve(inout_words(A) + dev_words(A->ncols, A->ncols));
  InOut ia = inout_begin(A);
  DevMat dR;
  dev_alloc(dR, A->ncols, A->ncols);  // the kernel has at most ncols columns; its real width is known after the PLUQ
  HIPDIE(hipMemsetAsync(dR.p, 0, (size_t)A->ncols * dR.stride * 8, nullptr));
  int32_t rank = 0;
  HIPDIE(m4ri_amd_kernel_left_pluq_dev(ia.d.p, ia.d.stride, A->nrows, A->ncols, dR.p, dR.stride, cutoff < 0 ? 0 : cutoff, &rank, nullptr));
  inout_end(ia, A);
  mzd_t *R = nullptr;
  if (rank < A->ncols) {
    R = result_init(A->ncols, A->ncols - rank);
    download(dR, R);
  }
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return R;
}

mzd_t *mzd_inv_m4ri(mzd_t *B, mzd_t const *A, int k) {  // brilliantrussian.c:971-997
  (void)k;
  if (A->nrows != A->ncols) die("mzd_inv_m4ri: A must be square and is found to be (%d) x (%d).\n", A->nrows, A->ncols);
  if (B == nullptr) B = result_init(A->nrows, A->ncols);
  else if (B->nrows != A->nrows || B->ncols != A->ncols) die("mzd_inv_m4ri: B (%d x %d) has wrong dimensions.\n", B->nrows, B->ncols);
  if (A->nrows == 0) return B;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  arena_reserve((find_pin(A) ? 0 : dev_words(A->nrows, A->ncols)) + inout_words(B));
  const DevMat dA = operand(A, true);
  InOut ib        = inout_begin(B);
  HIPDIE(m4ri_amd_inv_dev(ib.d.p, ib.d.stride, dA.p, dA.stride, A->nrows, nullptr));
  inout_end(ib, B);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return B;
}

// ---- transposes and triangular inverses (transpose.hip, trsm.hip) ----------------------------------------------------
mzd_t *mzd_transpose(mzd_t *DST, mzd_t const *A) {  // mzd.c:1118-1139
  if (DST == nullptr) DST = result_init(A->ncols, A->nrows);
  else if (DST->nrows != A->ncols || DST->ncols != A->nrows) die("mzd_transpose: Wrong size for return matrix.\n");
  if (A->nrows == 0 || A->ncols == 0) return DST;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  // the result is always built in staging: DST may be A itself, a window with neighbours in its last word, or pinned
  arena_reserve((find_pin(A) ? 0 : dev_words(A->nrows, A->ncols)) + dev_words(DST->nrows, DST->ncols));
  const DevMat dA = operand(A, true);
  DevMat dD;
  dev_alloc(dD, DST->nrows, DST->ncols);
  HIPDIE(m4ri_amd_transpose_dev(dD.p, dD.stride, dA.p, dA.stride, A->nrows, A->ncols, nullptr));
  if (Pin *pd = find_pin(DST)) {
    const DevMat dst = operand(DST, false);
    HIPDIE(gf2_launch_copy_masked(nullptr, dst.p, dst.stride, dD.p, dD.stride, DST->nrows, DST->ncols));
    set_newer(*pd, true);
  } else {
    download(dD, DST);
  }
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return DST;
}

static mzd_t *run_trtri_upper(mzd_t *A, const char *who) {  // triangular.c:518-547, triangular_russian.c:384-470
  if (A->nrows != A->ncols) die("%s: matrix must be square and is found to be (%d) x (%d).\n", who, A->nrows, A->ncols);
  if (A->nrows <= 1) return A;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  arena_reserve(inout_words(A));
  InOut io = inout_begin(A);
  HIPDIE(m4ri_amd_trtri_upper_dev(io.d.p, io.d.stride, A->nrows, nullptr));
  inout_end(io, A);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
  return A;
}
mzd_t *mzd_trtri_upper(mzd_t *A) { return run_trtri_upper(A, "mzd_trtri_upper"); }
mzd_t *mzd_trtri_upper_russian(mzd_t *A, int k) { (void)k; return run_trtri_upper(A, "mzd_trtri_upper_russian"); }

// ---- the table primitives of the elimination routines (elim.hip) -------------------------
static word *arena_raw(size_t words) {  // plain words from the staging arena (256-byte granules)
  word *p = g_arena.base + g_arena.used;
  g_arena.used += (words + 31) & ~(size_t)31;
  if (g_arena.used > g_arena.cap) die("m4ri_amd: staging arena overrun (internal error)\n");
  return p;
}

// how mzd_process_rowsN cuts the k-bit strip into N groups, lowest bits first (brilliantrussian.c:357-361,
// :394-398, :440-445, :490-494, :546-552)
static void split_k(int k, int n, int32_t *kb) {
  if (n == 1) { kb[0] = k; return; }
  if (n == 2) { kb[0] = k / 2; kb[1] = k - k / 2; return; }
  const int rem = k % n;
  for (int i = 0; i < n; ++i) kb[i] = k / n + ((i < n - 1 && rem >= n - 1 - i) ? 1 : 0);
}

static void run_process_rows(mzd_t *M, rci_t startrow, rci_t stoprow, rci_t startcol, int k, int nt, mzd_t const *const *T,
                             rci_t const *const *L) {
  if (stoprow <= startrow || M->ncols == 0) return;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  int32_t kb[6] = {0, 0, 0, 0, 0, 0};
  split_k(k, nt, kb);
  mzd_t W = *M;  // the rows that are touched, as a window of M
  W.data  = M->data + (int64_t)startrow * M->rowstride;
  W.nrows = stoprow - startrow;
  W.flags |= FLAG_WINDOW;
  size_t need = inout_words(&W) + 64 + (((size_t)3 * (size_t)W.nrows + 31) & ~(size_t)31);
  for (int t = 0; t < nt; ++t) need += (find_pin(T[t]) ? 0 : dev_words(T[t]->nrows, T[t]->ncols)) + ((((size_t)1 << kb[t]) / 2 + 1 + 31) & ~(size_t)31);
  arena_reserve(need);
  InOut io = inout_begin(&W);
  const word *dT[6];
  int64_t ts[6];
  const int32_t *dL[6];
  for (int t = 0; t < nt; ++t) {
    const DevMat d = operand(T[t], true);
    dT[t] = d.p; ts[t] = d.stride;
    const size_t n = (size_t)1 << kb[t];
    int32_t *l = reinterpret_cast<int32_t *>(arena_raw(n / 2 + 1));
    HIPDIE(hipMemcpyAsync(l, L[t], n * 4, hipMemcpyHostToDevice, nullptr));
    dL[t] = l;
  }
  int32_t *idx = reinterpret_cast<int32_t *>(arena_raw((size_t)3 * (size_t)W.nrows));
  HIPDIE(m4ri_amd_process_rows_dev(io.d.p, io.d.stride, M->width, 0, W.nrows, startcol, nt, kb, dT, ts, dL, idx, nullptr));
  inout_end(io, &W);
  HIPDIE(hipDeviceSynchronize());
  { std::lock_guard<std::mutex> sl(g_stats_mu); g_api_stats.calls += 1; }
}

void mzd_process_rows(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T, rci_t const *L) {  // brilliantrussian.c:213
  run_process_rows(M, startrow, endrow, startcol, k, 1, &T, &L);
}
void mzd_process_rows2(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1) {  // :350
  mzd_t const *T[2] = {T0, T1}; rci_t const *L[2] = {L0, L1};
  run_process_rows(M, startrow, endrow, startcol, k, 2, T, L);
}
void mzd_process_rows3(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2) {  // :386
  mzd_t const *T[3] = {T0, T1, T2}; rci_t const *L[3] = {L0, L1, L2};
  run_process_rows(M, startrow, endrow, startcol, k, 3, T, L);
}
void mzd_process_rows4(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2, mzd_t const *T3, rci_t const *L3) {  // :431
  mzd_t const *T[4] = {T0, T1, T2, T3}; rci_t const *L[4] = {L0, L1, L2, L3};
  run_process_rows(M, startrow, endrow, startcol, k, 4, T, L);
}
void mzd_process_rows5(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2, mzd_t const *T3, rci_t const *L3, mzd_t const *T4,
                       rci_t const *L4) {  // :481
  mzd_t const *T[5] = {T0, T1, T2, T3, T4}; rci_t const *L[5] = {L0, L1, L2, L3, L4};
  run_process_rows(M, startrow, endrow, startcol, k, 5, T, L);
}
void mzd_process_rows6(mzd_t *M, rci_t startrow, rci_t endrow, rci_t startcol, int k, mzd_t const *T0, rci_t const *L0, mzd_t const *T1,
                       rci_t const *L1, mzd_t const *T2, rci_t const *L2, mzd_t const *T3, rci_t const *L3, mzd_t const *T4,
                       rci_t const *L4, mzd_t const *T5, rci_t const *L5) {  // :537
  mzd_t const *T[6] = {T0, T1, T2, T3, T4, T5}; rci_t const *L[6] = {L0, L1, L2, L3, L4, L5};
  run_process_rows(M, startrow, endrow, startcol, k, 6, T, L);
}

void mzd_make_table(mzd_t const *M, rci_t r, rci_t c, int k, mzd_t *T, rci_t *L) {  // brilliantrussian.c:163-211
  const int twokay = 1 << k;
  std::vector<int32_t> jstar((size_t)twokay, 0);
  L[0] = 0;
  int js = 0;
  for (int i = 1; i < twokay; ++i) {
    L[i ^ (i >> 1)] = i;                                  // ord[i] = the reflected Gray code (graycode.c:31-62)
    if (r + __builtin_ctz((unsigned)i) >= M->nrows) js = i;  // inc[i-1] = the bit that flips: that row does not exist (:181)
    jstar[(size_t)i] = js;
  }
  if (M->ncols == 0 || c / 64 >= M->width) return;
  ApiLock lk;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  mzd_t W = *M;  // the k source rows
  W.data  = M->data + (int64_t)r * M->rowstride;
  W.nrows = (r + k <= M->nrows) ? k : (M->nrows > r ? M->nrows - r : 0);
  W.flags |= FLAG_WINDOW;
  Pin *pinT = find_pin(T);  // a pinned T IS its device copy: build the table there (the host copy goes stale like any pinned result)
  arena_reserve((find_pin(&W) ? 0 : dev_words(W.nrows, W.ncols)) + (pinT ? 0 : dev_words(T->nrows, T->ncols)) + (((size_t)twokay / 2 + 1 + 31) & ~(size_t)31) + 64);
  DevMat dM{};
  if (W.nrows > 0) dM = operand(&W, true);
  DevMat dT;
  const int64_t home = c / 64, wide = M->width - home;
  if (pinT) {
    dT = operand(T, false);
  } else {
    dev_alloc(dT, T->nrows, T->ncols);
    // the table's present content, unmasked (rows whose source row is missing keep it, and T[0] seeds the chain)
    HIPDIE(hipMemcpy2D(dT.p + home, (size_t)dT.stride * 8, T->data + home, (size_t)T->rowstride * 8, (size_t)wide * 8, (size_t)twokay, hipMemcpyHostToDevice));
  }
  int32_t *dj = reinterpret_cast<int32_t *>(arena_raw((size_t)twokay / 2 + 1));
  HIPDIE(hipMemcpyAsync(dj, jstar.data(), (size_t)twokay * 4, hipMemcpyHostToDevice, nullptr));
  HIPDIE(m4ri_amd_make_table_dev(dM.p, dM.stride, W.nrows, M->ncols, 0, c, k, dT.p, dT.p, dT.stride, dj, nullptr));
  if (pinT) set_newer(*pinT, true);
  else
    HIPDIE(hipMemcpy2D(T->data + (int64_t)T->rowstride + home, (size_t)T->rowstride * 8, dT.p + dT.stride + home, (size_t)dT.stride * 8, (size_t)wide * 8,
                       (size_t)(twokay - 1), hipMemcpyDeviceToHost));
  HIPDIE(hipDeviceSynchronize());
}

void gf2_release_staging(void) {  // called by m4ri_amd_release_workspace: the current device's arena, the parked result blocks
  g_big_cache.drop();
  ApiLock lk;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ARENA_DEVICES) return;
  if (g_host_stage[dev].p) {
    (void)hipDeviceSynchronize();
    (void)hipHostFree(g_host_stage[dev].p);
    g_host_stage[dev] = HostStage{};
  }
  Arena &a = g_arenas[dev];
  if (!a.base) return;
  (void)hipDeviceSynchronize();
  (void)hipFree(a.base);
  a = Arena{};
}

// Free a result the entry points allocated for C == NULL with the allocator it came from: the host
// program's mzd_free when result_init used its mzd_init (libm4ri's blocks come from its own caches,
// free() on them corrupts the heap), m4ri_amd_mzd_free otherwise.
void m4ri_amd_result_free(mzd_t *A) {
  typedef void (*free_fn)(mzd_t *);
  static free_fn host_free = dlsym(RTLD_DEFAULT, "mzd_init") ? reinterpret_cast<free_fn>(dlsym(RTLD_DEFAULT, "mzd_free")) : nullptr;
  if (host_free) host_free(A);
  else m4ri_amd_mzd_free(A);
}

// ---- part 3: residency ----------------------------------------------------------------------------
int m4ri_amd_pin(mzd_t *M) {
  if (!M || (M->flags & FLAG_WINDOW)) return -1;  // pin the owner of the block; windows into it follow
  {
    PinLock old(M);  // (the lock of the device the OLD entry is on, which need not be this thread's)
    if (old.p) {
      if (old.p->owner == M && old.p->hbase == M->data && old.p->nrows == M->nrows && old.p->ncols == M->ncols) return 0;
      // a matrix freed without unpin left this entry behind and the allocator reused its address: the
      // device copy belongs to a dead matrix -- drop it (without a download) and pin M afresh
      (void)hipFree(old.p->dbase);
      std::lock_guard<std::mutex> pl(g_pin_mu);
      for (auto it = g_pins.begin(); it != g_pins.end(); ++it)
        if (&*it == old.p) { g_pins.erase(it); break; }
    }
  }
  ApiLock lk;
  if (M->nrows == 0 || M->ncols == 0 || !M->data) return -1;
  int dev = 0;
  HIPDIE(hipGetDevice(&dev));
  HIPDIE(m4ri_amd_init(dev));
  Pin p{};
  p.device = dev;
  p.hbase = M->data; p.words = (size_t)M->nrows * (size_t)M->rowstride; p.rowstride = M->rowstride;
  p.nrows = M->nrows; p.ncols = M->ncols; p.owner = M;
  HIPDIE(hipMalloc(reinterpret_cast<void **>(&p.dbase), p.words * 8));
  pin_upload(p);
  {
    std::lock_guard<std::mutex> pl(g_pin_mu);
    for (Pin &q : g_pins)  // another thread pinned the same block in the meantime: keep theirs
      if (M->data >= q.hbase && M->data < q.hbase + q.words && M->rowstride == q.rowstride) { (void)hipFree(p.dbase); return 0; }
    g_pins.push_back(p);
  }
  return 0;
}

int m4ri_amd_sync(mzd_t *M) {
  PinLock pl(M);
  if (!pl.p) return -1;
  pin_download(*pl.p);
  return 0;
}

int m4ri_amd_host_modified(mzd_t *M) {
  PinLock pl(M);
  if (!pl.p) return -1;
  pin_upload(*pl.p);
  return 0;
}

int m4ri_amd_unpin(mzd_t *M) {
  PinLock pl(M);  // the lock of the pin's device: no product that reads or writes the copy is in flight
  if (!pl.p) return -1;
  pin_download(*pl.p);
  HIPDIE(hipFree(pl.p->dbase));
  {
    std::lock_guard<std::mutex> gl(g_pin_mu);
    for (auto it = g_pins.begin(); it != g_pins.end(); ++it)
      if (&*it == pl.p) { g_pins.erase(it); break; }
  }
  return 0;
}

// A status query: the list's short lock only, never the device lock -- it does not wait for a product running on the pin's device
// (the flag it reads is the one the LAST completed call left; a product in flight sets it before it returns)
int m4ri_amd_is_pinned(const mzd_t *M) {
  if (!M || !M->data) return 0;
  std::lock_guard<std::mutex> pl(g_pin_mu);
  for (const Pin &p : g_pins)
    if (M->data >= p.hbase && M->data < p.hbase + p.words && M->rowstride == p.rowstride) return is_newer(p) ? 2 : 1;
  return 0;
}

int gf2_multi_wanted(int64_t m, int64_t l, int64_t n);  // multi.hip

// mp.c:277-297 / :299-324.  Several devices + a product large enough: the top Strassen-Winograd
// sub-products go over the devices (multi.hip); else the single-GPU schedule.
static mzd_t *mul_mp(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff, bool add, const char *who) {
  if (A->ncols != B->nrows) die("%s: A ncols (%d) need to match B nrows (%d).\n", who, A->ncols, B->nrows);
  cutoff = norm_cutoff(cutoff, who);
  if (C == NULL) C = result_init(A->nrows, B->ncols);
  else if (C->nrows != A->nrows || C->ncols != B->ncols)
    die("%s: C (%d x %d) has wrong dimensions, expected (%d x %d)\n", who, C->nrows, C->ncols, A->nrows, B->ncols);
  if (add && (A->nrows == 0 || A->ncols == 0 || B->ncols == 0)) return C;
  bool pinned;
  pinned = find_pin(A) || find_pin(B) || find_pin(C);
  if (!pinned && gf2_multi_wanted(A->nrows, A->ncols, B->ncols)) {
    const int rc = m4ri_amd_mul_multi(C, A, B, add ? 1 : 0, cutoff, 0);
    if (rc) die("m4ri_amd: multi-device product failed (hipError_t %d: %s)\n", rc, hipGetErrorString((hipError_t)rc));
    return C;
  }
  return run(C, A, B, add, true, cutoff);
}
mzd_t *mzd_mul_mp(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return mul_mp(C, A, B, cutoff, false, "mzd_mul_mp"); }
mzd_t *mzd_addmul_mp(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) { return mul_mp(C, A, B, cutoff, true, "mzd_addmul_mp"); }

}  // extern "C"
