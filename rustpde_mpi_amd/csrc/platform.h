// Platform layer: one kernel source, two builds.
//
//  * HIP build (product): hipcc --offload-arch=gfx950.  Kernels are __global__ functions, one
//    workgroup per line, LDS = dynamic shared memory, phases separated by __syncthreads().
//  * EMU build (tests only, -DRPDE_EMU, plain g++): the SAME kernel bodies are compiled for the
//    host; a "phase" becomes a loop over thread ids and per-thread registers that live across a
//    barrier become (T x K) arrays.  It exists so that the index arithmetic of every line
//    program can be checked against the oracle without a GPU.  It is never built into, loaded
//    by, or reachable from the product library (see tests/emu/README.md).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>
#include <map>
#include <mutex>
#include <iterator>

#ifndef RPDE_EMU
#include <hip/hip_runtime.h>
#endif

namespace rpde {
#ifdef RPDE_EMU
using std::min;   // device code has HIP's min(int, int)
using std::max;
#endif

// ---------------------------------------------------------------------------------------------
// error handling (host)
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] inline void fail(const std::string& msg) { throw Error(msg); }
#define RPDE_REQUIRE(cond, msg)                                                           \
  do {                                                                                    \
    if (!(cond)) ::rpde::fail(std::string(msg) + "  [" #cond "] " + __FILE__ + ":" +      \
                              std::to_string(__LINE__));                                  \
  } while (0)

#ifndef RPDE_EMU
#define RPDE_HIP(call)                                                                    \
  do {                                                                                    \
    hipError_t _e = (call);                                                               \
    if (_e != hipSuccess)                                                                 \
      ::rpde::fail(std::string("HIP error: ") + hipGetErrorString(_e) + " in " #call " "  \
                   + __FILE__ + ":" + std::to_string(__LINE__));                          \
  } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// device code markers and the phase / barrier / per-thread-register model
#ifdef RPDE_EMU
#define RPDE_HD
#define RPDE_DEV
#define RPDE_DEVN
struct Blk {          // one workgroup
  int line;           // blockIdx.x
  int comp;           // blockIdx.y
  int T;              // blockDim.x
  double* lds;        // base of the workgroup's LDS
  int t0 = 0;         // (device only: first thread of the part of a workgroup that runs this line)
};
#define RPDE_PHASE(blk, tid) for (int tid = 0; tid < (blk).T; ++tid)
#define RPDE_PIN(x) ((void)0)
#define RPDE_SYNC(blk) ((void)0)
#define RPDE_MARK(blk, id) ((void)0)
#define RPDE_TLS(blk, type, name, K) std::vector<type> name##_st((size_t)(blk).T * (K)); const int name##_K = (K)
#define RPDE_T(name) (&name##_st[(size_t)tid * name##_K])
#define RPDE_TLS_PTR(name) (name##_st.data())          // a thread-local array handed to a function ...
#define RPDE_TPK(base, K) ((base) + (size_t)tid * (K))   // ... and the current thread's part of it inside a phase of that function
#else
#define RPDE_HD __host__ __device__
#define RPDE_DEV __device__ __forceinline__
#define RPDE_DEVN __device__ __forceinline__
struct Blk {
  int line, comp, T;
  double* lds;
  long long* trc;   // diagnostics: this workgroup's record of Program::trace (null: no tracing)
  int nm;           // marks written so far
  int t0 = 0;       // first thread of the part of the workgroup that runs this line: phases see tid = threadIdx.x - t0
                    // (hdct_pair_line: two halves of a workgroup run two transforms side by side)
};
// one (id, shader clock) pair per mark: id >= 0 = thread 0 reaches op id of the program, -1 = thread 0 leaves a barrier
constexpr int kTraceMarks = 126;
constexpr int kTraceStride = 4 + 2 * kTraceMarks;   // [0], [1]: 100 MHz wall clock at entry / exit, [2]: marks, then the pairs from [4]
#define RPDE_MARK(blk, id) do { if ((blk).trc) { if (threadIdx.x == 0 && (blk).nm < kTraceMarks) { \
    (blk).trc[4 + 2 * (blk).nm] = (id); (blk).trc[5 + 2 * (blk).nm] = (long long)clock64(); } ++(blk).nm; } } while (0)
// The thread index is laundered through an empty volatile asm in every phase: otherwise the
// compiler hoists the per-thread index arithmetic of ALL ops out of the interpreter loop, keeps it
// live across the whole program and spills it in the prologue (160 B/lane of scratch writes per
// line -- more HBM traffic than the line itself; rocprofv3 WRITE_SIZE, profiles/README.md)
__device__ __forceinline__ int rpde_tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < 1024);
  return t;
}
// the value lane l of the wave holds, as a wave-uniform number (two v_readlane_b32)
__device__ __forceinline__ double rpde_lane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
#define RPDE_PHASE(blk, tid) for (int tid = rpde_tid() - (blk).t0, _once = 1; _once; _once = 0)
// A value loaded from global memory is pinned where all loads of the phase have been issued: the compiler
// may not sink the load into the (per-lane predicated) block of its only user, where it would be followed by
// an s_waitcnt vmcnt(0) -- one dependent memory round trip per element (tools/check_load_issue.py)
#define RPDE_PIN(x) asm volatile("" : "+v"(x))
#define RPDE_SYNC(blk) do { __syncthreads(); RPDE_MARK(blk, -1); } while (0)
#define RPDE_TLS(blk, type, name, K) type name[K]
#define RPDE_T(name) name
#define RPDE_TLS_PTR(name) (name)
#define RPDE_TPK(base, K) (base)
#endif

// ---------------------------------------------------------------------------------------------
// device memory + stream (host side)
struct Stream {
#ifdef RPDE_EMU
  int dummy = 0;
#else
  hipStream_t s = nullptr;
#endif
};

constexpr size_t kEmuGuardBytes = 256;

// Device memory of every engine and operator of the process comes out of a few large slabs: one raw allocation (hipMalloc)
// of a multiple of the 2 MB a page-table block covers, mapped ONCE, and a best-fit free list with coalescing on top of it.
// Nothing this library allocates shares a 2 MB block of the runtime's own small-allocation heap (ROCr packs hipMalloc requests
// below 2 MB into 2 MB blocks it also uses for its internal objects, and gives such blocks back piecemeal), no mapping
// appears or disappears while a step runs, and an engine's 300 tables cost 300 free-list operations instead of 300 driver
// calls (DESIGN.md section 10-0 has the measurements that led here).
//  * **Keyed by device** (round 5): a slab belongs to the device that was current when it was allocated; an allocation only
//    ever takes space from slabs of the device that is current NOW (every C-ABI entry selects its handle's device first), and
//    a block goes back to the slab it came from whatever device is current at free time.
//  * trim(device): slabs nobody uses go back to the driver -- called when an engine / operator set is destroyed and when an
//    allocation fails.  The first slab of a device is small (256 MB), later ones 1 GB.
//  * Guards (RPDE_ARENA_GUARD=1, tests): every block is followed by one granule filled with a byte pattern, the unused tail
//    of its last granule as well; check() / free() count blocks whose pattern was overwritten -- a contiguous slab turns
//    an overrun into a silent hit on the neighbour, the guards keep WRITES beyond a block detectable (reads beyond a block
//    stay invisible here: RPDE_ARENA=0 gives every buffer its own mapping for that).
// The bookkeeping is a template over the raw allocator so that the keying logic runs in a unit test without a second GPU
// (rpde_arena_selftest: a host backend with two pretended devices).
template <class Backend>
class ArenaT {
 public:
  static constexpr size_t kPage = 4096;
  static constexpr size_t kHuge = size_t(2) << 20;
  static constexpr size_t kFirstSlabBytes = size_t(256) << 20;
  static constexpr size_t kSlabBytes = size_t(1) << 30;
  static constexpr unsigned char kGuardByte = 0xA5;
  explicit ArenaT(bool guard = false) : guard_(guard) {}
  ~ArenaT() = default;   // process exit: the driver reclaims the slabs (no HIP calls from static destructors)
  void* alloc(size_t bytes) {
    const size_t want = bytes ? bytes : 8;
    const size_t n = round_up(want, kPage) + (guard_ ? kPage : 0);
    const int dev = Backend::current_device();
    std::lock_guard<std::mutex> lk(mu_);
    void* p = take(n, want, dev);
    if (!p) {
      if (!add_slab(n, dev)) { trim_locked(-1); if (!add_slab(n, dev)) return nullptr; }
      p = take(n, want, dev);
    }
    if (p && guard_) Backend::fill(static_cast<char*>(p) + want, kGuardByte, n - want, dev);
    return p;
  }
  void free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu_);
    auto u = used_.find(static_cast<char*>(p));
    if (u == used_.end()) return;
    if (guard_ && !guard_intact(u->first, u->second)) ++violations_;
    Slab& sl = slabs_[u->second.slab];
    char* a = u->first; size_t n = u->second.bytes;
    used_.erase(u);
    auto nx = sl.free.lower_bound(a);                       // coalesce with the free neighbours inside this slab
    if (nx != sl.free.end() && a + n == nx->first) { n += nx->second; nx = sl.free.erase(nx); }
    if (nx != sl.free.begin()) {
      auto pv = std::prev(nx);
      if (pv->first + pv->second == a) { a = pv->first; n += pv->second; sl.free.erase(pv); }
    }
    sl.free[a] = n;
  }
  // slabs of `device` (-1: of every device) that hold no block go back to the driver; returns the bytes released
  size_t trim(int device = -1) { std::lock_guard<std::mutex> lk(mu_); return trim_locked(device); }
  // blocks in use whose guard pattern was overwritten (+ those found at free time since the last call); 0 without guards
  long check() {
    std::lock_guard<std::mutex> lk(mu_);
    long bad = violations_;
    violations_ = 0;
    if (guard_) for (auto& u : used_) if (!guard_intact(u.first, u.second)) ++bad;
    return bad;
  }
  bool guarded() const { return guard_; }
  size_t slab_bytes(int device = -1) {
    std::lock_guard<std::mutex> lk(mu_); size_t t = 0;
    for (auto& s : slabs_) if (s.base && (device < 0 || s.device == device)) t += s.size;
    return t;
  }
  size_t used_bytes(int device = -1) {
    std::lock_guard<std::mutex> lk(mu_); size_t t = 0;
    for (auto& u : used_) if (device < 0 || slabs_[u.second.slab].device == device) t += u.second.bytes;
    return t;
  }
  int device_of(void* p) {   // the device whose slab holds p; -1: not ours
    std::lock_guard<std::mutex> lk(mu_);
    auto u = used_.find(static_cast<char*>(p));
    return u == used_.end() ? -1 : slabs_[u->second.slab].device;
  }
  template <class F> void for_each_used(F f) { std::lock_guard<std::mutex> lk(mu_); for (auto& u : used_) f(u.first, u.second.bytes); }

 private:
  struct Slab { char* base = nullptr; size_t size = 0; int device = -1; std::map<char*, size_t> free; };
  struct Used { size_t bytes; size_t slab; size_t want; };
  static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
  void* take(size_t n, size_t want, int dev) {                // best fit over the slabs of THIS device
    size_t bs = 0; std::map<char*, size_t>::iterator bi; bool found = false;
    for (size_t k = 0; k < slabs_.size(); ++k) {
      if (!slabs_[k].base || slabs_[k].device != dev) continue;
      for (auto it = slabs_[k].free.begin(); it != slabs_[k].free.end(); ++it)
        if (it->second >= n && (!found || it->second < bi->second)) { bs = k; bi = it; found = true; }
    }
    if (!found) return nullptr;
    char* a = bi->first; const size_t rest = bi->second - n;
    slabs_[bs].free.erase(bi);
    if (rest) slabs_[bs].free[a + n] = rest;
    used_[a] = Used{n, bs, want};
    return a;
  }
  bool add_slab(size_t n, int dev) {
    bool first = true;
    for (auto& s : slabs_) if (s.base && s.device == dev) first = false;
#ifdef RPDE_FIRST_SLAB_1GB     // experiment build only (tools/r05_call4.sh): the slab sizes of round 4
    const size_t size = std::max(kSlabBytes, round_up(n, kHuge));
    (void)first;
#else
    const size_t size = std::max(first ? kFirstSlabBytes : kSlabBytes, round_up(n, kHuge));
#endif
    void* p = Backend::raw_alloc(size, dev);
    if (!p) return false;
    Slab* sl = nullptr;
    for (auto& s : slabs_) if (!s.base) { sl = &s; break; }
    if (!sl) { slabs_.emplace_back(); sl = &slabs_.back(); }
    sl->base = static_cast<char*>(p); sl->size = size; sl->device = dev; sl->free.clear(); sl->free[sl->base] = size;
    return true;
  }
  size_t trim_locked(int device) {
    size_t released = 0;
    for (auto& s : slabs_)
      if (s.base && (device < 0 || s.device == device) && s.free.size() == 1 && s.free.begin()->second == s.size) {
        Backend::raw_free(s.base, s.device);
        released += s.size;
        s.base = nullptr; s.size = 0; s.device = -1; s.free.clear();
      }
    return released;
  }
  bool guard_intact(char* a, const Used& u) {
    const size_t len = u.bytes - u.want;
    scratch_.resize(len);
    Backend::read(scratch_.data(), a + u.want, len, slabs_[u.slab].device);
    for (size_t i = 0; i < len; ++i) if (scratch_[i] != kGuardByte) return false;
    return true;
  }
  const bool guard_;
  std::mutex mu_;
  std::vector<Slab> slabs_;
  std::map<char*, Used> used_;
  std::vector<unsigned char> scratch_;
  long violations_ = 0;
};

// the backend of rpde_arena_selftest: host memory, `device` is whatever the test pretends is current
struct ArenaHostBackend {
  static int& current() { static thread_local int d = 0; return d; }
  static int current_device() { return current(); }
  static void* raw_alloc(size_t n, int) { return std::malloc(n); }
  static void raw_free(void* p, int) { std::free(p); }
  static void fill(void* p, unsigned char b, size_t n, int) { std::memset(p, b, n); }
  static void read(void* dst, const void* src, size_t n, int) { std::memcpy(dst, src, n); }
};

#ifndef RPDE_EMU
struct ArenaHipBackend {
  static int current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }
  static void* raw_alloc(size_t n, int) {      // the device is current (alloc() read it from the runtime)
    void* p = nullptr;
    if (hipMalloc(&p, n) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
  }
  struct OnDevice {                            // run a few runtime calls with `dev` current, then restore
    int prev = 0; bool sw = false;
    explicit OnDevice(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) sw = hipSetDevice(dev) == hipSuccess; }
    ~OnDevice() { if (sw) (void)hipSetDevice(prev); }
  };
  static void raw_free(void* p, int dev) { OnDevice g(dev); (void)hipFree(p); }
  static void fill(void* p, unsigned char b, size_t n, int dev) { OnDevice g(dev); (void)hipMemset(p, b, n); }
  static void read(void* dst, const void* src, size_t n, int dev) { OnDevice g(dev); (void)hipDeviceSynchronize(); (void)hipMemcpy(dst, src, n, hipMemcpyDeviceToHost); }
};
class DevArena {
 public:
  static ArenaT<ArenaHipBackend>& get() {
    static ArenaT<ArenaHipBackend> a([] { const char* e = std::getenv("RPDE_ARENA_GUARD"); return e && std::atoi(e) != 0; }());
    return a;
  }
};
inline bool dev_arena_on() {   // A/B only (tools/archive/fault_hunt_r04c.sh): RPDE_ARENA=0 = one hipMalloc per buffer, as rounds 1-3
  static const bool on = [] { const char* e = std::getenv("RPDE_ARENA"); return !e || std::atoi(e) != 0; }();
  return on;
}
#endif
// give unused slabs of the current device back to the driver (after an engine / operator set has been destroyed)
inline size_t dev_trim() {
#ifdef RPDE_EMU
  return 0;
#else
  return dev_arena_on() ? DevArena::get().trim(ArenaHipBackend::current_device()) : 0;
#endif
}

#ifdef RPDE_EMU
// RPDE_ALLOC_LOG=<file> (emulation build only): the sequence of device allocations / uploads / frees an engine performs, one
// line each ("A id bytes", "U id bytes", "F id") -- tools/fault_repro replays it with bare HIP calls on a GPU
inline void alloc_log(char op, const void* p, size_t bytes) {
  static const char* path = std::getenv("RPDE_ALLOC_LOG");
  if (!path) return;
  static std::map<const void*, long> ids;
  static long next = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  long id;
  if (op == 'A') { id = next++; ids[p] = id; }
  else {
    auto it = ids.upper_bound(p);            // an upload may target the middle of a buffer
    if (it == ids.begin()) return;
    id = std::prev(it)->second;
    if (op == 'F') ids.erase(std::prev(it));
  }
  if (FILE* f = std::fopen(path, "a")) { std::fprintf(f, "%c %ld %zu\n", op, id, bytes); std::fclose(f); }
}
#endif
inline void* dev_alloc(size_t bytes) {
#ifdef RPDE_EMU
  // a NaN-filled guard in front of every buffer: a read below the start of a table or array
  // (a page fault on the device when the buffer opens an allocation) poisons the results here
  char* raw = static_cast<char*>(std::calloc(bytes + kEmuGuardBytes + 1, 1));
  RPDE_REQUIRE(raw, "host allocation failed");
  std::memset(raw, 0xFF, kEmuGuardBytes);
  alloc_log('A', raw + kEmuGuardBytes, bytes);
  return raw + kEmuGuardBytes;
#else
  void* p = nullptr;
  if (dev_arena_on()) {
    p = DevArena::get().alloc(bytes);
    RPDE_REQUIRE(p, "device allocation failed (hipMalloc of a slab)");
  } else {
    RPDE_HIP(hipMalloc(&p, bytes ? bytes : 8));
  }
  RPDE_HIP(hipMemset(p, 0, bytes ? bytes : 8));
  return p;
#endif
}
inline void dev_free(void* p) {
#ifdef RPDE_EMU
  if (p) { alloc_log('F', p, 0); std::free(static_cast<char*>(p) - kEmuGuardBytes); }
#else
  if (!p) return;
  if (dev_arena_on()) DevArena::get().free(p);
  else (void)hipFree(p);
#endif
}
inline void dev_upload(void* dst, const void* src, size_t bytes) {
#ifdef RPDE_EMU
  alloc_log('U', dst, bytes);
  std::memcpy(dst, src, bytes);
#else
  RPDE_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
#endif
}
inline void dev_download(void* dst, const void* src, size_t bytes) {
#ifdef RPDE_EMU
  std::memcpy(dst, src, bytes);
#else
  RPDE_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
#endif
}
// pitched copies: `rows` rows of `cols` doubles; host is dense, device has leading dimension ld
inline void dev_upload2d(double* dst, long ld, const double* src, long rows, long cols) {
#ifdef RPDE_EMU
  for (long r = 0; r < rows; ++r) std::memcpy(dst + r * ld, src + r * cols, cols * sizeof(double));
#else
  RPDE_HIP(hipMemcpy2D(dst, ld * sizeof(double), src, cols * sizeof(double), cols * sizeof(double),
                       rows, hipMemcpyHostToDevice));
#endif
}
inline void dev_download2d(double* dst, const double* src, long ld, long rows, long cols) {
#ifdef RPDE_EMU
  for (long r = 0; r < rows; ++r) std::memcpy(dst + r * cols, src + r * ld, cols * sizeof(double));
#else
  RPDE_HIP(hipMemcpy2D(dst, cols * sizeof(double), src, ld * sizeof(double), cols * sizeof(double),
                       rows, hipMemcpyDeviceToHost));
#endif
}
inline void dev_zero(void* p, size_t bytes, Stream& st) {
#ifdef RPDE_EMU
  (void)st;
  std::memset(p, 0, bytes);
#else
  RPDE_HIP(hipMemsetAsync(p, 0, bytes, st.s));
#endif
}
inline void dev_sync(Stream& st) {
#ifdef RPDE_EMU
  (void)st;
#else
  RPDE_HIP(hipStreamSynchronize(st.s));
#endif
}

// owning device buffer of doubles
struct DBuf {
  double* p = nullptr;
  size_t n = 0;
  double* base = nullptr;   // what the allocator returned
  DBuf() = default;
  explicit DBuf(size_t count) { alloc(count); }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : p(o.p), n(o.n), base(o.base) { o.p = nullptr; o.n = 0; o.base = nullptr; }
  DBuf& operator=(DBuf&& o) noexcept {
    if (this != &o) { dev_free(base); p = o.p; n = o.n; base = o.base; o.p = nullptr; o.n = 0; o.base = nullptr; }
    return *this;
  }
  ~DBuf() { dev_free(base); }
  void alloc(size_t count) {
    dev_free(base);
    n = count;
    base = static_cast<double*>(dev_alloc(count * sizeof(double)));
    p = base;
  }
  // tables are uploaded with kUploadSlack zero doubles behind them: device code reads table
  // entries without bounds checks up to the per-thread capacity of a line kernel
  static constexpr size_t kUploadSlack = 5200;
  void upload(const std::vector<double>& h) {
    alloc(h.size() + kUploadSlack);
    dev_upload(p, h.data(), h.size() * sizeof(double));
  }
};

}  // namespace rpde
