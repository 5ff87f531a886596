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

#ifndef RPDE_EMU
// Device memory of every engine and operator of the process comes out of a few large slabs: one hipMalloc of
// kSlabBytes (a multiple of the 2 MB a page-table block covers; a larger request gets a slab of its own size) that is
// mapped ONCE, and a best-fit free list with coalescing on top of it.  Nothing this library allocates shares a
// 2 MB block of the runtime's own small-allocation heap (ROCr packs hipMalloc requests below 2 MB into 2 MB blocks it
// also uses for its internal objects, and gives such blocks back piecemeal), no mapping appears or disappears while a
// step runs, and an engine's 300 tables cost 300 free-list operations instead of 300 driver calls.  Slabs go back to
// the driver only when an allocation fails (trim()) -- DESIGN.md section 10-0 has the measurements that led here.
class DevArena {
 public:
  static constexpr size_t kPage = 4096;
  static constexpr size_t kHuge = size_t(2) << 20;
  static constexpr size_t kSlabBytes = size_t(1) << 30;
  static DevArena& get() { static DevArena a; return a; }
  void* alloc(size_t bytes) {
    const size_t n = round_up(bytes ? bytes : 8, kPage);
    std::lock_guard<std::mutex> lk(mu_);
    void* p = take(n);
    if (!p) {
      if (!add_slab(n)) { trim(); if (!add_slab(n)) return nullptr; }
      p = take(n);
    }
    return p;
  }
  void free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu_);
    auto u = used_.find(static_cast<char*>(p));
    if (u == used_.end()) return;
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
  size_t slab_bytes() { std::lock_guard<std::mutex> lk(mu_); size_t t = 0; for (auto& s : slabs_) t += s.size; return t; }
  size_t used_bytes() { std::lock_guard<std::mutex> lk(mu_); size_t t = 0; for (auto& u : used_) t += u.second.bytes; return t; }
  template <class F> void for_each_used(F f) { std::lock_guard<std::mutex> lk(mu_); for (auto& u : used_) f(u.first, u.second.bytes); }

 private:
  struct Slab { char* base = nullptr; size_t size = 0; std::map<char*, size_t> free; };
  struct Used { size_t bytes; size_t slab; };
  static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
  void* take(size_t n) {                                      // best fit over all slabs
    size_t bs = 0; std::map<char*, size_t>::iterator bi; bool found = false;
    for (size_t k = 0; k < slabs_.size(); ++k)
      for (auto it = slabs_[k].free.begin(); it != slabs_[k].free.end(); ++it)
        if (it->second >= n && (!found || it->second < bi->second)) { bs = k; bi = it; found = true; }
    if (!found) return nullptr;
    char* a = bi->first; const size_t rest = bi->second - n;
    slabs_[bs].free.erase(bi);
    if (rest) slabs_[bs].free[a + n] = rest;
    used_[a] = Used{n, bs};
    return a;
  }
  bool add_slab(size_t n) {
    const size_t size = std::max(kSlabBytes, round_up(n, kHuge));
    void* p = nullptr;
    if (hipMalloc(&p, size) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (auto& s : slabs_) if (!s.base) { s.base = static_cast<char*>(p); s.size = size; s.free.clear(); s.free[s.base] = size; return true; }
    slabs_.emplace_back();
    slabs_.back().base = static_cast<char*>(p); slabs_.back().size = size; slabs_.back().free[slabs_.back().base] = size;
    return true;
  }
  void trim() {                                               // slabs nobody uses go back to the driver
    for (auto& s : slabs_)
      if (s.base && s.free.size() == 1 && s.free.begin()->second == s.size) { (void)hipFree(s.base); s.base = nullptr; s.size = 0; s.free.clear(); }
  }
  std::mutex mu_;
  std::vector<Slab> slabs_;
  std::map<char*, Used> used_;
};
inline bool dev_arena_on() {   // A/B only (tools/archive/fault_hunt_r04c.sh): RPDE_ARENA=0 = one hipMalloc per buffer, as rounds 1-3
  static const bool on = [] { const char* e = std::getenv("RPDE_ARENA"); return !e || std::atoi(e) != 0; }();
  return on;
}
#endif

inline void* dev_alloc(size_t bytes) {
#ifdef RPDE_EMU
  // a NaN-filled guard in front of every buffer: a read below the start of a table or array
  // (a page fault on the device when the buffer opens an allocation) poisons the results here
  char* raw = static_cast<char*>(std::calloc(bytes + kEmuGuardBytes + 1, 1));
  RPDE_REQUIRE(raw, "host allocation failed");
  std::memset(raw, 0xFF, kEmuGuardBytes);
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
  if (p) std::free(static_cast<char*>(p) - kEmuGuardBytes);
#else
  if (!p) return;
  if (dev_arena_on()) DevArena::get().free(p);
  else (void)hipFree(p);
#endif
}
inline void dev_upload(void* dst, const void* src, size_t bytes) {
#ifdef RPDE_EMU
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
