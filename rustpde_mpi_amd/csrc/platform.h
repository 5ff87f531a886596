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
};
#define RPDE_PHASE(blk, tid) for (int tid = 0; tid < (blk).T; ++tid)
#define RPDE_PIN(x) ((void)0)
#define RPDE_SYNC(blk) ((void)0)
#define RPDE_MARK(blk, id) ((void)0)
#define RPDE_TLS(blk, type, name, K) std::vector<type> name##_st((size_t)(blk).T * (K)); const int name##_K = (K)
#define RPDE_T(name) (&name##_st[(size_t)tid * name##_K])
#else
#define RPDE_HD __host__ __device__
#define RPDE_DEV __device__ __forceinline__
#define RPDE_DEVN __device__ __forceinline__
struct Blk {
  int line, comp, T;
  double* lds;
  long long* trc;   // diagnostics: this workgroup's record of Program::trace (null: no tracing)
  int nm;           // marks written so far
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
#define RPDE_PHASE(blk, tid) for (int tid = rpde_tid(), _once = 1; _once; _once = 0)
// A value loaded from global memory is pinned where all loads of the phase have been issued: the compiler
// may not sink the load into the (per-lane predicated) block of its only user, where it would be followed by
// an s_waitcnt vmcnt(0) -- one dependent memory round trip per element (tools/check_load_issue.py)
#define RPDE_PIN(x) asm volatile("" : "+v"(x))
#define RPDE_SYNC(blk) do { __syncthreads(); RPDE_MARK(blk, -1); } while (0)
#define RPDE_TLS(blk, type, name, K) type name[K]
#define RPDE_T(name) name
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
  RPDE_HIP(hipMalloc(&p, bytes ? bytes : 8));
  RPDE_HIP(hipMemset(p, 0, bytes ? bytes : 8));
  // diagnostics: RPDE_LOG_ALLOC=1 prints every allocation, so that a fault address can be placed
  static const bool log = [] { const char* e = std::getenv("RPDE_LOG_ALLOC"); return e && std::atoi(e) != 0; }();
  if (log) fprintf(stderr, "[alloc] %p %zu\n", p, bytes);
  return p;
#endif
}
inline void dev_free(void* p) {
#ifdef RPDE_EMU
  if (p) std::free(static_cast<char*>(p) - kEmuGuardBytes);
#else
  static const bool keep = [] { const char* e = std::getenv("RPDE_NO_FREE"); return e && std::atoi(e) != 0; }();   // diagnostics: leak instead of freeing
  if (p && !keep) (void)hipFree(p);
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
