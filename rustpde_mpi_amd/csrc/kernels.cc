// Device kernels (gfx950) and their launchers.  With -DRPDE_EMU the same line-VM source is
// compiled for the host (tests only, see platform.h); transposes/GEMM/reductions are then plain
// loops that define the expected results of the HIP kernels.
#include "kernels.h"

#include "dct_line.h"
#include "hdct_line.h"
#include "rhs_line.h"

#include <algorithm>
#include <atomic>
#include <cmath>

namespace rpde {

// three compile-time configurations of the line kernel, selected by slot length
using CfgS = LineCfg<128, 10, 2, 1024, 8>;      // slots up to 1280 doubles
using CfgM = LineCfg<256, 10, 1024, 2048, 8>;   // slots up to 2560 doubles
using CfgL = LineCfg<512, 10, 2048, 4096, 8>;   // slots up to 5120 doubles
using CfgX = LineCfg<1024, 18, 4096, 8192, 8>;  // Fourier lines of 8192 / 16384 reals: one slot of up to 17408 doubles
using CfgXC = LineCfg<1024, 10, 8192, 8192, 8, true>;   // Chebyshev lines of 2049 .. 4096 points other than 2^k + 1: Bluestein with
                                                        // M = 8192, two slots of 8704 doubles (153 KB with the scan carries)
static_assert(CfgS::EPT % 2 == 0 && CfgS::C == CfgS::EPT, "scan chunk must equal EPT");

// ------------------------------------------------------------------------------- transposed tile copy
// out[(c0 + j) * ldo + r0 + i] = in[(r0 + i) * ldi + c0 + j] for one TS x TS tile through LDS
// (padded pitch TS + 1), 256 threads.  Shared by the device kernels and the host emulation (same
// source, RPDE_PHASE = thread loop there).  The K = TS * TS / 256 loads of a thread are issued as
// one batch into registers before the first LDS write: written as `for (i = ty; i < TS; i += RY)`
// the loop had a run-time trip count, was not unrolled, and kept a single load in flight.
struct alignas(16) Cplx { double re, im; };

template <class E, int TS>
RPDE_DEV void transpose_tile(Blk& blk, E* tile, const E* __restrict__ in, long ldi, E* __restrict__ out,
                             long ldo, int rows, int cols, int r0, int c0) {
  constexpr int RY = 256 / TS, K = TS / RY, LP = TS + 1;
  static_assert(256 % TS == 0 && TS % RY == 0, "tile shape");
  RPDE_PHASE(blk, tid) {
    const int tx = tid % TS, ty = tid / TS;
    const int c = c0 + tx;
    E v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int r = r0 + ty + k * RY;
      v[k] = (r < rows && c < cols) ? in[(long)r * ldi + c] : E{};
    }
#pragma unroll
    for (int k = 0; k < K; ++k) tile[(ty + k * RY) * LP + tx] = v[k];
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int tx = tid % TS, ty = tid / TS;
    const int r = r0 + tx;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = ty + k * RY, c = c0 + i;
      if (r < rows && c < cols) out[(long)c * ldo + r] = tile[tx * LP + i];
    }
  }
}

#ifndef RPDE_EMU
// =================================================================================== HIP build
// Three kernels per configuration (line_vm.h kVar*): light, with the second-order back-substitution
// scan, with the register stash.  All are held to 128 VGPRs (4 waves per SIMD) so that two
// workgroups of a two-slot program share a CU; none spills more than a few dwords.
// TRACE: the instrumented twin (Program::trace, tools/trace_ops.py), built for the 512-thread configuration
// only; in the product kernels the record pointer is a compile-time null and every RPDE_MARK folds away
template <class Cfg, int VAR, bool TRACE = false>
__global__ __launch_bounds__(Cfg::T, 4) void line_kernel(const Program pg) {
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];
  // XCD-aware line map: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
  // block b works on line (b % 8) * chunk + b / 8 -- every XCD sweeps one contiguous band of lines
  // and the cross-line stencil (rows j and j-2, OP_LOADX) finds its second row in the local L2
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= pg.nlines) return;
  Blk blk{line, (int)blockIdx.y, Cfg::T, rpde_lds,
          TRACE ? pg.trace + ((long)blockIdx.y * gridDim.x + blockIdx.x) * kTraceStride : nullptr, 0};
  run_line_program<Cfg, VAR>(blk, pg);
}

template <class Cfg, int VAR, bool TRACE = false>
static void launch_kernel(const Program& pg, size_t bytes, Stream& st) {
  if constexpr (!TRACE && Cfg::T == 512) {
    if (pg.trace) { launch_kernel<Cfg, VAR, true>(pg, bytes, st); return; }
  }
  RPDE_REQUIRE(TRACE || pg.trace == nullptr, "Program::trace: only the 512-thread configuration has the instrumented kernels");
  // the dynamic-LDS permission of a kernel is raised lazily, per device (several handles on
  // several devices / host threads in one process stay correct)
  static std::atomic<size_t> configured[32];
  int dev = 0;
  RPDE_HIP(hipGetDevice(&dev));
  std::atomic<size_t>& have = configured[dev & 31];
  if (bytes > have.load(std::memory_order_acquire)) {
    RPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&line_kernel<Cfg, VAR, TRACE>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have.store(bytes, std::memory_order_release);
  }
  dim3 grid(8 * ((pg.nlines + 7) / 8), pg.ncomp), block(Cfg::T);   // 8 bands of ceil(nlines / 8) lines
  hipLaunchKernelGGL((line_kernel<Cfg, VAR, TRACE>), grid, block, bytes, st.s, pg);
  RPDE_HIP(hipGetLastError());
}

template <class Cfg>
static void launch_cfg(const Program& pg, Stream& st) {
  const size_t bytes = line_lds_doubles(pg.nslots, pg.slot_len, Cfg::kMaxSlotLen, Cfg::kCarryLen) * sizeof(double);
  RPDE_REQUIRE(bytes <= 160 * 1024, "line program needs more than 160 KiB of LDS");
  bool rec2 = false, stash = false;
  for (int i = 0; i < pg.nops; ++i) {
    rec2 |= pg.ops[i].code == OP_REC2;
    stash |= pg.ops[i].code == OP_PUSH || pg.ops[i].code == OP_POPAXPY;
  }
  RPDE_REQUIRE(!(rec2 && stash), "a line program cannot combine the register stash with a banded solve");
  if constexpr (Cfg::kCheb) {
    if (rec2) { launch_kernel<Cfg, kVarRec2>(pg, bytes, st); return; }
    if (stash) { launch_kernel<Cfg, kVarStash>(pg, bytes, st); return; }
  } else {
    RPDE_REQUIRE(!rec2 && !stash, "this line configuration has the light kernel only");
  }
  launch_kernel<Cfg, kVarLight>(pg, bytes, st);
}

// ------------------------------------------------------------------------------- transpose
template <class E, int TS>
__global__ __launch_bounds__(256) void transpose_kernel(const E* __restrict__ in, long ldi,
                                                        E* __restrict__ out, long ldo, int rows,
                                                        int cols) {
  __shared__ E tile[TS * (TS + 1)];
  Blk blk{0, 0, 256, nullptr};
  transpose_tile<E, TS>(blk, tile, in, ldi, out, ldo, rows, cols, (int)blockIdx.y * TS, (int)blockIdx.x * TS);
}

// several arrays of one shape in ONE launch (blockIdx.z picks the array): the six transposes of T1 and the three of T2
template <class E, int TS>
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TransposeBatch b, long ldi, long ldo, int rows, int cols) {
  __shared__ E tile[TS * (TS + 1)];
  Blk blk{0, 0, 256, nullptr};
  transpose_tile<E, TS>(blk, tile, reinterpret_cast<const E*>(b.in[blockIdx.z]), ldi, reinterpret_cast<E*>(b.out[blockIdx.z]), ldo,
                        rows, cols, (int)blockIdx.y * TS, (int)blockIdx.x * TS);
}
void launch_transpose_batch(const TransposeBatch& b, int n, long ldi, long ldo, int rows, int cols, int elem, Stream& st) {
  if (rows <= 0 || cols <= 0 || n <= 0) return;
  RPDE_REQUIRE(n <= kMaxTransposeBatch, "transpose batch too large");
  if (elem == 1) {
    dim3 grid((cols + 63) / 64, (rows + 63) / 64, n);
    hipLaunchKernelGGL((transpose_batch_kernel<double, 64>), grid, dim3(256), 0, st.s, b, ldi, ldo, rows, cols);
  } else {
    RPDE_REQUIRE(elem == 2 && ldi % 2 == 0 && ldo % 2 == 0, "complex transpose needs even pitches");
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, n);
    hipLaunchKernelGGL((transpose_batch_kernel<Cplx, 32>), grid, dim3(256), 0, st.s, b, ldi / 2, ldo / 2, rows, cols);
  }
  RPDE_HIP(hipGetLastError());
}

void launch_transpose(const double* in, long ldi, double* out, long ldo, int rows, int cols,
                      int elem, Stream& st) {
  if (rows <= 0 || cols <= 0) return;
  if (elem == 1) {
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    hipLaunchKernelGGL((transpose_kernel<double, 64>), grid, dim3(256), 0, st.s, in, ldi, out, ldo,
                       rows, cols);
  } else {
    RPDE_REQUIRE(elem == 2 && ldi % 2 == 0 && ldo % 2 == 0, "complex transpose needs even pitches");
    dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    hipLaunchKernelGGL((transpose_kernel<Cplx, 32>), grid, dim3(256), 0, st.s,
                       reinterpret_cast<const Cplx*>(in), ldi / 2,
                       reinterpret_cast<Cplx*>(out), ldo / 2, rows, cols);
  }
  RPDE_HIP(hipGetLastError());
}

// (the f64 MFMA GEMM lives in gemm.cc)
// ------------------------------------------------------------------------------- callback diagnostics
__global__ __launch_bounds__(256) void diag_rows_kernel(const double* __restrict__ T, const double* __restrict__ dT,
                                                        const double* __restrict__ ux, const double* __restrict__ uy,
                                                        long ld, int ny, const double* __restrict__ wx,
                                                        const double* __restrict__ wy, double c_nu, double c_v1,
                                                        double c_v2, double c_re, double* __restrict__ partial) {
  __shared__ double sv[256], sr[256];
  const int i = (int)blockIdx.x;
  const long row = (long)i * ld;
  double av = 0.0, ar = 0.0;
  for (int j = (int)threadIdx.x; j < ny; j += 256) {
    const double w = wy[j], u = ux[row + j], v = uy[row + j];
    av += w * (c_v1 * dT[row + j] + c_v2 * T[row + j] * v);
    ar += w * (c_re * sqrt(u * u + v * v));
  }
  sv[threadIdx.x] = av; sr[threadIdx.x] = ar;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sv[threadIdx.x] += sv[threadIdx.x + o]; sr[threadIdx.x] += sr[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double w = wx[i];
    partial[4 * i + 0] = w * c_nu * dT[row];
    partial[4 * i + 1] = w * c_nu * dT[row + ny - 1];
    partial[4 * i + 2] = w * sv[0];
    partial[4 * i + 3] = w * sr[0];
  }
}
__global__ __launch_bounds__(256) void diag_final_kernel(const double* __restrict__ partial, int nx, double* out4) {
  __shared__ double s[4][256];
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = (int)threadIdx.x; i < nx; i += 256)
    for (int c = 0; c < 4; ++c) a[c] += partial[4 * i + c];
  for (int c = 0; c < 4; ++c) s[c][threadIdx.x] = a[c];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int c = 0; c < 4; ++c) s[c][threadIdx.x] += s[c][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 4) out4[threadIdx.x] = s[threadIdx.x][0];
}
void launch_diag_reduce(const double* T, const double* dT, const double* ux, const double* uy, long ld, int nx, int ny,
                        const double* wx, const double* wy, double c_nu, double c_v1, double c_v2, double c_re,
                        double* partial, double* out4, Stream& st) {
  hipLaunchKernelGGL(diag_rows_kernel, dim3(nx), dim3(256), 0, st.s, T, dT, ux, uy, ld, ny, wx, wy, c_nu, c_v1, c_v2, c_re, partial);
  hipLaunchKernelGGL(diag_final_kernel, dim3(1), dim3(256), 0, st.s, partial, nx, out4);
  RPDE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------- column scans (colscan.h)
template <int PASS>
__global__ __launch_bounds__(256) void col_hholtz_kernel(const ColHhArgs a) {
  int i = (int)(blockIdx.x * blockDim.x + threadIdx.x), f = (int)blockIdx.z;
  if ((PASS == 0 || PASS == 4) && a.pair) {
    // two fields, one input: workgroups go to the XCDs round robin (grid.x is a multiple of 8), so x and x + 8 share an L2 and
    // run back to back -- they take the same column tile, one field each
    const int q = (int)blockIdx.x >> 3;
    f = q & 1;
    i = (((q >> 1) << 3) + ((int)blockIdx.x & 7)) * (int)blockDim.x + (int)threadIdx.x;
  }
  if (i >= a.ncols) return;
  if constexpr (PASS == 0) colhh_block<false>(a, f, (int)blockIdx.y, i);
  if constexpr (PASS == 1) colhh_carry<0>(a, f, i, (int)blockIdx.y);
  if constexpr (PASS == 2) colhh_carry<1>(a, f, i, (int)blockIdx.y);
  if constexpr (PASS == 3) colhh_carry<2>(a, f, i, (int)blockIdx.y);
  if constexpr (PASS == 4) colhh_block<true>(a, f, (int)blockIdx.y, i);
}
void launch_col_hholtz_phase(const ColHhArgs& a, int phase, Stream& st) {
  if (a.ncols <= 0 || a.n <= 0 || a.nf <= 0) return;
  // the carry kernels are serial chains over the blocks, one thread per (column, parity): small workgroups spread
  // the few thousand threads over all CUs
  constexpr int ct = 64;   // (128 / 256 threads per workgroup measured 3 % slower: DESIGN.md section 8)
  const int tiles = (a.ncols + 255) / 256;
  const dim3 blk(256), gb(a.pair ? 16 * ((tiles + 7) / 8) : tiles, std::max(a.NB, 1), a.pair ? 1 : a.nf), cblk(ct), gc((a.ncols + ct - 1) / ct, 2, a.nf);
  if (phase == 0) { if (a.NB > 0) hipLaunchKernelGGL(col_hholtz_kernel<0>, gb, blk, 0, st.s, a); }
  else if (phase == 1) {
    if (a.nranks <= 1) hipLaunchKernelGGL(col_hholtz_kernel<1>, gc, cblk, 0, st.s, a);
    else hipLaunchKernelGGL(col_hholtz_kernel<2>, gc, cblk, 0, st.s, a);
  } else if (phase == 2) hipLaunchKernelGGL(col_hholtz_kernel<3>, gc, cblk, 0, st.s, a);
  else if (a.NB > 0) hipLaunchKernelGGL(col_hholtz_kernel<4>, gb, blk, 0, st.s, a);
  RPDE_HIP(hipGetLastError());
  static const bool sync_each = [] { const char* e = std::getenv("RPDE_SYNC_LAUNCHES"); return e && std::atoi(e) > 1; }();   // diagnostics
  if (sync_each) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st.s, &cs);
    if (cs == hipStreamCaptureStatusNone) { (void)hipStreamSynchronize(st.s); fprintf(stderr, " [phase %d ok]", phase); fflush(stderr); }
  }
}
// single-pass column scan (colscan1.h): W waves = W blocks of 64 columns per workgroup.  Everything a wave indexes tables
// with (field, tile, super-block, its block) is made wave-uniform with readfirstlane: the row coefficients are scalar loads.
// The aggregates travel between the workgroups as agent-scope relaxed atomics (write-through stores, loads that do not
// hit stale lines of this XCD's L2): an acquire / release fence at agent scope is a write-back or an invalidation of the
// WHOLE L2 on this chip -- in a polling loop that takes the L2 away from every other workgroup of the XCD.
template <int W, bool TRACE = false>
__global__ __launch_bounds__(W * 64, 4) void col_hholtz1_kernel(const ColHh1Args A) {   // four waves per SIMD: 128 VGPRs
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];
  double* loc = rpde_lds;                                // [W][7][64] block states (colhh1_chain works in place)
  double* stg = loc + W * kCol1Agg * kCol1Tile;          // [NSB][6][64] aggregates of the tile's super-blocks
  double* kapl = stg + A.NSB * kCol1Stg * kCol1Tile;     // [64]
  double* tbl = kapl + kCol1Tile;                        // [W][14] transfers of this super-block's blocks
  double* twl = tbl + W * kCol1TabPerBlock;              // [NSB][14] transfers of the tile's super-blocks
  __shared__ int tk, tke[2];
  const ColHhArgs& a = A.a;
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long* trc = TRACE ? A.trace + (long)blockIdx.x * kTraceStride : nullptr;
  int nm = 0;
#define RPDE_C1_MARK(id) do { if (TRACE && tid == 0 && nm < kTraceMarks) { trc[4 + 2 * nm] = (id); trc[5 + 2 * nm] = (long long)clock64(); ++nm; } } while (0)
  if (TRACE && tid == 0) trc[0] = (long long)wall_clock64();
  RPDE_C1_MARK(0);
  if (tid == 0) {   // ticket and epoch of this workgroup (the counters of a launch site are never reset)
    const unsigned long long t = atomicAdd(&A.sync[0], 1ull), ep = t / gridDim.x + 1ull;
    tk = (int)(t - (ep - 1ull) * gridDim.x);
    tke[0] = (int)(unsigned)ep; tke[1] = (int)(unsigned)(ep >> 32);
  }
  __syncthreads();
  RPDE_C1_MARK(1);
  const int ticket = __builtin_amdgcn_readfirstlane(tk);
  const unsigned long long epoch = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(tke[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(tke[0]);
  // ticket -> (field, tile, super-block): the super-blocks of a tile are consecutive; with `pair` the two fields that read
  // the same rows sit eight tickets apart (workgroups go to the XCDs round robin: same L2)
  // (integer division runs on the vector unit: every quotient goes back into a scalar register BEFORE anything branches
  // on it, so that the branches stay scalar and field / tile / super-block stay wave-uniform for the compiler)
  const int per = A.NSB * A.tiles;                      // (tile, super-block) combinations of one field
  int f, c;
  if (a.pair && a.nf == 2) { const int ch = ticket >> 4, r = ticket & 15; f = r >> 3; c = ch * 8 + (r & 7); }
  else { f = __builtin_amdgcn_readfirstlane(ticket / per); c = ticket - f * per; }
  if (c >= per || f >= a.nf) return;                    // padding tickets of the paired order
  const int tile = __builtin_amdgcn_readfirstlane(c / A.NSB), q = __builtin_amdgcn_readfirstlane(c - tile * A.NSB);
  const ColHhTabs& t = a.tab[f];
  const ColHh1Tabs& x = A.x[f];
  const int b = __builtin_amdgcn_readfirstlane(q * W + w), i = tile * kCol1Tile + lane;
  // the transfers the chains and the sweep multiply with: one vector load per thread, issued in front of the row loads and
  // put into LDS behind the zero-inflow solve (a scalar load inside a chain step is a cold miss of the scalar cache in front
  // of every step)
  double tv = 0.0;
  const bool tfetch = tid < (W + A.NSB) * kCol1TabPerBlock;
  if (tfetch) {
    const int e = tid - W * kCol1TabPerBlock;
    const double* p = (e < 0) ? colhh1_block_tab(t, q * W + tid / kCol1TabPerBlock, a.NB, tid % kCol1TabPerBlock)
                              : colhh1_super_tab(x, e / kCol1TabPerBlock, e % kCol1TabPerBlock);
    tv = p ? *(const __attribute__((address_space(1))) double*)p : colhh1_ident_tab(tid % kCol1TabPerBlock);
  }
  // a wave whose block exists runs all 64 lanes (lanes behind the last column take the last column's data and store
  // nothing): the lanes carry the row coefficients for each other (Col1TabLanes)
  const bool rowsok = b < a.NB, store = rowsok && i < a.ncols;
  const int ic = (i < a.ncols) ? i : a.ncols - 1;
  double r[kColBR + 4];
  ColLoc L;
#pragma unroll
  for (int k = 0; k < kCol1Agg; ++k) L.v[k] = 0.0;
  // a wave whose block does not exist (super-blocks behind the last block) fetches block 0's coefficients and never uses
  // them: the tables are padded for one block behind the last row, not for W of them
  const int bt = rowsok ? b : 0;
  const Col1TabLanes tabs(t, x, bt * kColBR, bt * kColBR - a.shift[f], lane);
  if (rowsok) colhh1_local(a, f, b, ic, r, L, tabs);
  if (tfetch) tbl[tid] = tv;                             // (twl follows tbl)
  RPDE_C1_MARK(2);
#pragma unroll
  for (int k = 0; k < kCol1Agg; ++k) loc[(w * kCol1Agg + k) * kCol1Tile + lane] = L.v[k];
  __syncthreads();
  RPDE_C1_MARK(3);
  double* ag = A.agg + col1_agg(A, f, tile, 0);
  double* mine = ag + (long)q * (kCol1Agg * kCol1Tile);
  if (w < 2) {                                           // wave p: the chains of parity p from zero inflow = this super-block's aggregate
    double so, T0, T1;
    colhh1_chain<true>(loc, tbl, W, w, lane, 0.0, 0.0, 0.0, so, T0, T1);
    __hip_atomic_store(mine + w * kCol1Tile + lane, so, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + (2 + 2 * w) * kCol1Tile + lane, T0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + (3 + 2 * w) * kCol1Tile + lane, T1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (w == 2 && t.w) {
    double d = 0.0;
    for (int u = 0; u < W; ++u) d += loc[(u * kCol1Agg + 6) * kCol1Tile + lane];
    __hip_atomic_store(mine + 6 * kCol1Tile + lane, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the write-through stores have arrived
  RPDE_C1_MARK(4);
  __syncthreads();
  unsigned long long* arrivals = A.sync + 1 + f * A.tiles + tile;
  unsigned long long* ready = A.sync + 1 + kColMaxFields * A.tiles + f * A.tiles + tile;
  if (tid == 0) tk = (int)(__hip_atomic_fetch_add(arrivals, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % (unsigned long long)A.NSB);
  __syncthreads();
  const bool last = __builtin_amdgcn_readfirstlane(tk) == A.NSB - 1;
  RPDE_C1_MARK(5);
  if (TRACE && tid == 0) trc[3] = last ? 1 : 0;
  double* mystg = stg + (long)q * kCol1Stg * kCol1Tile;  // this super-block's inflow states
  if (last) {
    // the last workgroup of the tile: all aggregates into LDS (eight rows of 64 doubles per wave in flight at a time) ...
    const int rows = A.NSB * kCol1Stg;
    for (int k0 = w; k0 < rows; k0 += 8 * W) {
      double v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e * W;
        if (k < rows) { const int qq = k / kCol1Stg; v[e] = __hip_atomic_load(ag + ((long)qq * kCol1Agg + (k - qq * kCol1Stg)) * kCol1Tile + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e * W;
        if (k < rows) stg[(long)k * kCol1Tile + lane] = v[e];
      }
    }
    if (w == W - 1) {                                    // the rank-one sum of the column (fields with tab.w)
      double kp = 0.0;
      if (t.w) for (int u = 0; u < A.NSB; ++u) kp += __hip_atomic_load(ag + ((long)u * kCol1Agg + 6) * kCol1Tile + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      kapl[lane] = kp;
    }
    __syncthreads();
    if (w < 2) colhh1_sweep(stg, twl, A.NSB, w, lane);    // ... the inflow states of every super-block ...
    __syncthreads();
    for (int k = w; k < rows; k += W) {                  // ... published in place of the aggregates
      const int qq = k / kCol1Stg;
      __hip_atomic_store(ag + ((long)qq * kCol1Agg + (k - qq * kCol1Stg)) * kCol1Tile + lane, stg[(long)k * kCol1Tile + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (w == W - 1) __hip_atomic_store(ag + 6 * kCol1Tile + lane, kapl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ready, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (w == 0) {
      int it = 0;
      while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(8);
        if (++it > (1 << 22)) { if (lane == 0) *A.err = 1; break; }
      }
    }
    __syncthreads();
    if (w < kCol1Stg) mystg[w * kCol1Tile + lane] = __hip_atomic_load(ag + ((long)q * kCol1Agg + w) * kCol1Tile + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (w == kCol1Stg) kapl[lane] = __hip_atomic_load(ag + 6 * kCol1Tile + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  RPDE_C1_MARK(6);
  if (w < 2) {
    double so, T0, T1;
    colhh1_chain<false>(loc, tbl, W, w, lane, mystg[w * kCol1Tile + lane], mystg[(2 + 2 * w) * kCol1Tile + lane],
                        mystg[(3 + 2 * w) * kCol1Tile + lane], so, T0, T1);
  }
  RPDE_C1_MARK(7);
  __syncthreads();
  RPDE_C1_MARK(8);
  if (rowsok) {
    double in6[kCol1Inf];
#pragma unroll
    for (int k = 0; k < kCol1Inf; ++k) in6[k] = loc[(w * kCol1Agg + k) * kCol1Tile + lane];
    colhh1_final(a, f, b, i, r, in6, kapl[lane], tabs, store);
  }
  RPDE_C1_MARK(9);
  if (TRACE && tid == 0) { trc[1] = (long long)wall_clock64(); trc[2] = nm; }
#undef RPDE_C1_MARK
}
void launch_col_hholtz1(const ColHh1Args& A, Stream& st) {
  const ColHhArgs& a = A.a;
  if (a.ncols <= 0 || a.n <= 0 || a.nf <= 0 || a.NB <= 0) return;
  RPDE_REQUIRE(A.NSB <= kCol1MaxNSB && A.W * A.NSB >= a.NB, "colhh1: super-block partition");
  RPDE_REQUIRE(A.W >= 8 && A.sync != nullptr, "colhh1: synchronisation area of the launch site");
  RPDE_REQUIRE((A.W + A.NSB) * kCol1TabPerBlock <= 64 * A.W, "colhh1: one thread per transfer coefficient (W + NSB blocks of 14)");
  const int per = A.NSB * A.tiles;
  const int wgs = (a.pair && a.nf == 2) ? 16 * ((per + 7) / 8) : per * a.nf;
  const size_t bytes = sizeof(double) * col1_lds_doubles(A.W, A.NSB);
  auto go = [&](auto kernel, int w) {
    static std::atomic<size_t> configured[32];           // dynamic-LDS permission, per device (as launch_kernel above)
    int dev = 0;
    RPDE_HIP(hipGetDevice(&dev));
    std::atomic<size_t>& have = configured[dev & 31];
    if (bytes > have.load(std::memory_order_acquire)) {
      RPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      have.store(bytes, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(64 * w), bytes, st.s, A);
  };
  if (A.trace) { if (A.W == 16) go(col_hholtz1_kernel<16, true>, 16); else go(col_hholtz1_kernel<8, true>, 8); }
  else if (A.W == 16) go(col_hholtz1_kernel<16>, 16);
  else if (A.W == 8) go(col_hholtz1_kernel<8>, 8);
  else fail("colhh1: 8 or 16 blocks per workgroup");
  RPDE_HIP(hipGetLastError());
}
int col_hholtz1_resident_workgroups(int W, int NSB) {
  int dev = 0, cus = 0, per_cu = 0;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const size_t bytes = sizeof(double) * col1_lds_doubles(W, NSB);
  const void* k = (W == 16) ? reinterpret_cast<const void*>(col_hholtz1_kernel<16>) : reinterpret_cast<const void*>(col_hholtz1_kernel<8>);
  RPDE_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  RPDE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * W, bytes));
  return per_cu * cus;
}
// single-pass y-derivative (colscan1.h)
__global__ __launch_bounds__(kDiff1W * 64, 8) void col_diff1_kernel(const ColDiff1Args A) {   // eight waves per SIMD: 64 VGPRs
  __shared__ double lt[kDiff1W][2][kCol1Tile];          // the blocks' sums
  __shared__ double sup[2][kCol1Tile];                  // the sums of the super-blocks above
  __shared__ int tk, tke[2];
  const ColDiffArgs& a = A.a;
  const int tid = (int)threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {   // ticket and epoch of this workgroup (the counters of a launch site are never reset)
    const unsigned long long t = atomicAdd(&A.sync[0], 1ull), ep = t / gridDim.x + 1ull;
    tk = (int)(t - (ep - 1ull) * gridDim.x);
    tke[0] = (int)(unsigned)ep; tke[1] = (int)(unsigned)(ep >> 32);
  }
  __syncthreads();
  const int ticket = __builtin_amdgcn_readfirstlane(tk);
  const unsigned long long epoch = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(tke[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(tke[0]);
  const int tile = __builtin_amdgcn_readfirstlane(ticket / A.NSB);
  const int q = __builtin_amdgcn_readfirstlane(A.NSB - 1 - (ticket - tile * A.NSB));   // from the top
  if (tile >= A.tiles) return;
  const int b = q * kDiff1W + w, j0 = b * kDiff1BR, i = tile * kCol1Tile + lane;
  const bool rowsok = j0 < a.nout, store = rowsok && i < a.ncols;
  const int ic = (i < a.ncols) ? i : a.ncols - 1;
  // the stencil coefficients of the block's rows, one per lane (row u: lane u), read back with v_readlane
  double lowv = 0.0;
  if (a.low && lane < kDiff1BR) lowv = ((const __attribute__((address_space(1))) double*)a.low)[(j0 - 1 + lane > 0) ? j0 - 1 + lane : 0];
  double d[kDiff1BR], tot[2] = {0.0, 0.0};
  if (rowsok) coldiff1_local(a, j0, ic, d, tot, [&](int u) { return col1_lane(lowv, u); });
  lt[w][0][lane] = tot[0]; lt[w][1][lane] = tot[1];
  __syncthreads();
  double in[2] = {0.0, 0.0};
  for (int u = kDiff1W - 1; u > w; --u) { in[0] += lt[u][0][lane]; in[1] += lt[u][1][lane]; }
  if (w == 0) {
    double* mine = A.tot + coldiff1_tot(A, tile, q);
    __hip_atomic_store(mine + lane, in[0] + tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + kCol1Tile + lane, in[1] + tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* flags = A.sync + 1 + tile * A.NSB;
    if (lane == 0) __hip_atomic_store(flags + q, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // lane l waits for super-block q + 1 + l
    const int qw = q + 1 + lane;
    int it = 0;
    while (__builtin_amdgcn_ballot_w64(qw < A.NSB && __hip_atomic_load(flags + (qw < A.NSB ? qw : q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) != 0) {
      __builtin_amdgcn_s_sleep(4);
      if (++it > (1 << 22)) { if (lane == 0) *A.err = 1; break; }
    }
    double s0 = 0.0, s1 = 0.0;
    for (int u = q + 1; u < A.NSB; ++u) {
      const double* th = A.tot + coldiff1_tot(A, tile, u);
      s0 += __hip_atomic_load(th + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s1 += __hip_atomic_load(th + kCol1Tile + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sup[0][lane] = s0; sup[1][lane] = s1;
  }
  __syncthreads();
  in[0] += sup[0][lane]; in[1] += sup[1][lane];
  if (store) coldiff1_store(a, j0, i, d, in);
}
void launch_col_diff1(const ColDiff1Args& A, Stream& st) {
  const ColDiffArgs& a = A.a;
  if (a.ncols <= 0 || a.nout <= 0) return;
  RPDE_REQUIRE(A.NSB * kDiff1Rows >= a.nout && A.NSB <= 64 && a.row0 == 0 && a.nranks <= 1, "coldiff1: one rank, at most 64 super-blocks");
  RPDE_REQUIRE(A.sync != nullptr, "coldiff1: synchronisation area of the launch site");
  hipLaunchKernelGGL(col_diff1_kernel, dim3(A.tiles * A.NSB), dim3(kDiff1W * 64), 0, st.s, A);
  RPDE_HIP(hipGetLastError());
}
template <int PASS>
__global__ __launch_bounds__(256) void col_diff_kernel(const ColDiffArgs a) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= a.ncols) return;
  if constexpr (PASS == 0) coldiff_pass<false>(a, (int)blockIdx.y, i);
  if constexpr (PASS == 1) coldiff_carry<0>(a, i, (int)blockIdx.y);
  if constexpr (PASS == 2) coldiff_carry<1>(a, i, (int)blockIdx.y);
  if constexpr (PASS == 3) coldiff_carry<2>(a, i, (int)blockIdx.y);
  if constexpr (PASS == 4) coldiff_pass<true>(a, (int)blockIdx.y, i);
}
void launch_col_diff_phase(const ColDiffArgs& a, int phase, Stream& st) {
  if (a.ncols <= 0 || a.nout <= 0) return;
  constexpr int ct = 64;   // (128 / 256 threads per workgroup measured 3 % slower: DESIGN.md section 8)
  const dim3 blk(256), gb((a.ncols + 255) / 256, std::max(a.NB, 1)), cblk(ct), gc((a.ncols + ct - 1) / ct, 2);
  if (phase == 0) { if (a.NB > 0) hipLaunchKernelGGL(col_diff_kernel<0>, gb, blk, 0, st.s, a); }
  else if (phase == 1) {
    if (a.nranks <= 1) hipLaunchKernelGGL(col_diff_kernel<1>, gc, cblk, 0, st.s, a);
    else hipLaunchKernelGGL(col_diff_kernel<2>, gc, cblk, 0, st.s, a);
  } else if (phase == 2) hipLaunchKernelGGL(col_diff_kernel<3>, gc, cblk, 0, st.s, a);
  else if (a.NB > 0) hipLaunchKernelGGL(col_diff_kernel<4>, gb, blk, 0, st.s, a);
  RPDE_HIP(hipGetLastError());
}

// TRACE: instrumented twins (Navier2DEngine::trace_launch): thread 0 leaves a clock value behind every barrier
#define RPDE_TRACE_BEGIN(trace) \
  Blk blk{line, 0, N / 16, buf, TRACE ? (trace) + (long)blockIdx.x * kTraceStride : nullptr, 0}; \
  if (TRACE && threadIdx.x == 0) { blk.trc[0] = (long long)wall_clock64(); } \
  RPDE_MARK(blk, 0)
#define RPDE_TRACE_END() \
  do { RPDE_MARK(blk, 1); if (TRACE && threadIdx.x == 0) { blk.trc[1] = (long long)wall_clock64(); blk.trc[2] = blk.nm; } } while (0)
// hdct_line.h: the same transforms through the half-length FFT (8 complex points per thread)
template <int N, bool TRACE = false>
__global__ __launch_bounds__(N / 16, 4) void hdct_line_kernel(const DctLineArgs a, long long* trace) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.nlines) return;
  RPDE_TRACE_BEGIN(trace);
  hdct_bwd_line_by_mode<N>(blk, a);
  RPDE_TRACE_END();
}
template <int N, int WPC = 4>
__global__ __launch_bounds__(N / 16, WPC) void hdct_line2_kernel(const DctLineArgs a0, const DctLineArgs a1) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a0.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
  hdct_bwd_line<N>(blk, a0);
  __syncthreads();
  hdct_bwd_line<N>(blk, a1);
}
// S1 with the line loaded once: two halves of a workgroup of N / 8 threads, one transform each (hdct_pair_line)
template <int N>
__global__ __launch_bounds__(N / 8, 4) void hdct_pair_kernel(const DctLineArgs a0, const DctLineArgs a1) {
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a0.nlines) return;
  hdct_pair_line<N>(line, rpde_lds, a0, a1);
}
// rfft_line.h: S1 and S3 of the periodic step
template <int N>
__global__ __launch_bounds__(N / 8, 4) void rfft_pair_kernel(const RfftLineArgs a0, const RfftLineArgs a1) {
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a0.nlines) return;
  rfft_pair_line<N>(line, rpde_lds, a0, a1);
}
template <int N>
__global__ __launch_bounds__(N / 16, 4) void four_rhs_kernel(const FourRhsArgs a) {
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];   // hdct_lds_doubles(N) (140 KB at N = 16384)
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.f.nlines) return;
  Blk blk{line, 0, N / 16, rpde_lds, nullptr, 0};
  four_rhs_line<N>(blk, a);
}
template <int N>
__global__ __launch_bounds__(N / 16, 4) void rfft_seq2_kernel(const RfftLineArgs a0, const RfftLineArgs a1) {
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a0.nlines) return;
  Blk blk{line, 0, N / 16, rpde_lds, nullptr, 0};
  rfft_seq2_line<N>(blk, a0, a1);
}
// dynamic LDS above 64 KB needs the permission once per kernel and device
template <class K>
static void lds_permission(K kernel, size_t bytes) {
  static std::atomic<size_t> have[32];
  int dev = 0;
  RPDE_HIP(hipGetDevice(&dev));
  if (bytes > have[dev & 31].load(std::memory_order_acquire)) {
    RPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have[dev & 31].store(bytes, std::memory_order_release);
  }
}
#ifndef RPDE_HCONV_WPC
#define RPDE_HCONV_WPC 1               // waves per SIMD of the convection term on the half-length core (1025- / 2049-point lines; A/B builds: 2, 3)
#endif
template <int N, int WPC, int MEAN = 0>   // MEAN: the linearised term of Navier2DLnse (ConvLineArgs::um, vm)
__global__ __launch_bounds__(N / 16, WPC) void hconv_line_kernel(const ConvLineArgs c) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= c.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
  hconv_line<N, MEAN>(blk, c);
}
template <int N, int WHICH, bool TRACE = false, int WPC = 3>
__global__ __launch_bounds__(N / 16, WPC) void rhs_line_kernel(const RhsLineArgs a, long long* trace) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.nlines) return;
  RPDE_TRACE_BEGIN(trace);
  rhs_line<N, WHICH>(blk, a);
  RPDE_TRACE_END();
}
// ---- batched forms for 1025-point lines (kernels.h LineBatch): blockIdx.y picks the field
struct Dct1Batch { DctLineArgs a[kLineBatch]; };
struct Dct2Batch { DctLineArgs a0[kLineBatch], a1[kLineBatch]; };
struct ConvBatch { ConvLineArgs c[kLineBatch]; };
struct RhsBatch { RhsLineArgs r[kLineBatch]; };
#define RPDE_BATCH_LINE(nl) \
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS]; \
  const int chunk = (int)gridDim.x >> 3; \
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3); \
  if (line >= (nl)) return; \
  Blk blk{line, 0, N / 16, buf, nullptr, 0}
template <int N>
__global__ __launch_bounds__(N / 16, 4) void hdct_line_batch_kernel(const Dct1Batch b) {
  const DctLineArgs& a = b.a[blockIdx.y];
  RPDE_BATCH_LINE(a.nlines);
  hdct_bwd_line_by_mode<N>(blk, a);
}
template <int N>
__global__ __launch_bounds__(N / 16, 4) void hdct_line2_batch_kernel(const Dct2Batch b) {
  const DctLineArgs& a0 = b.a0[blockIdx.y];
  const DctLineArgs& a1 = b.a1[blockIdx.y];
  RPDE_BATCH_LINE(a0.nlines);
  hdct_bwd_line<N>(blk, a0);
  __syncthreads();
  hdct_bwd_line<N>(blk, a1);
}
template <int N>
__global__ __launch_bounds__(N / 8, 4) void hdct_pair_batch_kernel(const Dct2Batch b) {
  extern __shared__ __attribute__((aligned(16))) double rpde_lds[];
  const DctLineArgs& a0 = b.a0[blockIdx.y];
  const DctLineArgs& a1 = b.a1[blockIdx.y];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a0.nlines) return;
  hdct_pair_line<N>(line, rpde_lds, a0, a1);
}
template <int N, int WPC, int MEAN = 0>
__global__ __launch_bounds__(N / 16, WPC) void hconv_line_batch_kernel(const ConvBatch b) {
  const ConvLineArgs& c = b.c[blockIdx.y];
  RPDE_BATCH_LINE(c.nlines);
  hconv_line<N, MEAN>(blk, c);
}
template <int N>
__global__ __launch_bounds__(N / 16, 3) void rhs_line_batch_kernel(const RhsBatch b) {
  const RhsLineArgs& a = b.r[blockIdx.y];
  RPDE_BATCH_LINE(a.nlines);
  if (a.which == 0) rhs_line<N, 0>(blk, a);
  else if (a.which == 1) rhs_line<N, 1>(blk, a);
  else rhs_line<N, 2>(blk, a);
}
// A/B switch: RPDE_S1_PAIR=0 runs the two transforms of S1 one after the other in one workgroup of N / 16 threads
static bool s1_pair_on() {
  static const bool on = [] { const char* e = std::getenv("RPDE_S1_PAIR"); return !e || std::atoi(e) != 0; }();
  return on;
}
// RPDE_S1_SPLIT=1 (A/B): the two transforms of S1 as two launches of the one-transform kernel -- the line is read twice, but four
// lines are resident per CU instead of the pair kernel's two
static bool s1_split_on() {
  static const bool on = [] { const char* e = std::getenv("RPDE_S1_SPLIT"); return e && std::atoi(e) != 0; }();
  return on;
}
// lines of 4097 points (round 5): the convection term of the full-length core, one field per blockIdx.y like the others
#ifndef RPDE_CONV_WPC
#define RPDE_CONV_WPC 3                // workgroups per CU of the 4097-point convection term (A/B builds: 4 = 128 registers, 8 spilled)
#endif
template <int N, int MEAN = 0>
__global__ __launch_bounds__(N / 16, RPDE_CONV_WPC) void conv_line_batch_kernel(const ConvBatch b) {
  const ConvLineArgs& c = b.c[blockIdx.y];
#ifdef RPDE_CONV4096_HALF              // (A/B build: the 4097-point term on the half-length core -- 155 registers with the guarded loads)
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
#else
  __shared__ __attribute__((aligned(16))) double buf[DctGeom<N>::LDS];
#endif
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= c.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
#ifdef RPDE_CONV4096_HALF
  hconv_line<N, MEAN>(blk, c);
#else
  conv_line<N, MEAN>(blk, c);
#endif
}
bool line_batch_ok(int N) { return N == 1024 || N == 4096; }
// N = 1024: one wave per line, a launch of one field's lines is over after one line's latency.  N = 4096: the
// chip is full either way; what the common launch saves is the drain of one field's last workgroups before the next field's
// first ones may start (the fields of a stage are independent: blockIdx.y = field, dispatched in order).
template <int N>
static void launch_line_batch_n(const LineBatch& b, int nl, Stream& st) {
  const dim3 grid(8 * ((nl + 7) / 8), b.n), block(N / 16);
  if (b.kind == 0) {
    Dct1Batch k; for (int i = 0; i < b.n; ++i) k.a[i] = b.d0[i];
    hipLaunchKernelGGL(hdct_line_batch_kernel<N>, grid, block, 0, st.s, k);
  } else if (b.kind == 1) {
    if (s1_split_on()) {   // the A/B form has no batched twin: one field after the other
      for (int i = 0; i < b.n; ++i) RPDE_REQUIRE(launch_dct_line2(b.d0[i], b.d1[i], st), "line batch: shape");
      return;
    }
    Dct2Batch k; for (int i = 0; i < b.n; ++i) { k.a0[i] = b.d0[i]; k.a1[i] = b.d1[i]; }
    bool pair = s1_pair_on();
    for (int i = 0; i < b.n; ++i) pair = pair && hdct_pair_ok(b.d0[i], b.d1[i]);
    if (pair) {
      const size_t bytes = 2 * sizeof(double) * hdct_lds_doubles(N);
      if (bytes > 65536) lds_permission(hdct_pair_batch_kernel<N>, bytes);
      hipLaunchKernelGGL(hdct_pair_batch_kernel<N>, grid, dim3(N / 8), bytes, st.s, k);
    } else hipLaunchKernelGGL(hdct_line2_batch_kernel<N>, grid, block, 0, st.s, k);
  } else if (b.kind == 2) {
    ConvBatch k; for (int i = 0; i < b.n; ++i) k.c[i] = b.c[i];
    // N = 1024: one wave per SIMD (416 VGPRs): budgets of two / three waves spill 159 / 274 registers and measured 0.110 / 0.156 ms
    // against 0.080 ms at 1025^2 (profiles/r04_experiments, call 10); N = 4096: the full-length core like conv_line_kernel
    bool mean = b.c[0].um != nullptr;
    for (int i = 0; i < b.n; ++i) RPDE_REQUIRE((b.c[i].um != nullptr) == mean && (!mean || (b.c[i].vm && b.c[i].bx && b.c[i].by)), "line batch: convection terms of one kind");
    const bool adj = mean && b.c[0].tp != nullptr;
    for (int i = 0; i < b.n; ++i) RPDE_REQUIRE(!mean || ((b.c[i].tp != nullptr) == adj && (!adj || b.c[i].cz)), "line batch: convection terms of one kind");
    if (adj) {
      if constexpr (N == 1024) hipLaunchKernelGGL((hconv_line_batch_kernel<N, RPDE_HCONV_WPC, 3>), grid, block, 0, st.s, k);
      else hipLaunchKernelGGL((conv_line_batch_kernel<N, 3>), grid, block, 0, st.s, k);
      RPDE_HIP(hipGetLastError());
      return;
    }
    const bool nl = mean && b.c[0].nonlin != 0;
    for (int i = 0; i < b.n; ++i) RPDE_REQUIRE(!mean || (b.c[i].nonlin != 0) == nl, "line batch: convection terms of one kind");
    if constexpr (N == 1024) {
      if (nl) hipLaunchKernelGGL((hconv_line_batch_kernel<N, RPDE_HCONV_WPC, 2>), grid, block, 0, st.s, k);
      else if (mean) hipLaunchKernelGGL((hconv_line_batch_kernel<N, RPDE_HCONV_WPC, 1>), grid, block, 0, st.s, k);
      else hipLaunchKernelGGL((hconv_line_batch_kernel<N, RPDE_HCONV_WPC>), grid, block, 0, st.s, k);
    } else {
      if (nl) hipLaunchKernelGGL((conv_line_batch_kernel<N, 2>), grid, block, 0, st.s, k);
      else if (mean) hipLaunchKernelGGL((conv_line_batch_kernel<N, 1>), grid, block, 0, st.s, k);
      else hipLaunchKernelGGL(conv_line_batch_kernel<N>, grid, block, 0, st.s, k);
    }
  } else {
    RhsBatch k; for (int i = 0; i < b.n; ++i) k.r[i] = b.r[i];
    hipLaunchKernelGGL(rhs_line_batch_kernel<N>, grid, block, 0, st.s, k);
  }
  RPDE_HIP(hipGetLastError());
}
void launch_line_batch(const LineBatch& b, Stream& st) {
  RPDE_REQUIRE(b.n >= 1 && b.n <= kLineBatch, "line batch: 1 .. 3 fields");
  int nl = 0, N = 0;
  for (int i = 0; i < b.n; ++i) {
    const int li = b.kind == 0 ? b.d0[i].nlines : b.kind == 1 ? b.d0[i].nlines : b.kind == 2 ? b.c[i].nlines : b.r[i].nlines;
    const int Ni = b.kind <= 1 ? b.d0[i].N : b.kind == 2 ? b.c[i].N : b.r[i].N;
    RPDE_REQUIRE(line_batch_ok(Ni) && (i == 0 || Ni == N), "line batch: lines of 1025 or of 4097 points, one length per batch");
    N = Ni;
    nl = std::max(nl, li);
  }
  if (nl <= 0) return;
  if (N == 1024) launch_line_batch_n<1024>(b, nl, st);
  else launch_line_batch_n<4096>(b, nl, st);
}
// waves per SIMD the register budget of S5 / S8 is set for (4: 128 VGPRs, four lines per CU, 8 / 7 spilled registers; 3: 168,
// three lines, no scratch) -- compile-time switches for A/B builds (python -m rustpde_mpi_amd.build <variant> -DRPDE_S5_WPC=4).
// Measured in one call of round 6 (profiles/r06_experiments/call6_ab_wpc.txt): S5 0.141 -> 0.132 ms, S8 0.268 -> 0.265 ms with 3.
#ifndef RPDE_S5_WPC
#define RPDE_S5_WPC 3
#endif
#ifndef RPDE_S8_WPC
#define RPDE_S8_WPC 3
#endif
template <int N>
__global__ __launch_bounds__(N / 16, N == 4096 ? RPDE_S5_WPC : 4) void div_line_kernel(const DivLineArgs a) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
  div_line<N>(blk, a);
}
bool launch_div_line(const DivLineArgs& a, Stream& st) {
  if ((a.N != 4096 && a.N != 1024) || !div_line_ok(a)) return false;
  if (a.nlines <= 0) return true;
  const dim3 grid(8 * ((a.nlines + 7) / 8)), block(a.N / 16);
  if (a.N == 1024) hipLaunchKernelGGL(div_line_kernel<1024>, grid, block, 0, st.s, a);
  else hipLaunchKernelGGL(div_line_kernel<4096>, grid, block, 0, st.s, a);
  RPDE_HIP(hipGetLastError());
  return true;
}
template <int N>
__global__ __launch_bounds__(N / 16, N == 4096 ? RPDE_S8_WPC : 4) void corr_line_kernel(const CorrLineArgs a) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
  corr_line<N>(blk, a);
}
bool launch_corr_line(const CorrLineArgs& a, Stream& st) {
  if ((a.N != 4096 && a.N != 1024) || !corr_line_ok(a)) return false;
  if (a.nlines <= 0) return true;
  const dim3 grid(8 * ((a.nlines + 7) / 8)), block(a.N / 16);
  if (a.N == 1024) hipLaunchKernelGGL(corr_line_kernel<1024>, grid, block, 0, st.s, a);
  else hipLaunchKernelGGL(corr_line_kernel<4096>, grid, block, 0, st.s, a);
  RPDE_HIP(hipGetLastError());
  return true;
}
template <int N, bool KEEP, bool DER = false>
__global__ __launch_bounds__(N / 16, N <= 2048 ? 2 : (KEEP ? 3 : 4)) void prow_line_kernel(const ProwLineArgs a) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
  prow_line<N, KEEP, DER>(blk, a);
}
bool launch_prow_line(const ProwLineArgs& a, Stream& st) {
  if ((a.N != 4096 && a.N != 2048 && a.N != 1024) || !prow_line_ok(a)) return false;
  if (a.nlines <= 0) return true;
  const dim3 grid(8 * ((a.nlines + 7) / 8)), block(a.N / 16);
  if (a.derive) {   // one factor row per line, the others derived from it (prow_line.h DERIVE)
    if (a.N == 1024) hipLaunchKernelGGL((prow_line_kernel<1024, true, true>), grid, block, 0, st.s, a);
    else if (a.N == 2048) hipLaunchKernelGGL((prow_line_kernel<2048, true, true>), grid, block, 0, st.s, a);
    else hipLaunchKernelGGL((prow_line_kernel<4096, true, true>), grid, block, 0, st.s, a);
  } else if (a.N == 1024) hipLaunchKernelGGL((prow_line_kernel<1024, true>), grid, block, 0, st.s, a);   // one wave per line: registers to spare
  else if (a.N == 2048) hipLaunchKernelGGL((prow_line_kernel<2048, true>), grid, block, 0, st.s, a);   // two waves per line (2049-point y-lines: BASELINE config 5)
  else if (a.keep) hipLaunchKernelGGL((prow_line_kernel<4096, true>), grid, block, 0, st.s, a);
  else hipLaunchKernelGGL((prow_line_kernel<4096, false>), grid, block, 0, st.s, a);
  RPDE_HIP(hipGetLastError());
  return true;
}
template <int N>
__global__ __launch_bounds__(N / 16, N == 1024 ? 2 : 4) void pres_line_kernel(const PresLineArgs a) {
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= a.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
  pres_line<N>(blk, a);
}
bool launch_pres_line(const PresLineArgs& a, Stream& st) {
  if ((a.N != 4096 && a.N != 1024) || !pres_line_ok(a)) return false;
  if (a.nlines <= 0) return true;
  const dim3 grid(8 * ((a.nlines + 7) / 8)), block(a.N / 16);
  if (a.N == 1024) hipLaunchKernelGGL(pres_line_kernel<1024>, grid, block, 0, st.s, a);
  else hipLaunchKernelGGL(pres_line_kernel<4096>, grid, block, 0, st.s, a);
  RPDE_HIP(hipGetLastError());
  return true;
}
__global__ __launch_bounds__(256) void per_rows_kernel(const PerRowsArgs a) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x), line = (int)blockIdx.y;
  if (k < a.kx) per_rows_point(a, line, k);
}
void launch_per_rows(const PerRowsArgs& a, Stream& st) {
  if (a.nlines <= 0 || a.kx <= 0) return;
  hipLaunchKernelGGL(per_rows_kernel, dim3((a.kx + 255) / 256, a.nlines), dim3(256), 0, st.s, a);
  RPDE_HIP(hipGetLastError());
}
bool launch_rhs_line(const RhsLineArgs& a, Stream& st, long long* trace) {
  if ((a.N != 4096 && a.N != 1024) || !rhs_line_ok(a)) return false;
  if (a.nlines <= 0) return true;
  const dim3 grid(8 * ((a.nlines + 7) / 8)), block(a.N / 16);
  if (a.N == 1024) {   // one wave per line
    if (a.which == 0) hipLaunchKernelGGL((rhs_line_kernel<1024, 0>), grid, block, 0, st.s, a, nullptr);
    else if (a.which == 1) hipLaunchKernelGGL((rhs_line_kernel<1024, 1>), grid, block, 0, st.s, a, nullptr);
    else hipLaunchKernelGGL((rhs_line_kernel<1024, 2>), grid, block, 0, st.s, a, nullptr);
  } else if (trace) {
    if (a.which == 0) hipLaunchKernelGGL((rhs_line_kernel<4096, 0, true>), grid, block, 0, st.s, a, trace);
    else if (a.which == 1) hipLaunchKernelGGL((rhs_line_kernel<4096, 1, true>), grid, block, 0, st.s, a, trace);
    else hipLaunchKernelGGL((rhs_line_kernel<4096, 2, true>), grid, block, 0, st.s, a, trace);
  } else {   // register budget of three workgroups per CU (no spills; four per CU spilled 70 - 90 registers and was slower)
    if (a.which == 0) hipLaunchKernelGGL((rhs_line_kernel<4096, 0>), grid, block, 0, st.s, a, trace);
    else if (a.which == 1) hipLaunchKernelGGL((rhs_line_kernel<4096, 1>), grid, block, 0, st.s, a, trace);
    else hipLaunchKernelGGL((rhs_line_kernel<4096, 2>), grid, block, 0, st.s, a, trace);
  }
  RPDE_HIP(hipGetLastError());
  return true;
}
bool launch_dct_line(const DctLineArgs& a, Stream& st, long long* trace) {
  if ((a.N != 4096 && a.N != 2048 && a.N != 1024) || !dct_line_ok(a)) return false;
  if (a.nlines <= 0) return true;
  if (a.N == 1024) {   // one wave per line, the half-length core only
    hipLaunchKernelGGL((hdct_line_kernel<1024>), dim3(8 * ((a.nlines + 7) / 8)), dim3(64), 0, st.s, a, nullptr);
    RPDE_HIP(hipGetLastError());
    return true;
  }
  if (a.N == 2048) {   // two waves per line (2049-point y-lines)
    hipLaunchKernelGGL((hdct_line_kernel<2048>), dim3(8 * ((a.nlines + 7) / 8)), dim3(128), 0, st.s, a, nullptr);
    RPDE_HIP(hipGetLastError());
    return true;
  }
  if (trace) hipLaunchKernelGGL((hdct_line_kernel<4096, true>), dim3(8 * ((a.nlines + 7) / 8)), dim3(256), 0, st.s, a, trace);
  else hipLaunchKernelGGL((hdct_line_kernel<4096>), dim3(8 * ((a.nlines + 7) / 8)), dim3(256), 0, st.s, a, trace);
  RPDE_HIP(hipGetLastError());
  return true;
}
template <int N, int MEAN = 0>
__global__ __launch_bounds__(N / 16, RPDE_CONV_WPC) void conv_line_kernel(const ConvLineArgs c) {
#ifdef RPDE_CONV4096_HALF              // (A/B build: the 4097-point term on the half-length core -- 155 registers with the guarded loads)
  __shared__ __attribute__((aligned(16))) double buf[HdctGeom<N>::LDS];
#else
  __shared__ __attribute__((aligned(16))) double buf[DctGeom<N>::LDS];
#endif
  const int chunk = (int)gridDim.x >> 3;
  const int line = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  if (line >= c.nlines) return;
  Blk blk{line, 0, N / 16, buf, nullptr, 0};
#ifdef RPDE_CONV4096_HALF
  hconv_line<N, MEAN>(blk, c);
#else
  conv_line<N, MEAN>(blk, c);
#endif
}
bool launch_conv_line(const ConvLineArgs& c, Stream& st) {
  if ((c.N != 4096 && c.N != 2048 && c.N != 1024) || !conv_line_ok(c)) return false;
  const bool mean = c.um != nullptr;   // the linearised term (Navier2DLnse): needs all four factor arrays
  if (mean && !(c.vm && c.bx && c.by)) return false;
  if (c.tp && !(mean && c.cz)) return false;
  if (c.nlines <= 0) return true;
  const dim3 grid(8 * ((c.nlines + 7) / 8));
  if (c.N == 2048) {   // two waves per line on the half-length core, one wave per SIMD (two lines per CU, like the line program, without its phases)
    if (mean && c.tp) hipLaunchKernelGGL((hconv_line_kernel<2048, RPDE_HCONV_WPC, 3>), grid, dim3(128), 0, st.s, c);
    else if (mean && c.nonlin) hipLaunchKernelGGL((hconv_line_kernel<2048, RPDE_HCONV_WPC, 2>), grid, dim3(128), 0, st.s, c);
    else if (mean) hipLaunchKernelGGL((hconv_line_kernel<2048, RPDE_HCONV_WPC, 1>), grid, dim3(128), 0, st.s, c);
    else hipLaunchKernelGGL((hconv_line_kernel<2048, RPDE_HCONV_WPC>), grid, dim3(128), 0, st.s, c);
    RPDE_HIP(hipGetLastError());
    return true;
  }
  if (c.N == 1024) {   // one wave per line on the half-length core; a 1025^2 grid is four lines per CU: the whole register file per wave
    if (mean && c.tp) hipLaunchKernelGGL((hconv_line_kernel<1024, RPDE_HCONV_WPC, 3>), grid, dim3(64), 0, st.s, c);
    else if (mean && c.nonlin) hipLaunchKernelGGL((hconv_line_kernel<1024, RPDE_HCONV_WPC, 2>), grid, dim3(64), 0, st.s, c);
    else if (mean) hipLaunchKernelGGL((hconv_line_kernel<1024, RPDE_HCONV_WPC, 1>), grid, dim3(64), 0, st.s, c);
    else hipLaunchKernelGGL((hconv_line_kernel<1024, RPDE_HCONV_WPC>), grid, dim3(64), 0, st.s, c);
    RPDE_HIP(hipGetLastError());
    return true;
  }
  // 4097-point lines: the full-length core (168 VGPRs, three workgroups per CU); on the half-length core the term spills
  if (mean && c.tp) hipLaunchKernelGGL((conv_line_kernel<4096, 3>), grid, dim3(256), 0, st.s, c);
  else if (mean && c.nonlin) hipLaunchKernelGGL((conv_line_kernel<4096, 2>), grid, dim3(256), 0, st.s, c);
  else if (mean) hipLaunchKernelGGL((conv_line_kernel<4096, 1>), grid, dim3(256), 0, st.s, c);
  else hipLaunchKernelGGL(conv_line_kernel<4096>, grid, dim3(256), 0, st.s, c);
  RPDE_HIP(hipGetLastError());
  return true;
}
bool launch_rfft_pair(const RfftLineArgs& a0, const RfftLineArgs& a1, Stream& st) {
  const int N = a0.N;
  if (N != a1.N || a0.nlines != a1.nlines || !rfft_line_ok(a0) || !rfft_line_ok(a1) || N == 256) return false;
  if (a0.nlines <= 0) return true;
  const dim3 grid(8 * ((a0.nlines + 7) / 8));
  const size_t one = sizeof(double) * hdct_lds_doubles(N);
  if (N == 1024) hipLaunchKernelGGL(rfft_pair_kernel<1024>, grid, dim3(128), 2 * one, st.s, a0, a1);
  else if (N == 4096) {
    lds_permission(rfft_pair_kernel<4096>, 2 * one);
    hipLaunchKernelGGL(rfft_pair_kernel<4096>, grid, dim3(512), 2 * one, st.s, a0, a1);
  } else if (N == 8192) {   // long lines: the two transforms in a row in one buffer (rfft_seq2_line)
    lds_permission(rfft_seq2_kernel<8192>, one);
    hipLaunchKernelGGL(rfft_seq2_kernel<8192>, grid, dim3(512), one, st.s, a0, a1);
  } else {
    lds_permission(rfft_seq2_kernel<16384>, one);
    hipLaunchKernelGGL(rfft_seq2_kernel<16384>, grid, dim3(1024), one, st.s, a0, a1);
  }
  RPDE_HIP(hipGetLastError());
  return true;
}
bool launch_four_rhs(const FourRhsArgs& a, Stream& st) {
  const int N = a.f.N;
  if (!four_rhs_ok(a) || N == 256) return false;
  if (a.f.nlines <= 0) return true;
  const dim3 grid(8 * ((a.f.nlines + 7) / 8));
  const size_t one = sizeof(double) * hdct_lds_doubles(N);
  if (N == 1024) hipLaunchKernelGGL(four_rhs_kernel<1024>, grid, dim3(64), one, st.s, a);
  else if (N == 4096) hipLaunchKernelGGL(four_rhs_kernel<4096>, grid, dim3(256), one, st.s, a);
  else if (N == 8192) { lds_permission(four_rhs_kernel<8192>, one); hipLaunchKernelGGL(four_rhs_kernel<8192>, grid, dim3(512), one, st.s, a); }
  else { lds_permission(four_rhs_kernel<16384>, one); hipLaunchKernelGGL(four_rhs_kernel<16384>, grid, dim3(1024), one, st.s, a); }
  RPDE_HIP(hipGetLastError());
  return true;
}
bool launch_dct_line2(const DctLineArgs& a0, const DctLineArgs& a1, Stream& st) {
  if (a0.N != a1.N || a0.nlines != a1.nlines || !dct_line_ok(a0) || !dct_line_ok(a1) || (a0.N != 4096 && a0.N != 1024)) return false;
  if (a0.nlines <= 0) return true;
  const dim3 grid(8 * ((a0.nlines + 7) / 8));
  if (s1_split_on()) return launch_dct_line(a0, st) && launch_dct_line(a1, st);
  if (s1_pair_on() && hdct_pair_ok(a0, a1)) {
    const size_t bytes = 2 * sizeof(double) * hdct_lds_doubles(a0.N);
    if (a0.N == 1024) hipLaunchKernelGGL(hdct_pair_kernel<1024>, grid, dim3(128), bytes, st.s, a0, a1);
    else {
      static std::atomic<int> configured[32];             // dynamic-LDS permission above 64 KB, per device
      int dev = 0;
      RPDE_HIP(hipGetDevice(&dev));
      if (!configured[dev & 31].load(std::memory_order_acquire)) {
        RPDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hdct_pair_kernel<4096>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        configured[dev & 31].store(1, std::memory_order_release);
      }
      hipLaunchKernelGGL(hdct_pair_kernel<4096>, grid, dim3(512), bytes, st.s, a0, a1);
    }
  } else if (a0.N == 1024) hipLaunchKernelGGL((hdct_line2_kernel<1024>), grid, dim3(64), 0, st.s, a0, a1);
  else hipLaunchKernelGGL(hdct_line2_kernel<4096>, grid, dim3(256), 0, st.s, a0, a1);
  RPDE_HIP(hipGetLastError());
  return true;
}

// sustained f64 MFMA rate of the chip (no memory traffic): 4 waves per workgroup, 8 independent
// accumulator chains per wave, `iters` x 8 v_mfma_f64_16x16x4_f64 per wave.  The achievable peak a
// GEMM can be priced against once the clock has settled under the matrix load.
// KIND 0: accumulators pinned to VGPRs, 1: to AGPRs (inline asm, in-place accumulate: the compiler adds no
// register moves).  16 independent chains per wave, `iters` x 16 v_mfma_f64_16x16x4_f64 per wave.
typedef double dbl4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* out, int iters) {
  dbl4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = dbl4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;   // keep the chains alive
}
void launch_mfma_peak(double* out, int blocks, int iters, Stream& st, int kind) {
  if (kind == 0) hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(256), 0, st.s, out, iters);
  else hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0, st.s, out, iters);
  RPDE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------- exchange pack / unpack
template <class E, int TS>
__global__ __launch_bounds__(256) void xchg_pack_kernel(const XchgDesc d, E* __restrict__ send) {
  __shared__ E tile[TS * (TS + 1)];
  const int q = blockIdx.z / d.nA, a = blockIdx.z % d.nA;
  const int c0 = d.c0[q], cq = d.c0[q + 1] - c0, rl = d.rl;
  const int bc = blockIdx.x * TS, br = blockIdx.y * TS;
  if (bc >= cq || br >= rl) return;
  // the block of columns [c0, c0 + cq) of array a, transposed into destination q's segment
  const long ldi = d.ldi / (long)(sizeof(E) / sizeof(double));
  const E* in = reinterpret_cast<const E*>(d.in[a]) + c0;
  E* out = send + d.soff[q] / (long)(sizeof(E) / sizeof(double)) + (long)a * cq * rl;
  Blk blk{0, 0, 256, nullptr};
  transpose_tile<E, TS>(blk, tile, in, ldi, out, (long)rl, rl, cq, br, bc);
}
template <class E>
__global__ __launch_bounds__(256) void xchg_unpack_kernel(const XchgDesc d, const E* __restrict__ recv) {
  const int s = blockIdx.z / d.nA, a = blockIdx.z % d.nA;
  const int r0 = d.r0[s], rs = d.r0[s + 1] - r0, cl = d.cl;
  const E* in = recv + d.roff[s] / (long)(sizeof(E) / sizeof(double)) + (long)a * cl * rs;
  E* out = reinterpret_cast<E*>(d.out[a]);
  const long ldo = d.ldo / (long)(sizeof(E) / sizeof(double));
  for (int r = blockIdx.y; r < cl; r += gridDim.y)
    for (int c = blockIdx.x * 256 + threadIdx.x; c < rs; c += gridDim.x * 256)
      out[(long)r * ldo + r0 + c] = in[(long)r * rs + c];
}
void launch_xchg_pack(const XchgDesc& d, double* send, Stream& st) {
  int cmax = 0;
  for (int q = 0; q < d.P; ++q) cmax = std::max(cmax, d.c0[q + 1] - d.c0[q]);
  if (cmax <= 0 || d.rl <= 0) return;
  if (d.elem == 1) {
    dim3 grid((cmax + 63) / 64, (d.rl + 63) / 64, d.P * d.nA);
    hipLaunchKernelGGL((xchg_pack_kernel<double, 64>), grid, dim3(256), 0, st.s, d, send);
  } else {
    dim3 grid((cmax + 31) / 32, (d.rl + 31) / 32, d.P * d.nA);
    hipLaunchKernelGGL((xchg_pack_kernel<Cplx, 32>), grid, dim3(256), 0, st.s, d, reinterpret_cast<Cplx*>(send));
  }
  RPDE_HIP(hipGetLastError());
}
void launch_xchg_unpack(const XchgDesc& d, const double* recv, Stream& st) {
  int rmax = 0;
  for (int s = 0; s < d.P; ++s) rmax = std::max(rmax, d.r0[s + 1] - d.r0[s]);
  if (rmax <= 0 || d.cl <= 0) return;
  dim3 grid(std::min((rmax + 255) / 256, 16), std::min(d.cl, 2048), d.P * d.nA);
  if (d.elem == 1) hipLaunchKernelGGL((xchg_unpack_kernel<double>), grid, dim3(256), 0, st.s, d, recv);
  else hipLaunchKernelGGL((xchg_unpack_kernel<double2>), grid, dim3(256), 0, st.s, d, reinterpret_cast<const double2*>(recv));
  RPDE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------- small kernels
__global__ __launch_bounds__(256) void copy2d_kernel(const double* __restrict__ in, long ldi,
                                                     double* __restrict__ out, long ldo, int rows,
                                                     int cols) {
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    for (int c = blockIdx.x * 256 + threadIdx.x; c < cols; c += gridDim.x * 256)
      out[(long)r * ldo + c] = in[(long)r * ldi + c];
}
void launch_copy2d(const double* in, long ldi, double* out, long ldo, int rows, int cols, Stream& st) {
  if (rows <= 0 || cols <= 0) return;
  dim3 grid((cols + 255) / 256 < 64 ? (cols + 255) / 256 : 64, rows < 4096 ? rows : 4096);
  hipLaunchKernelGGL(copy2d_kernel, grid, dim3(256), 0, st.s, in, ldi, out, ldo, rows, cols);
  RPDE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------- three-term stencil (pdma.h)
__global__ __launch_bounds__(256) void sten3_rows_kernel(const Sten3RowsArgs a) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= a.ncols) return;
  for (int j = a.row0 + blockIdx.y; j < a.row0 + a.nrows; j += gridDim.y) sten3_rows_point(a, j, c);
}
void launch_sten3_rows(const Sten3RowsArgs& a, Stream& st) {
  if (a.nrows <= 0 || a.ncols <= 0) return;
  dim3 grid((a.ncols + 255) / 256, std::min(a.nrows, 1024));
  hipLaunchKernelGGL(sten3_rows_kernel, grid, dim3(256), 0, st.s, a);
  RPDE_HIP(hipGetLastError());
}
__global__ __launch_bounds__(64) void pdma_cols_kernel(const PdmaColsArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.ncols) return;
  if (pdma_column(a, c) && a.nanflag) *a.nanflag = 1;
}
// blocked form (pdma.h): PH 0 forward from zero inflow, 2 forward correction + backward from zero inflow, 4 backward correction:
// one thread per column of one block; PH 1 / 3: the serial carries, one thread per column
template <int PH>
__global__ __launch_bounds__(64) void pdma_cols_blk_kernel(const PdmaColsArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= a.ncols) return;
  if constexpr (PH == 0) pdma_blk_fwd_local(a, (int)blockIdx.y, c);
  else if constexpr (PH == 1) pdma_blk_fwd_carry(a, c);
  else if constexpr (PH == 2) pdma_blk_mid(a, (int)blockIdx.y, c);
  else if constexpr (PH == 3) pdma_blk_bwd_carry(a, c);
  else { if (pdma_blk_final(a, (int)blockIdx.y, c) && a.nanflag) *a.nanflag = 1; }
}
void launch_pdma_cols(const PdmaColsArgs& a, Stream& st) {
  if (a.n <= 0 || a.ncols <= 0) return;
  if (a.blk.phi1) {
    RPDE_REQUIRE(a.ws && a.ldw >= a.ncols && a.blk.NB == (a.n + kPdmaBR - 1) / kPdmaBR, "pdma_cols: workspace / tables of the blocked form");
    RPDE_REQUIRE(a.in != a.out, "pdma_cols (blocked form): in place is not supported -- a block reads rows its neighbours write");
    const dim3 tiles((a.ncols + 63) / 64), blocks((a.ncols + 63) / 64, a.blk.NB);
    hipLaunchKernelGGL(pdma_cols_blk_kernel<0>, blocks, dim3(64), 0, st.s, a);
    hipLaunchKernelGGL(pdma_cols_blk_kernel<1>, tiles, dim3(64), 0, st.s, a);
    hipLaunchKernelGGL(pdma_cols_blk_kernel<2>, blocks, dim3(64), 0, st.s, a);
    hipLaunchKernelGGL(pdma_cols_blk_kernel<3>, tiles, dim3(64), 0, st.s, a);
    hipLaunchKernelGGL(pdma_cols_blk_kernel<4>, blocks, dim3(64), 0, st.s, a);
    RPDE_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(pdma_cols_kernel, dim3((a.ncols + 63) / 64), dim3(64), 0, st.s, a);   // one wave per workgroup: the columns spread over as many CUs as there are waves
  RPDE_HIP(hipGetLastError());
}
__global__ __launch_bounds__(64) void sten3_lines_kernel(const Sten3LinesArgs a) {
  const int line = blockIdx.x * 64 + threadIdx.x;
  if (line < a.nlines) sten3_line(a, line, blockIdx.y);
}
void launch_sten3_lines(const Sten3LinesArgs& a, Stream& st) {
  if (a.nlines <= 0) return;
  hipLaunchKernelGGL(sten3_lines_kernel, dim3((a.nlines + 63) / 64, a.ncomp), dim3(64), 0, st.s, a);
  RPDE_HIP(hipGetLastError());
}
__global__ __launch_bounds__(64) void pdma_lines_kernel(const PdmaLinesArgs a) {
  const int line = blockIdx.x * 64 + threadIdx.x;
  if (line < a.nlines) pdma_line(a, line, blockIdx.y);
}
void launch_pdma_lines(const PdmaLinesArgs& a, Stream& st) {
  if (a.nlines <= 0) return;
  hipLaunchKernelGGL(pdma_lines_kernel, dim3((a.nlines + 63) / 64, a.ncomp), dim3(64), 0, st.s, a);
  RPDE_HIP(hipGetLastError());
}

__global__ void set_element_kernel(double* p, long idx, double v) { p[idx] = v; }
void launch_set_element(double* p, long idx, double value, Stream& st) {
  hipLaunchKernelGGL(set_element_kernel, dim3(1), dim3(1), 0, st.s, p, idx, value);
  RPDE_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void sumsq_kernel(const double* __restrict__ a, long ld, int rows,
                                                    int cols, double* out2) {
  __shared__ double ssum[256];
  __shared__ double snan[256];
  double s = 0.0, nn = 0.0;
  for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x; c < cols; c += 256) {
      const double v = a[(long)r * ld + c];
      if (v != v) nn += 1.0; else s += v * v;
    }
  ssum[threadIdx.x] = s; snan[threadIdx.x] = nn;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; snan[threadIdx.x] += snan[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicAdd(&out2[0], ssum[0]); atomicAdd(&out2[1], snan[0]); }
}
void launch_sumsq(const double* a, long ld, int rows, int cols, double* out2, Stream& st) {
  RPDE_HIP(hipMemsetAsync(out2, 0, 2 * sizeof(double), st.s));
  const int grid = rows < 1024 ? rows : 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, st.s, a, ld, rows, cols, out2);
  RPDE_HIP(hipGetLastError());
}

#else
// =================================================================================== EMU build
template <class Cfg>
static void launch_cfg(const Program& pg, Stream&) {
  const size_t nd = line_lds_doubles(pg.nslots, pg.slot_len, Cfg::kMaxSlotLen, Cfg::kCarryLen);
  RPDE_REQUIRE(nd * sizeof(double) <= 160 * 1024, "line program needs more than 160 KiB of LDS");
  constexpr size_t guard = 16;   // NaN in front of slot 0 as well
  std::vector<double> lds(nd + guard);
  for (int comp = 0; comp < pg.ncomp; ++comp)
    for (int line = 0; line < pg.nlines; ++line) {
      // poison the LDS like uninitialised hardware memory would be
      std::fill(lds.begin(), lds.end(), std::nan(""));
      Blk blk{line, comp, Cfg::T, lds.data() + guard};
      run_line_program<Cfg>(blk, pg);
    }
}

template <class E, int TS>
static void emu_transpose(const E* in, long ldi, E* out, long ldo, int rows, int cols) {
  std::vector<E> tile((size_t)TS * (TS + 1));
  Blk blk{0, 0, 256, nullptr};
  for (int r0 = 0; r0 < rows; r0 += TS)
    for (int c0 = 0; c0 < cols; c0 += TS) transpose_tile<E, TS>(blk, tile.data(), in, ldi, out, ldo, rows, cols, r0, c0);
}
void launch_transpose(const double* in, long ldi, double* out, long ldo, int rows, int cols,
                      int elem, Stream&) {
  if (rows <= 0 || cols <= 0) return;
  if (elem == 1) {
    emu_transpose<double, 64>(in, ldi, out, ldo, rows, cols);
  } else {
    RPDE_REQUIRE(elem == 2 && ldi % 2 == 0 && ldo % 2 == 0, "complex transpose needs even pitches");
    emu_transpose<Cplx, 32>(reinterpret_cast<const Cplx*>(in), ldi / 2, reinterpret_cast<Cplx*>(out), ldo / 2, rows, cols);
  }
}
void launch_transpose_batch(const TransposeBatch& b, int n, long ldi, long ldo, int rows, int cols, int elem, Stream& st) {
  for (int a = 0; a < n; ++a) launch_transpose(b.in[a], ldi, b.out[a], ldo, rows, cols, elem, st);
}
void launch_gemm_nt(int M, int N, int K, const double* A, long lda, const double* B, long ldb,
                    double* C, long ldc, Stream&) {
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += A[(long)m * lda + k] * B[(long)n * ldb + k];
      C[(long)m * ldc + n] = s;
    }
}
void launch_gemm_nn(int M, int N, int K, const double* A, long lda, const double* B, long ldb,
                    double* C, long ldc, Stream&) {
  std::vector<double> row(N);
  for (int m = 0; m < M; ++m) {
    std::fill(row.begin(), row.end(), 0.0);
    for (int k = 0; k < K; ++k) {
      const double a = A[(long)m * lda + k];
      const double* b = B + (long)k * ldb;
      for (int n = 0; n < N; ++n) row[n] += a * b[n];
    }
    for (int n = 0; n < N; ++n) C[(long)m * ldc + n] = row[n];
  }
}
void launch_gemm_pair(bool nn, const GemmProblem& p0, const GemmProblem& p1, Stream& st) {
  for (const GemmProblem* p : {&p0, &p1}) {
    if (p->M <= 0 || p->N <= 0) continue;
    std::vector<double> tmp;
    double* C = p->C;
    long ldc = p->ldc;
    if (p->ct) { tmp.resize((size_t)p->M * p->N); C = tmp.data(); ldc = p->N; }
    if (nn) launch_gemm_nn(p->M, p->N, p->K, p->A, p->lda, p->B, p->ldb, C, ldc, st);
    else launch_gemm_nt(p->M, p->N, p->K, p->A, p->lda, p->B, p->ldb, C, ldc, st);
    if (p->ct)
      for (int m = 0; m < p->M; ++m)
        for (int n = 0; n < p->N; ++n) p->C[(long)n * p->ldc + m] = tmp[(size_t)m * p->N + n];
    if (p->zero00) p->C[0] = 0.0;
  }
}
void launch_xchg_pack(const XchgDesc& d, double* send, Stream&) {
  for (int q = 0; q < d.P; ++q)
    for (int a = 0; a < d.nA; ++a) {
      const int c0 = d.c0[q], cq = d.c0[q + 1] - c0;
      if (cq <= 0 || d.rl <= 0) continue;
      if (d.elem == 1) {
        emu_transpose<double, 64>(d.in[a] + c0, d.ldi, send + d.soff[q] + (long)a * cq * d.rl, d.rl, d.rl, cq);
      } else {
        emu_transpose<Cplx, 32>(reinterpret_cast<const Cplx*>(d.in[a]) + c0, d.ldi / 2,
                                reinterpret_cast<Cplx*>(send + d.soff[q]) + (long)a * cq * d.rl, d.rl, d.rl, cq);
      }
    }
}
void launch_xchg_unpack(const XchgDesc& d, const double* recv, Stream&) {
  for (int s = 0; s < d.P; ++s)
    for (int a = 0; a < d.nA; ++a) {
      const int r0 = d.r0[s], rs = d.r0[s + 1] - r0, e = d.elem;
      const double* in = recv + d.roff[s] + (long)a * d.cl * rs * e;
      for (int r = 0; r < d.cl; ++r)
        for (int c = 0; c < rs * e; ++c) d.out[a][(long)r * d.ldo + (long)r0 * e + c] = in[(long)r * rs * e + c];
    }
}
void launch_copy2d(const double* in, long ldi, double* out, long ldo, int rows, int cols, Stream&) {
  for (int r = 0; r < rows; ++r)
    for (int cc = 0; cc < cols; ++cc) out[(long)r * ldo + cc] = in[(long)r * ldi + cc];
}
void launch_col_diff1(const ColDiff1Args& A, Stream&) {
  // the workgroups of a column tile from the top super-block down (the ticket order of the device)
  const ColDiffArgs& a = A.a;
  if (a.ncols <= 0 || a.nout <= 0) return;
  RPDE_REQUIRE(A.NSB * kDiff1Rows >= a.nout && A.NSB <= 64 && a.row0 == 0 && a.nranks <= 1, "coldiff1: one rank, at most 64 super-blocks");
  std::vector<double> lt((size_t)kDiff1W * 2 * kCol1Tile), ds((size_t)kDiff1W * kCol1Tile * kDiff1BR);
  for (int tile = 0; tile < A.tiles; ++tile)
    for (int q = A.NSB - 1; q >= 0; --q) {
      for (int w = 0; w < kDiff1W; ++w)
        for (int lane = 0; lane < kCol1Tile; ++lane) {
          const int j0 = (q * kDiff1W + w) * kDiff1BR, i = tile * kCol1Tile + lane;
          double d[kDiff1BR] = {}, tot[2] = {0.0, 0.0};
          if (j0 < a.nout && i < a.ncols) coldiff1_local(a, j0, i, d, tot, [&](int u) { return a.low[(j0 - 1 + u > 0) ? j0 - 1 + u : 0]; });
          lt[((size_t)w * 2 + 0) * kCol1Tile + lane] = tot[0]; lt[((size_t)w * 2 + 1) * kCol1Tile + lane] = tot[1];
          for (int u = 0; u < kDiff1BR; ++u) ds[((size_t)w * kCol1Tile + lane) * kDiff1BR + u] = d[u];
        }
      double* mine = A.tot + coldiff1_tot(A, tile, q);
      for (int lane = 0; lane < kCol1Tile; ++lane)
        for (int par = 0; par < 2; ++par) {
          double t = 0.0;
          for (int w = kDiff1W - 1; w >= 0; --w) t += lt[((size_t)w * 2 + par) * kCol1Tile + lane];   // (the device adds wave 0's own sum last as well)
          mine[par * kCol1Tile + lane] = t;
        }
      for (int w = 0; w < kDiff1W; ++w)
        for (int lane = 0; lane < kCol1Tile; ++lane) {
          const int j0 = (q * kDiff1W + w) * kDiff1BR, i = tile * kCol1Tile + lane;
          if (!(j0 < a.nout && i < a.ncols)) continue;
          double in[2] = {0.0, 0.0}, sup[2] = {0.0, 0.0}, d[kDiff1BR];
          for (int u = kDiff1W - 1; u > w; --u) { in[0] += lt[((size_t)u * 2 + 0) * kCol1Tile + lane]; in[1] += lt[((size_t)u * 2 + 1) * kCol1Tile + lane]; }
          for (int u = q + 1; u < A.NSB; ++u) {
            const double* th = A.tot + coldiff1_tot(A, tile, u);
            sup[0] += th[lane]; sup[1] += th[kCol1Tile + lane];
          }
          in[0] += sup[0]; in[1] += sup[1];
          for (int u = 0; u < kDiff1BR; ++u) d[u] = ds[((size_t)w * kCol1Tile + lane) * kDiff1BR + u];
          coldiff1_store(a, j0, i, d, in);
        }
    }
}
void launch_col_hholtz1(const ColHh1Args& A, Stream&) {
  // the same per-thread functions and chains, the workgroups of a column tile one after the other: all their aggregates
  // first (what the arrival counter waits for on the device), then the inflows and the rows
  const ColHhArgs& a = A.a;
  if (a.ncols <= 0 || a.n <= 0 || a.nf <= 0 || a.NB <= 0) return;
  RPDE_REQUIRE(A.NSB <= kCol1MaxNSB && A.W * A.NSB >= a.NB, "colhh1: super-block partition");
  const int W = A.W;
  std::vector<double> loc((size_t)A.NSB * W * kCol1Agg * kCol1Tile), stg((size_t)A.NSB * kCol1Stg * kCol1Tile), tbl((size_t)A.NSB * W * kCol1TabPerBlock),
      twl((size_t)A.NSB * kCol1TabPerBlock);
  double r[kColBR + 4];
  for (int f = 0; f < a.nf; ++f)
    for (int tile = 0; tile < A.tiles; ++tile) {
      double* ag = A.agg + col1_agg(A, f, tile, 0);
      for (int e = 0; e < A.NSB * W * kCol1TabPerBlock; ++e) { const double* p = colhh1_block_tab(a.tab[f], e / kCol1TabPerBlock, a.NB, e % kCol1TabPerBlock); tbl[e] = p ? *p : colhh1_ident_tab(e % kCol1TabPerBlock); }
      for (int e = 0; e < A.NSB * kCol1TabPerBlock; ++e) twl[e] = *colhh1_super_tab(A.x[f], e / kCol1TabPerBlock, e % kCol1TabPerBlock);
      for (int q = 0; q < A.NSB; ++q) {
        double* lq = loc.data() + (size_t)q * W * kCol1Agg * kCol1Tile;
        for (int w = 0; w < W; ++w)
          for (int lane = 0; lane < kCol1Tile; ++lane) {
            ColLoc L{};
            const int b = q * W + w, i = tile * kCol1Tile + lane;
            if (b < a.NB && i < a.ncols) colhh1_local(a, f, b, i, r, L, Col1TabDirect(a.tab[f], A.x[f], b * kColBR, b * kColBR - a.shift[f]));
            for (int k = 0; k < kCol1Agg; ++k) lq[(w * kCol1Agg + k) * kCol1Tile + lane] = L.v[k];
          }
        double* mine = ag + (long)q * (kCol1Agg * kCol1Tile);
        for (int lane = 0; lane < kCol1Tile; ++lane) {
          double d = 0.0;
          if (a.tab[f].w) for (int u = 0; u < W; ++u) d += lq[(u * kCol1Agg + 6) * kCol1Tile + lane];
          mine[6 * kCol1Tile + lane] = d;
          for (int par = 0; par < 2; ++par) {
            double so, T0, T1;
            colhh1_chain<true>(lq, tbl.data() + (size_t)q * W * kCol1TabPerBlock, W, par, lane, 0.0, 0.0, 0.0, so, T0, T1);
            mine[par * kCol1Tile + lane] = so; mine[(2 + 2 * par) * kCol1Tile + lane] = T0; mine[(3 + 2 * par) * kCol1Tile + lane] = T1;
          }
        }
      }
      // the last workgroup of the tile: aggregates -> inflow states of every super-block, published in place
      std::vector<double> kap(kCol1Tile, 0.0);
      for (int u = 0; u < A.NSB; ++u)
        for (int k = 0; k < kCol1Stg; ++k)
          for (int lane = 0; lane < kCol1Tile; ++lane) stg[((size_t)u * kCol1Stg + k) * kCol1Tile + lane] = ag[((long)u * kCol1Agg + k) * kCol1Tile + lane];
      for (int lane = 0; lane < kCol1Tile; ++lane) {
        for (int par = 0; par < 2; ++par) colhh1_sweep(stg.data(), twl.data(), A.NSB, par, lane);
        if (a.tab[f].w) for (int u = 0; u < A.NSB; ++u) kap[lane] += ag[((long)u * kCol1Agg + 6) * kCol1Tile + lane];
      }
      for (int u = 0; u < A.NSB; ++u)
        for (int k = 0; k < kCol1Stg; ++k)
          for (int lane = 0; lane < kCol1Tile; ++lane) ag[((long)u * kCol1Agg + k) * kCol1Tile + lane] = stg[((size_t)u * kCol1Stg + k) * kCol1Tile + lane];
      for (int lane = 0; lane < kCol1Tile; ++lane) ag[6 * kCol1Tile + lane] = kap[lane];
      for (int q = 0; q < A.NSB; ++q) {
        double* lq = loc.data() + (size_t)q * W * kCol1Agg * kCol1Tile;
        const double* mine = ag + (long)q * (kCol1Agg * kCol1Tile);
        for (int lane = 0; lane < kCol1Tile; ++lane)
          for (int par = 0; par < 2; ++par) {
            double so, T0, T1;
            colhh1_chain<false>(lq, tbl.data() + (size_t)q * W * kCol1TabPerBlock, W, par, lane, mine[par * kCol1Tile + lane], mine[(2 + 2 * par) * kCol1Tile + lane],
                                mine[(3 + 2 * par) * kCol1Tile + lane], so, T0, T1);
          }
        for (int w = 0; w < W; ++w)
          for (int lane = 0; lane < kCol1Tile; ++lane) {
            const int b = q * W + w, i = tile * kCol1Tile + lane;
            if (!(b < a.NB && i < a.ncols)) continue;
            ColLoc L{};
            const Col1TabDirect tabs(a.tab[f], A.x[f], b * kColBR, b * kColBR - a.shift[f]);
            colhh1_local(a, f, b, i, r, L, tabs);     // the rows again (registers on the device)
            double in6[kCol1Inf];
            for (int k = 0; k < kCol1Inf; ++k) in6[k] = lq[(w * kCol1Agg + k) * kCol1Tile + lane];
            colhh1_final(a, f, b, i, r, in6, ag[6 * kCol1Tile + lane], tabs, true);
          }
      }
    }
}
void launch_set_element(double* p, long idx, double value, Stream&) { p[idx] = value; }
void launch_sten3_rows(const Sten3RowsArgs& a, Stream&) {
  for (int j = a.row0; j < a.row0 + a.nrows; ++j)
    for (int c = 0; c < a.ncols; ++c) sten3_rows_point(a, j, c);
}
void launch_pdma_cols(const PdmaColsArgs& a, Stream&) {
  if (a.blk.phi1) {   // the blocked form, phase by phase like the device
    RPDE_REQUIRE(a.ws && a.ldw >= a.ncols && a.blk.NB == (a.n + kPdmaBR - 1) / kPdmaBR, "pdma_cols: workspace / tables of the blocked form");
    RPDE_REQUIRE(a.in != a.out, "pdma_cols (blocked form): in place is not supported -- a block reads rows its neighbours write");
    for (int b = 0; b < a.blk.NB; ++b) for (int c = 0; c < a.ncols; ++c) pdma_blk_fwd_local(a, b, c);
    for (int c = 0; c < a.ncols; ++c) pdma_blk_fwd_carry(a, c);
    for (int b = 0; b < a.blk.NB; ++b) for (int c = 0; c < a.ncols; ++c) pdma_blk_mid(a, b, c);
    for (int c = 0; c < a.ncols; ++c) pdma_blk_bwd_carry(a, c);
    for (int b = 0; b < a.blk.NB; ++b) for (int c = 0; c < a.ncols; ++c) if (pdma_blk_final(a, b, c) && a.nanflag) *a.nanflag = 1;
    return;
  }
  for (int c = 0; c < a.ncols; ++c)
    if (pdma_column(a, c) && a.nanflag) *a.nanflag = 1;
}
void launch_sten3_lines(const Sten3LinesArgs& a, Stream&) {
  for (int comp = 0; comp < a.ncomp; ++comp)
    for (int l = 0; l < a.nlines; ++l) sten3_line(a, l, comp);
}
void launch_pdma_lines(const PdmaLinesArgs& a, Stream&) {
  for (int comp = 0; comp < a.ncomp; ++comp)
    for (int l = 0; l < a.nlines; ++l) pdma_line(a, l, comp);
}
void launch_mfma_peak(double*, int, int, Stream&) {}
void launch_diag_reduce(const double* T, const double* dT, const double* ux, const double* uy, long ld, int nx, int ny,
                        const double* wx, const double* wy, double c_nu, double c_v1, double c_v2, double c_re,
                        double* partial, double* out4, Stream&) {
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < nx; ++i) {
    const long row = (long)i * ld;
    double av = 0.0, ar = 0.0;
    for (int j = 0; j < ny; ++j) {
      av += wy[j] * (c_v1 * dT[row + j] + c_v2 * T[row + j] * uy[row + j]);
      ar += wy[j] * (c_re * std::sqrt(ux[row + j] * ux[row + j] + uy[row + j] * uy[row + j]));
    }
    partial[4 * i + 0] = wx[i] * c_nu * dT[row];
    partial[4 * i + 1] = wx[i] * c_nu * dT[row + ny - 1];
    partial[4 * i + 2] = wx[i] * av;
    partial[4 * i + 3] = wx[i] * ar;
    for (int c = 0; c < 4; ++c) a[c] += partial[4 * i + c];
  }
  for (int c = 0; c < 4; ++c) out4[c] = a[c];
}
void launch_col_hholtz_phase(const ColHhArgs& a, int phase, Stream&) {
  for (int f = 0; f < a.nf; ++f) {
    if (phase == 0) for (int b = 0; b < a.NB; ++b) for (int i = 0; i < a.ncols; ++i) colhh_block<false>(a, f, b, i);
    if (phase == 1) for (int par = 0; par < 2; ++par) for (int i = 0; i < a.ncols; ++i) { if (a.nranks <= 1) colhh_carry<0>(a, f, i, par); else colhh_carry<1>(a, f, i, par); }
    if (phase == 2) for (int par = 0; par < 2; ++par) for (int i = 0; i < a.ncols; ++i) colhh_carry<2>(a, f, i, par);
    if (phase == 3) for (int b = 0; b < a.NB; ++b) for (int i = 0; i < a.ncols; ++i) colhh_block<true>(a, f, b, i);
  }
}
void launch_col_diff_phase(const ColDiffArgs& a, int phase, Stream&) {
  if (phase == 0) for (int b = 0; b < a.NB; ++b) for (int i = 0; i < a.ncols; ++i) coldiff_pass<false>(a, b, i);
  if (phase == 1) for (int par = 0; par < 2; ++par) for (int i = 0; i < a.ncols; ++i) { if (a.nranks <= 1) coldiff_carry<0>(a, i, par); else coldiff_carry<1>(a, i, par); }
  if (phase == 2) for (int par = 0; par < 2; ++par) for (int i = 0; i < a.ncols; ++i) coldiff_carry<2>(a, i, par);
  if (phase == 3) for (int b = 0; b < a.NB; ++b) for (int i = 0; i < a.ncols; ++i) coldiff_pass<true>(a, b, i);
}
bool launch_conv_line(const ConvLineArgs& c, Stream&) {
  if (!conv_line_ok(c)) return false;
  if (c.um && !(c.vm && c.bx && c.by)) return false;
  if (c.tp && !(c.um && c.cz)) return false;
  std::vector<double> lds(hdct_lds_doubles(c.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < c.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, c.N / 16, base};
    if (c.um && c.tp) {   // the adjoint LNSE term
      if (c.N == 1024) hconv_line<1024, 3>(blk, c);
      else if (c.N == 2048) hconv_line<2048, 3>(blk, c);
      else if (c.N == 4096) conv_line<4096, 3>(blk, c); else conv_line<256, 3>(blk, c);
      continue;
    }
    if (c.um && c.nonlin) {   // Navier2DNonLin: the mean velocities added to u, v
      if (c.N == 1024) hconv_line<1024, 2>(blk, c);
      else if (c.N == 2048) hconv_line<2048, 2>(blk, c);
      else if (c.N == 4096) conv_line<4096, 2>(blk, c); else conv_line<256, 2>(blk, c);
      continue;
    }
    if (c.um) {   // the linearised term (Navier2DLnse)
      if (c.N == 1024) hconv_line<1024, 1>(blk, c);
      else if (c.N == 2048) hconv_line<2048, 1>(blk, c);
#ifdef RPDE_CONV4096_HALF
      else if (c.N == 4096) hconv_line<4096, 1>(blk, c); else conv_line<256, 1>(blk, c);
#else
      else if (c.N == 4096) conv_line<4096, 1>(blk, c); else conv_line<256, 1>(blk, c);
#endif
      continue;
    }
    if (c.N == 1024) hconv_line<1024>(blk, c);
    else if (c.N == 2048) hconv_line<2048>(blk, c);
#ifdef RPDE_CONV4096_HALF
    else if (c.N == 4096) hconv_line<4096>(blk, c); else conv_line<256>(blk, c);
#else
    else if (c.N == 4096) conv_line<4096>(blk, c); else conv_line<256>(blk, c);   // the device's choice of core per length
#endif
  }
  return true;
}
bool launch_rfft_pair(const RfftLineArgs& a0, const RfftLineArgs& a1, Stream&) {
  if (a0.N != a1.N || a0.nlines != a1.nlines || !rfft_line_ok(a0) || !rfft_line_ok(a1)) return false;
  std::vector<double> lds(2 * hdct_lds_doubles(a0.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a0.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a0.N / 16, base};
    if (a0.N == 16384) rfft_seq2_line<16384>(blk, a0, a1);
    else if (a0.N == 8192) rfft_seq2_line<8192>(blk, a0, a1);
    else if (a0.N == 4096) rfft_pair_line<4096>(line, base, a0, a1);
    else if (a0.N == 1024) rfft_pair_line<1024>(line, base, a0, a1);
    else rfft_pair_line<256>(line, base, a0, a1);
  }
  return true;
}
bool launch_four_rhs(const FourRhsArgs& a, Stream&) {
  if (!four_rhs_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.f.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a.f.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.f.N / 16, base};
    if (a.f.N == 16384) four_rhs_line<16384>(blk, a);
    else if (a.f.N == 8192) four_rhs_line<8192>(blk, a);
    else if (a.f.N == 4096) four_rhs_line<4096>(blk, a);
    else if (a.f.N == 1024) four_rhs_line<1024>(blk, a);
    else four_rhs_line<256>(blk, a);
  }
  return true;
}
bool launch_dct_line2(const DctLineArgs& a0, const DctLineArgs& a1, Stream& st) {
  if (a0.N != a1.N || a0.nlines != a1.nlines || !dct_line_ok(a0) || !dct_line_ok(a1)) return false;
  static const bool pair_on = [] { const char* e = std::getenv("RPDE_S1_PAIR"); return !e || std::atoi(e) != 0; }();
  if (!pair_on || !hdct_pair_ok(a0, a1)) return launch_dct_line(a0, st) && launch_dct_line(a1, st);
  std::vector<double> lds(2 * hdct_lds_doubles(a0.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a0.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    if (a0.N == 4096) hdct_pair_line<4096>(line, base, a0, a1);
    else if (a0.N == 1024) hdct_pair_line<1024>(line, base, a0, a1);
    else hdct_pair_line<256>(line, base, a0, a1);
  }
  return true;
}
bool line_batch_ok(int N) { return N == 1024 || N == 256 || N == 4096; }
void launch_line_batch(const LineBatch& b, Stream& st) {   // the same lines, one field after the other
  for (int i = 0; i < b.n; ++i) {
    bool ok = false;
    if (b.kind == 0) ok = launch_dct_line(b.d0[i], st);
    else if (b.kind == 1) ok = launch_dct_line2(b.d0[i], b.d1[i], st);
    else if (b.kind == 2) ok = launch_conv_line(b.c[i], st);
    else ok = launch_rhs_line(b.r[i], st);
    RPDE_REQUIRE(ok, "line batch: shape");
  }
}
bool launch_div_line(const DivLineArgs& a, Stream&) {
  if (!div_line_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.N / 16, base};
    if (a.N == 4096) div_line<4096>(blk, a); else if (a.N == 1024) div_line<1024>(blk, a); else div_line<256>(blk, a);
  }
  return true;
}
bool launch_corr_line(const CorrLineArgs& a, Stream&) {
  if (!corr_line_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.N / 16, base};
    if (a.N == 4096) corr_line<4096>(blk, a); else if (a.N == 1024) corr_line<1024>(blk, a); else corr_line<256>(blk, a);
  }
  return true;
}
bool launch_prow_line(const ProwLineArgs& a, Stream&) {
  if (!prow_line_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.N / 16, base};
    if (a.derive) { if (a.N == 4096) prow_line<4096, true, true>(blk, a); else if (a.N == 2048) prow_line<2048, true, true>(blk, a); else if (a.N == 1024) prow_line<1024, true, true>(blk, a); else prow_line<256, true, true>(blk, a); }
    else if (a.keep) { if (a.N == 4096) prow_line<4096, true>(blk, a); else if (a.N == 2048) prow_line<2048, true>(blk, a); else if (a.N == 1024) prow_line<1024, true>(blk, a); else prow_line<256, true>(blk, a); }
    else if (a.N == 4096) prow_line<4096>(blk, a); else if (a.N == 2048) prow_line<2048>(blk, a); else if (a.N == 1024) prow_line<1024>(blk, a); else prow_line<256>(blk, a);
  }
  return true;
}
bool launch_pres_line(const PresLineArgs& a, Stream&) {
  if (!pres_line_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.N / 16, base};
    if (a.N == 4096) pres_line<4096>(blk, a); else if (a.N == 1024) pres_line<1024>(blk, a); else pres_line<256>(blk, a);
  }
  return true;
}
void launch_per_rows(const PerRowsArgs& a, Stream&) {
  for (int line = 0; line < a.nlines; ++line)
    for (int k = 0; k < a.kx; ++k) per_rows_point(a, line, k);
}
bool launch_rhs_line(const RhsLineArgs& a, Stream&, long long*) {
  if (!rhs_line_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);
  for (int line = 0; line < a.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.N / 16, base};
    if (a.N == 4096) { if (a.which == 0) rhs_line<4096, 0>(blk, a); else if (a.which == 1) rhs_line<4096, 1>(blk, a); else rhs_line<4096, 2>(blk, a); }
    else if (a.N == 1024) { if (a.which == 0) rhs_line<1024, 0>(blk, a); else if (a.which == 1) rhs_line<1024, 1>(blk, a); else rhs_line<1024, 2>(blk, a); }
    else { if (a.which == 0) rhs_line<256, 0>(blk, a); else if (a.which == 1) rhs_line<256, 1>(blk, a); else rhs_line<256, 2>(blk, a); }
  }
  return true;
}
bool launch_dct_line(const DctLineArgs& a, Stream&, long long*) {
  if (!dct_line_ok(a)) return false;
  std::vector<double> lds(hdct_lds_doubles(a.N) + 2);
  double* base = lds.data() + (((size_t)lds.data() & 15) ? 1 : 0);   // 16-byte aligned like the device buffer
  for (int line = 0; line < a.nlines; ++line) {
    std::fill(lds.begin(), lds.end(), std::nan(""));
    Blk blk{line, 0, a.N / 16, base};
    if (a.N == 1024) hdct_bwd_line_by_mode<1024>(blk, a);
    else if (a.N == 2048) hdct_bwd_line_by_mode<2048>(blk, a);
    else if (a.N == 4096) hdct_bwd_line_by_mode<4096>(blk, a); else hdct_bwd_line_by_mode<256>(blk, a);
  }
  return true;
}

void launch_sumsq(const double* a, long ld, int rows, int cols, double* out2, Stream&) {
  double s = 0.0, nn = 0.0;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const double v = a[(long)r * ld + c];
      if (v != v) nn += 1.0; else s += v * v;
    }
  out2[0] = s; out2[1] = nn;
}
#endif

// =================================================================================== common
static int class_index(int slot_len) {
  if (slot_len <= CfgS::kMaxSlotLen) return 0;
  if (slot_len <= CfgM::kMaxSlotLen) return 1;
  if (slot_len <= CfgL::kMaxSlotLen) return 2;
  if (slot_len <= CfgX::kMaxSlotLen) return 3;
  return -1;
}
LineClass line_class_for(int slot_len) {
  switch (class_index(slot_len)) {
    case 0: return {CfgS::T, CfgS::C};
    case 1: return {CfgM::T, CfgM::C};
    case 2: return {CfgL::T, CfgL::C};
    case 3: return slot_len <= CfgXC::kMaxSlotLen ? LineClass{CfgXC::T, CfgXC::C} : LineClass{CfgX::T, CfgX::C};   // scan tables: Chebyshev axes only
    default: fail("line too long for one workgroup: slot length " + std::to_string(slot_len));
  }
}
std::vector<double> chunk_major(const std::vector<double>& tab, LineClass lc, int dir, double pad) {
  std::vector<double> out((size_t)lc.T * lc.C, pad);
  for (int t = 0; t < lc.T; ++t)
    for (int i = 0; i < lc.C; ++i) {
      const size_t k = (size_t)(dir > 0 ? t : lc.T - 1 - t) * lc.C + i;
      if (k < tab.size()) out[(size_t)i * lc.T + t] = tab[k];
    }
  return out;
}

void launch_line_program(const Program& pg, Stream& st) {
  RPDE_REQUIRE(pg.nops > 0 && pg.nops <= kMaxOps, "bad line program");
  if (pg.nlines <= 0 || pg.ncomp <= 0) return;
  const int sl = pg.slot_len;
  const int ci = class_index(sl);
  const int flen = pg.blu_m > 0 ? pg.blu_m : pg.fft_n;   // the FFT length the kernel has to hold (Bluestein: the convolution length)
  auto fft_ok = [&](int fmin, int fmax) { return flen == 0 || (flen >= fmin && flen <= fmax); };
  if (ci == 0 && fft_ok(CfgS::FMIN, CfgS::FMAX)) launch_cfg<CfgS>(pg, st);
  else if (ci == 1 && fft_ok(CfgM::FMIN, CfgM::FMAX)) launch_cfg<CfgM>(pg, st);
  else if (ci == 2 && fft_ok(CfgL::FMIN, CfgL::FMAX)) launch_cfg<CfgL>(pg, st);
  else if (ci == 3) {
    bool cheb_ops = false;
    for (int i = 0; i < pg.nops; ++i) {
      const int c = pg.ops[i].code;
      cheb_ops |= c == OP_DCT || c == OP_REC1 || c == OP_REC2 || c == OP_CDIFF || c == OP_PUSH || c == OP_POPAXPY || c == OP_STEN || c == OP_MV3;
    }
    if (cheb_ops || pg.nslots > 1) {
      RPDE_REQUIRE(sl <= CfgXC::kMaxSlotLen && (flen == 0 || flen == CfgXC::FMAX) && (pg.fft_n == 0 || pg.blu_m > 0),
                   "the 1024-thread Chebyshev configuration runs lines of up to 4096 points with the Bluestein transform only");
      launch_cfg<CfgXC>(pg, st);
    } else {
      RPDE_REQUIRE(fft_ok(CfgX::FMIN, CfgX::FMAX), "no line-kernel configuration for FFT length " + std::to_string(flen) + " at slot length " + std::to_string(sl));
      launch_cfg<CfgX>(pg, st);
    }
  }
  else fail("no line-kernel configuration for slot length " + std::to_string(sl) +
            " / FFT length " + std::to_string(flen) +
            " (supported: Chebyshev n <= 4097, Fourier nx = 2^k <= 16384 or any nx <= 5461)");
}

}  // namespace rpde
