// Line VM: every 1-D operation of the spectral time step (stencils, Chebyshev differentiation,
// banded solves, DCT-I / real FFT, products, RHS assembly) acts on ONE line that is contiguous in
// HBM.  One workgroup owns one line, stages it in LDS "slots" and runs a short host-built
// program of ops over it, so a chain of operations along the same axis costs one read and one
// write of the array.  Axis changes are explicit transposes (transpose.h) - the same call sites
// become the all-to-all exchange when the field is sharded into pencils.
//
// Reference semantics implemented by the ops (file:line in /root/reference unless noted):
//   OP_STEN       composite -> ortho stencil, funspace `to_ortho`        (src/field.rs:113-115)
//   OP_MV3        banded mat-vec (offsets 0,+2,+4) `MatVecFdma`          (src/solver/matvec.rs:207-228)
//   OP_REC1/REC2  forward / backward substitution of `Fdma::fdma`        (src/solver/fdma.rs:101-118)
//                 and of the stride-2 TDMA behind funspace `from_ortho`  (src/field.rs:118-123)
//   OP_CDIFF      Chebyshev differentiation recurrence (funspace `gradient`, src/field.rs:127-129)
//   OP_DCT        DCT-I (rustdct `process_dct1` under funspace chebyshev forward/backward)
//   OP_RFFT_*     realfft r2c / c2r under funspace fourier_r2c
//   OP_TABDIV     `Sdma` diagonal solve                                  (src/solver/sdma.rs:36-46)
// Sequential recurrences are evaluated as chunked scans: each thread runs its chunk of the
// line, chunk carries are composed through LDS (16-way, two levels), then every thread re-runs
// its chunk with the exact inflow.  Arithmetic per element is identical to the sequential
// recurrence; only the inflow is obtained through affine-map composition.
#pragma once
#include <type_traits>

#include "platform.h"

namespace rpde {

// address-space qualified pointer types: without them every LDS / table access inside the
// non-inlined device functions would be a FLAT instruction (generic pointer), which is several
// times slower than ds_read / global_load and serialises the LDS and vector-memory counters.
#ifdef RPDE_EMU
using lds_t = double*;          // LDS (workgroup) memory
using clds_t = const double*;
using tab_t = const double*;    // read-only tables in global memory
using gmem_t = double*;         // arrays in global memory
using cgmem_t = const double*;
#else
using lds_t = __attribute__((address_space(3))) double*;
using clds_t = const __attribute__((address_space(3))) double*;
using tab_t = const __attribute__((address_space(1))) double*;
using gmem_t = __attribute__((address_space(1))) double*;
using cgmem_t = const __attribute__((address_space(1))) double*;
#endif

// a complex number / pair of doubles moved with one 16-byte LDS access
#ifdef RPDE_EMU
struct dbl2 { double x, y; };
using lds2_t = dbl2*;
#else
typedef double dbl2 __attribute__((ext_vector_type(2)));
using lds2_t = __attribute__((address_space(3))) dbl2*;
#endif
#ifdef RPDE_EMU
using cgmem2_t = const dbl2*;   // two consecutive doubles of a global line moved with one 16-byte access
using gmem2_t = dbl2*;
using clds2_t = const dbl2*;
#else
using clds2_t = const __attribute__((address_space(3))) dbl2*;
using cgmem2_t = const __attribute__((address_space(1))) dbl2*;
using gmem2_t = __attribute__((address_space(1))) dbl2*;
#endif

// A row of doubles read or written in 16-byte pairs through a buffer descriptor: ONE 32-bit per-thread byte offset
// serves every row and every array of a phase (the descriptor and the uniform part of the offset live in SGPRs), where
// flat global addressing spends a 64-bit VGPR pair and the arithmetic for it on every load stream.  The descriptor covers
// exactly `bytes` from the row's first element: an access outside (for instance the pair in front of the line, offset
// -16) reads zero and is not stored.
#ifdef RPDE_EMU
// the emulation follows the hardware's rule: the per-thread offset alone is range-checked (as an unsigned number), the
// uniform offset is simply added -- and it aborts on a final address outside the row, which the hardware would not notice
struct RowBuf { const double* p; long bytes; };
inline RowBuf row_buf(const double* p, long bytes) { return RowBuf{p, bytes}; }
inline bool row_in(const RowBuf& r, int voff, int soff, int size) {
  if (voff < 0 || (long)voff + size > r.bytes) return false;
  const long o = (long)voff + soff;
  if (soff < 0 || o + size > r.bytes) { std::fprintf(stderr, "row access outside its descriptor (%d + %d of %ld)\n", voff, soff, r.bytes); std::abort(); }
  return true;
}
inline dbl2 row_ld2(const RowBuf& r, int voff, int soff) {
  if (!row_in(r, voff, soff, 16)) return dbl2{0.0, 0.0};
  const long o = ((long)voff + soff) / 8;
  return dbl2{r.p[o], r.p[o + 1]};
}
inline double row_ld1(const RowBuf& r, int voff, int soff) {
  if (!row_in(r, voff, soff, 8)) return 0.0;
  return r.p[((long)voff + soff) / 8];
}
inline void row_st2(const RowBuf& r, int voff, int soff, dbl2 v) {
  if (!row_in(r, voff, soff, 16)) return;
  double* q = const_cast<double*>(r.p) + ((long)voff + soff) / 8;
  q[0] = v.x; q[1] = v.y;
}
inline void row_st1(const RowBuf& r, int voff, int soff, double v) {
  if (!row_in(r, voff, soff, 8)) return;
  const_cast<double*>(r.p)[((long)voff + soff) / 8] = v;
}
#else
typedef unsigned int rpde_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int rpde_u32x2 __attribute__((ext_vector_type(2)));
struct RowBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ RowBuf row_buf(const double* p, long bytes) {
  // dword 3: DATA_FORMAT = 32-bit (the raw-buffer word of gfx9 / CDNA); stride 0; num_records = bytes
  return RowBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, (int)bytes, 0x00020000)};
}
__device__ __forceinline__ dbl2 row_ld2(const RowBuf& r, int voff, int soff) {
  return __builtin_bit_cast(dbl2, __builtin_amdgcn_raw_buffer_load_b128(r.r, voff, soff, 0));
}
__device__ __forceinline__ double row_ld1(const RowBuf& r, int voff, int soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r.r, voff, soff, 0));
}
__device__ __forceinline__ void row_st2(const RowBuf& r, int voff, int soff, dbl2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rpde_u32x4, v), r.r, voff, soff, 0);
}
__device__ __forceinline__ void row_st1(const RowBuf& r, int voff, int soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(rpde_u32x2, v), r.r, voff, soff, 0);
}
#endif

// A chunk-major table (16 T doubles, entry i of thread t at [i * T + t]; rhs_line.h chunk_major16) read through a buffer descriptor:
// ONE per-thread byte offset (8 t) serves all sixteen entries of all tables of a phase, the entry is a uniform offset in a scalar
// register.  A flat load spends a 64-bit vector add per entry beyond the 4 KB an immediate offset reaches (round 6: a fifth of the
// vector instructions of the banded whole-line kernels was this address arithmetic).
struct ChunkTab { RowBuf rb; };
#ifdef RPDE_EMU
inline ChunkTab chunk_tab(const double* p, int T) { return ChunkTab{row_buf(p, 8L * 16 * T)}; }
inline double chunk_ld(const ChunkTab& t, int tid, int i, int T) { return row_ld1(t.rb, 8 * tid, 8 * i * T); }
#else
__device__ __forceinline__ ChunkTab chunk_tab(const double* p, int T) { return ChunkTab{row_buf(p, 8L * 16 * T)}; }
__device__ __forceinline__ double chunk_ld(const ChunkTab& t, int tid, int i, int T) { return row_ld1(t.rb, 8 * tid, 8 * i * T); }
#endif

enum OpCode : int {
  OP_END = 0,
  OP_LOAD,     // d[k] = (acc ? d[k] : 0) + s0 * A[line][map(k)]   k < n (zero tail if !acc); acc = 2: d[k] *= s0 * A; i0 = 1: parity map;
               // b = 1 (OP_LOAD / OP_LOADX): the NEXT op is a plain OP_LOAD executed together with this one (both lines' loads in flight at once)
               // i0 = 2: interleaved complex line times i*kappa (kappa = pair index): d[2j] (+)= -s0 j Im A_j, d[2j+1] (+)= s0 j Re A_j
  OP_LOADX,    // d[k] = (acc ? d[k] : 0) + s0 * ([line<i1] A[line][map(k)] + [line>=2] tab[line-2] A[line-2][map(k)]); i0 = half > 0: parity map (unpaired form only)
  OP_STORE,    // A[line][map(k)] = s0 * a[k]   k < n ; i0 = 1: parity de-interleave, half = i1;
               // acc = 1: NaN guard -- a stored NaN raises *Program::nanflag (Integrate::exit, navier.rs:482-489)
  OP_STEN,     // d[k] = [k<n-2] a[k] + [k>=2] tab[k-2] * a[k-2]              k < n  (n = ortho length);
               // acc = 1 (d != a only): d[k] = s1 * d[k] + s0 * (that)
  OP_MV3,      // d[k] = t0[k] a[k] + t1[k] a[k+2] + t2[k] a[k+4]             k < n  (tables tab, tab+1, tab+2)
  OP_CDIFF,    // d = s0 * d/dx of the Chebyshev series a (length n)
  OP_REC1,     // first-order stride-2 recurrence, x_k = p_k b_k + q_k x_{k-2 dir}; p = tab (-1: ones), q = i0, dir = i1
  OP_REC2,     // descending second-order: x_k = p_k b_k + q_k x_{k+2} + r_k x_{k+4}; p = tab, q = i0, r = i1; slot b (the last one) = scratch
  OP_DCT,      // slot d (scratch d+1): x <- DCT-I(x * pre) * post ; n = N+1; direct path: pre = table tab (-1 none), post = table i0;
               // FFT path: tab / i0 >= 0 = standard backward pre / forward post scaling on, a = first zeroed coefficient, s1 = 1/N
  OP_MUL,      // d[k] = (acc ? d[k] : 0) + s0 * a[k] * b[k]
  OP_AXPBY,    // d[k] = s0 * a[k] + s1 * b[k]
  OP_ZERO,     // d[k] = 0 for i0 <= k < i1
  OP_TABDIV,   // d[k] = a[k] / tab[k >> i0]   (i0 = 1 for interleaved complex lines)
  OP_RFFT_F,   // slot d: real (n = nx) -> interleaved complex (nx/2+1), unnormalised
  OP_RFFT_B,   // slot d: interleaved complex (nx/2+1) -> real (n = nx), scaled by 1/nx
  OP_CIK,      // complex line: d = (i k s0)^i0 * a  for k < n (complex count)
  OP_PUSH,     // per-thread register stash <- a[k]   (one stash per program run; saves an LDS slot)
  OP_POPAXPY,  // d[k] = s0 * d[k] + s1 * stash[k]    k < n
};

// kernel variants: which register-hungry features a program needs (each is compiled out of the
// others so that every variant fits 128 VGPRs without scratch traffic worth mentioning)
constexpr int kVarLight = 0, kVarRec2 = 1, kVarStash = 2, kVarAll = 3;

struct ArrayRef {
  double* p;
  long ld;    // doubles between consecutive lines
  long coff;  // doubles added per component (blockIdx.y)
  int es;     // element stride in doubles (2 for one component of an interleaved complex line)
  int pad;
};

struct Op {
  int code, d, a, b;
  int arr, n, tab, i0;
  int i1, acc;
  long tabld;  // per-line table stride (0: same table for every line)
  double s0, s1;
};

constexpr int kMaxOps = 40;
constexpr int kMaxArr = 16;
constexpr int kMaxTab = 24;

struct Program {
  int nops;
  int nslots;
  int slot_len;   // doubles per slot
  int nlines;     // grid.x (local lines of this rank)
  int line0;      // global index of local line 0 (pencil-sharded runs; 0 otherwise)
  int ncomp;      // grid.y
  int fft_n;      // complex FFT length used by OP_DCT / OP_RFFT (0: no plan of that length: Bluestein, or the direct O(n^2) DCT)
  int blu_m;      // > 0: lines of this length transform by Bluestein's chirp-z algorithm through two power-of-two FFTs of length
                  // blu_m (tw = their twiddles, tw2 = the chirp and filter tables of hostmath.cc bluestein_*_tables)
  int tw;         // table index of the FFT twiddles W_N (N complex: cos, -sin)
  int tw2;        // table index of the split twiddles (cos, sin)(pi k / N) resp. (2 pi k / nx)
  int* nanflag;   // device flag raised by guarded stores (OP_STORE with acc = 1); may be null
  long long* trace;   // diagnostics (tools/trace_ops.py): kTraceStride words per workgroup (platform.h RPDE_MARK); normally null
  Op ops[kMaxOps];
  ArrayRef arr[kMaxArr];
  const double* tabs[kMaxTab];
};

// ---------------------------------------------------------------------------------------------
// kernel configuration (compile time): T threads, EPT elements per thread
template <int T_, int EPT_, int FMIN_, int FMAX_, int RM_, bool CHEB_ = (T_ <= 512)>
struct LineCfg {
  static constexpr int T = T_;
  static constexpr int EPT = EPT_;
  static constexpr int RM = RM_;                       // main FFT radix (8 or 16)
  static constexpr int ZPT = (FMAX_ + T_ - 1) / T_;    // complex FFT points per thread
  static constexpr int FMIN = FMIN_;                   // smallest / largest complex FFT length
  static constexpr int FMAX = FMAX_;                   // instantiated in this configuration
  static constexpr int C = (EPT_ + 1) & ~1;           // scan chunk per thread (even)
  static constexpr int G = (T_ + 15) / 16;             // scan groups
  static constexpr int kMaxSlotLen = T_ * EPT_;
  // T = 1024 (16 waves), EPT = 18 is the long-Fourier-line configuration: one 139 KB work area per
  // workgroup, so no DCT (needs two slots) and no banded scans (a Fourier axis has none); T = 1024, EPT = 10 (CHEB_ = true)
  // is the configuration of Chebyshev lines of 2049 .. 4096 points other than 2^k + 1 (Bluestein, M = 8192: two slots of 70 KB)
  static constexpr bool kCheb = CHEB_;
  static constexpr int kCarryLen = 2 * ((T_ + 63) / 64) * 6 + 4;  // doubles: wave totals of a scan
};

// slots are slot_len apart; the last one is padded to T*EPT doubles so that unguarded reads of
// k = tid + q T < T*EPT (+4 for the stencil taps) stay inside the allocation
RPDE_HD inline size_t line_lds_doubles(int nslots, int slot_len, int t_ept, int carry_len) {
  return (size_t)nslots * slot_len + (t_ept - slot_len) + 8 + carry_len;
}

// ---------------------------------------------------------------------------------------------
// small DFTs in registers (decimation in time, compile-time twiddles)
template <int R>
struct SmallDft;
template <>
struct SmallDft<1> {
  static RPDE_DEV void run(double*, double*) {}
};
template <>
struct SmallDft<2> {
  static RPDE_DEV void run(double* re, double* im) {
    double ar = re[0], ai = im[0], br = re[1], bi = im[1];
    re[0] = ar + br; im[0] = ai + bi;
    re[1] = ar - br; im[1] = ai - bi;
  }
};
template <>
struct SmallDft<4> {
  static RPDE_DEV void run(double* re, double* im) {
    double t0r = re[0] + re[2], t0i = im[0] + im[2];
    double t1r = re[0] - re[2], t1i = im[0] - im[2];
    double t2r = re[1] + re[3], t2i = im[1] + im[3];
    double t3r = re[1] - re[3], t3i = im[1] - im[3];
    // forward transform: multiply t3 by -i  -> (t3i, -t3r)
    re[0] = t0r + t2r; im[0] = t0i + t2i;
    re[2] = t0r - t2r; im[2] = t0i - t2i;
    re[1] = t1r + t3i; im[1] = t1i - t3r;
    re[3] = t1r - t3i; im[3] = t1i + t3r;
  }
};
// R = 8, 16 as (R/4) x 4 Cooley-Tukey steps with compile-time twiddles
template <int R>
struct SmallDft {
  static_assert(R == 8 || R == 16, "radix");
  static RPDE_DEV void run(double* re, double* im) {
    constexpr int R1 = 4, R2 = R / 4;  // n = n1 * R2 + n2 ; k = k1 + R1 * k2   (n1 < R1, n2 < R2)
    double xr[R], xi[R];
    // column DFTs of length R1 over n1 for each n2
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
      double cr[R1], ci[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) { cr[n1] = re[n1 * R2 + n2]; ci[n1] = im[n1 * R2 + n2]; }
      SmallDft<R1>::run(cr, ci);
#pragma unroll
      for (int k1 = 0; k1 < R1; ++k1) {
        // twiddle W_R^{n2 k1} = exp(-2 pi i n2 k1 / R), from the 16th roots of unity
        constexpr double kC16[16] = {1.0, 0.9238795325112867, 0.7071067811865476, 0.3826834323650898, 0.0, -0.3826834323650898, -0.7071067811865476, -0.9238795325112867, -1.0, -0.9238795325112867, -0.7071067811865476, -0.3826834323650898, 0.0, 0.3826834323650898, 0.7071067811865476, 0.9238795325112867};
        constexpr double kS16[16] = {0.0, -0.3826834323650898, -0.7071067811865476, -0.9238795325112867, -1.0, -0.9238795325112867, -0.7071067811865476, -0.3826834323650898, 0.0, 0.3826834323650898, 0.7071067811865476, 0.9238795325112867, 1.0, 0.9238795325112867, 0.7071067811865476, 0.3826834323650898};
        const double c = kC16[(n2 * k1 * (16 / R)) & 15];
        const double s = kS16[(n2 * k1 * (16 / R)) & 15];
        xr[k1 * R2 + n2] = cr[k1] * c - ci[k1] * s;
        xi[k1 * R2 + n2] = cr[k1] * s + ci[k1] * c;
      }
    }
    // row DFTs of length R2 over n2 for each k1
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      double cr[R2], ci[R2];
#pragma unroll
      for (int n2 = 0; n2 < R2; ++n2) { cr[n2] = xr[k1 * R2 + n2]; ci[n2] = xi[k1 * R2 + n2]; }
      SmallDft<R2>::run(cr, ci);
#pragma unroll
      for (int k2 = 0; k2 < R2; ++k2) { re[k1 + R1 * k2] = cr[k2]; im[k1 + R1 * k2] = ci[k2]; }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// In-LDS complex FFT (Stockham autosort, forward sign).  The work area holds interleaved complex
// numbers at the PADDED index pidx(i) = i + (i >> 4): one 16-byte gap every 16 elements makes
// both the strided reads (i = j + t N/R) and the scattered writes (i = 16 j + t, ...) of every
// pass fall on distinct LDS banks within a ds_read/write_b128 lane group.
RPDE_HD inline int pidx(int i) { return i + (i >> 4); }
RPDE_HD inline int fft_work_doubles(int n) { return 2 * (n + (n >> 4)); }

struct LdsSrc { static constexpr bool kLds = true; };   // butterfly inputs come from the work area

// first-pass source of the DCT: packs z_i = e_{2i} + i e_{2i+1} of the even extension of the real
// line x on the fly (optionally with the composite->ortho stencil and the pre-scaling), so the
// separate pack step (one LDS round trip, two barriers) disappears
// STEN: 0 = plain line, 1 = composite->ortho stencil with the table `low`, 2 = the Dirichlet stencil
// (low = -1 everywhere: c_k = a_k - a_{k-2}) without a table fetch
template <int STEN>
struct DctSrc {
  static constexpr bool kLds = false;
  clds_t x; int N; bool pre; tab_t low;
  // lo_edge / hi_edge: this element may touch the ends of the line (m < 2, m >= N-1); elsewhere the
  // stencil needs no selects.  `odd` = parity of m.  All flags are compile-time constants after
  // unrolling.  pre: the backward pre-scaling (-1)^m / 2 (ends: 1) evaluated arithmetically --
  // fetching it from a table cost 16 % of a DCT program (profiles/README.md).
  RPDE_DEV double val(int m, int n, bool lo_edge, bool hi_edge, bool odd) const {
    double v;
    if constexpr (STEN != 0) {
      double a0 = x[m];
      if (hi_edge) a0 = (m < n - 1) ? a0 : 0.0;
      double t2;
      if (lo_edge) {
        const int m2 = m >= 2 ? m - 2 : 0;
        const double a2 = x[m2], l2 = (STEN == 2) ? -1.0 : low[m2];
        t2 = (m >= 2) ? l2 * a2 : 0.0;
      } else {
        t2 = (STEN == 2) ? -x[m - 2] : low[m - 2] * x[m - 2];
      }
      v = a0 + t2;
    } else {
      v = x[m];
    }
    if (pre) {
      double f = odd ? -0.5 : 0.5;
      if (!odd && ((lo_edge && m == 0) || (hi_edge && m == n))) f = 1.0;   // n is even: the ends are even
      v *= f;
    }
    return v;
  }
  // element i of the packed even extension, n = FFT length = DCT N (compile-time at the call
  // site); hi: i >= n/2, where the extension runs backwards (m0 = 2n - 2i, m1 = m0 - 1)
  RPDE_DEV dbl2 get(int i, int n, bool hi, bool lo_edge, bool hi_edge) const {
    const int m0 = hi ? 2 * n - 2 * i : 2 * i;
    const int m1 = hi ? m0 - 1 : m0 + 1;
    return dbl2{val(m0, n, lo_edge, hi_edge, false), val(m1, n, lo_edge, hi_edge, true)};
  }
};

template <class Cfg, int N, int R, int LGNS, class Src = LdsSrc>
RPDE_DEVN void fft_pass(Blk& blk, lds_t w, tab_t tw, const Src src = Src()) {
  constexpr int T = Cfg::T;
  constexpr int NB = N / R;                        // butterflies
  constexpr int Q = (NB + T - 1) / T;              // butterflies per thread
  constexpr int Ns = 1 << LGNS;
  constexpr int LGR = (R == 16) ? 4 : (R == 8) ? 3 : (R == 4) ? 2 : 1;
  static_assert(Q * R <= 16, "FFT too large for this kernel configuration");
  // with these conditions the padded index of element t of a butterfly is the padded index of
  // element 0 plus a compile-time constant, so the accesses become base + immediate offset
  constexpr bool kStaticRead = (NB % 16 == 0);
  constexpr bool kStaticWrite = (Ns >= 16) || ((Ns * R) % 16 == 0);
  lds2_t w2 = (lds2_t)w;
  RPDE_TLS(blk, double, xr, Q * R);
  RPDE_TLS(blk, double, xi, Q * R);
  RPDE_TLS(blk, double, tw1, 2 * Q);   // table twiddle of each butterfly, fetched ahead of the barrier
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = tid + q * T;
      if (j < NB) {
        if constexpr (LGNS > 0) {
          constexpr int tstep = N / (Ns * R);
          const int k = j & (Ns - 1);
          RPDE_T(tw1)[2 * q] = tw[2 * (k * tstep)];
          RPDE_T(tw1)[2 * q + 1] = tw[2 * (k * tstep) + 1];
        }
        const int base = pidx(j);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          dbl2 v;
          if constexpr (Src::kLds) {
            const int p = kStaticRead ? base + t * NB + ((t * NB) >> 4) : pidx(j + t * NB);
            v = w2[p];
          } else {
            // butterfly input t covers i in [t NB, (t+1) NB): one half of the extension; only the
            // first and last inputs reach m < 2 (i = 0: m = 0, 1; i = N-1: m = 2, 1) and only the
            // two middle ones reach m >= N-1
            v = src.get(j + t * NB, N, t * NB >= N / 2, t == 0 || t == R - 1, t == R / 2 || t == R / 2 - 1);
          }
          RPDE_T(xr)[q * R + t] = v.x;
          RPDE_T(xi)[q * R + t] = v.y;
        }
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = tid + q * T;
      if (j < NB) {
        const int k = j & (Ns - 1);
        double* pr = RPDE_T(xr) + q * R;
        double* pi = RPDE_T(xi) + q * R;
        if constexpr (LGNS > 0) {
          // twiddles W^(t k tstep), t = 1..R-1, as powers of the table entry for t = 1
          double wc[R], ws[R];
          wc[1] = RPDE_T(tw1)[2 * q];
          ws[1] = RPDE_T(tw1)[2 * q + 1];
#pragma unroll
          for (int t = 2; t < R; ++t) {
            if (t % 2 == 0) {
              const double c = wc[t / 2], s = ws[t / 2];
              wc[t] = c * c - s * s;
              ws[t] = 2.0 * c * s;
            } else {
              wc[t] = wc[t - 1] * wc[1] - ws[t - 1] * ws[1];
              ws[t] = wc[t - 1] * ws[1] + ws[t - 1] * wc[1];
            }
          }
#pragma unroll
          for (int t = 1; t < R; ++t) {
            const double ar = pr[t], ai = pi[t];
            pr[t] = ar * wc[t] - ai * ws[t];
            pi[t] = ar * ws[t] + ai * wc[t];
          }
        }
        SmallDft<R>::run(pr, pi);
        const int j0 = ((j >> LGNS) << (LGNS + LGR)) + k;
        const int b0 = pidx(j0);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const int p = kStaticWrite ? b0 + t * Ns + ((t * Ns) >> 4) : pidx(j0 + t * Ns);
          w2[p] = dbl2{pr[t], pi[t]};
        }
      }
    }
  }
  RPDE_SYNC(blk);
}

struct NoHook { RPDE_DEV void operator()() const {} };

// hook(): called between the last two passes -- the caller's chance to put global loads in flight that
// its next phase needs (the DCT's split twiddles: an L2 round trip otherwise exposed after the last barrier)
template <class Cfg, int N, class Src = LdsSrc, class Hook = NoHook>
RPDE_DEVN void fft_lds(Blk& blk, lds_t w, tab_t tw, const Src src = Src(), const Hook hook = Hook()) {
  // radix schedule: one of {2,4,8} first (if log2 N is not a multiple of 4), then 16s
  constexpr int L = (N >= 8192) ? 13 : (N >= 4096) ? 12 : (N >= 2048) ? 11 : (N >= 1024) ? 10
                  : (N >= 512) ? 9 : (N >= 256) ? 8 : (N >= 128) ? 7 : (N >= 64) ? 6
                  : (N >= 32) ? 5 : (N >= 16) ? 4 : (N >= 8) ? 3 : (N >= 4) ? 2 : 1;
  static_assert((1 << L) == N, "power of two");
  constexpr int RM = Cfg::RM;
  constexpr int LG = (RM == 16) ? 4 : 3;
  constexpr int nm = L / LG;
  constexpr int rem = L % LG;
  if constexpr (nm == 0) hook();
  if constexpr (rem == 1) fft_pass<Cfg, N, 2, 0, Src>(blk, w, tw, src);
  if constexpr (rem == 2) fft_pass<Cfg, N, 4, 0, Src>(blk, w, tw, src);
  if constexpr (rem == 3) fft_pass<Cfg, N, 8, 0, Src>(blk, w, tw, src);
  if constexpr (nm >= 1) {
    if constexpr (nm == 1) hook();
    if constexpr (rem == 0) fft_pass<Cfg, N, RM, 0, Src>(blk, w, tw, src);
    else fft_pass<Cfg, N, RM, rem>(blk, w, tw);
  }
  if constexpr (nm >= 2) { if constexpr (nm == 2) hook(); fft_pass<Cfg, N, RM, rem + LG>(blk, w, tw); }
  if constexpr (nm >= 3) { if constexpr (nm == 3) hook(); fft_pass<Cfg, N, RM, rem + 2 * LG>(blk, w, tw); }
  if constexpr (nm >= 4) { if constexpr (nm == 4) hook(); fft_pass<Cfg, N, RM, rem + 3 * LG>(blk, w, tw); }
}

template <class Cfg, class Src = LdsSrc, class Hook = NoHook>
RPDE_DEV void fft_dispatch(Blk& blk, lds_t w, int n, tab_t tw, const Src src = Src(), const Hook hook = Hook()) {
  switch (n) {
#define RPDE_FFT_CASE(NN) \
  case NN: if constexpr (NN >= Cfg::FMIN && NN <= Cfg::FMAX) fft_lds<Cfg, NN, Src, Hook>(blk, w, tw, src, hook); break;
    RPDE_FFT_CASE(2) RPDE_FFT_CASE(4) RPDE_FFT_CASE(8) RPDE_FFT_CASE(16) RPDE_FFT_CASE(32)
    RPDE_FFT_CASE(64) RPDE_FFT_CASE(128) RPDE_FFT_CASE(256) RPDE_FFT_CASE(512)
    RPDE_FFT_CASE(1024) RPDE_FFT_CASE(2048) RPDE_FFT_CASE(4096) RPDE_FFT_CASE(8192)
#undef RPDE_FFT_CASE
    default: break;
  }
}

// ---------------------------------------------------------------------------------------------
// DCT-I of the n = N+1 reals in slot x (work area = x .. x + fft_work_doubles(N), i.e. slot d, d+1):
//   E_k = x_0 + (-1)^k x_N + 2 sum_{j=1}^{N-1} x_j cos(pi j k / N)
// through an N-point complex FFT of the even extension (packed two reals per complex).
template <class Cfg>
RPDE_DEVN void dct1_lds(Blk& blk, lds_t x, int N, bool pre, bool post, int cut, double inv_n, tab_t tw,
                        tab_t tw2, int sten, tab_t low, gmem_t gdst, int ges, double gscale, int gn) {
  constexpr int T = Cfg::T;
  // FFT with the pack step fused into the reads of its first pass (the work area overlaps x: every
  // pass reads everything before the barrier that precedes its writes)
  constexpr int QH = Cfg::FMAX / 2 / T + 1;
  static_assert(2 * QH <= Cfg::EPT + 2, "pair split needs 2 QH registers");
  const int H = N >> 1;
  // the split twiddles (cos, sin)(pi k / N) of this thread's pairs go in flight before the last FFT pass
  RPDE_TLS(blk, double, cs, 2 * QH);
  auto fetch_split = [&]() {
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QH; ++q) {
        const int k = tid + q * T, kk = (k <= H) ? k : 0;
        RPDE_T(cs)[2 * q] = tw2[2 * kk];
        RPDE_T(cs)[2 * q + 1] = tw2[2 * kk + 1];
      }
    }
  };
  if (sten == 2) fft_dispatch<Cfg, DctSrc<2>>(blk, x, N, tw, DctSrc<2>{x, N, pre, low}, fetch_split);
  else if (sten == 1) fft_dispatch<Cfg, DctSrc<1>>(blk, x, N, tw, DctSrc<1>{x, N, pre, low}, fetch_split);
  else fft_dispatch<Cfg, DctSrc<0>>(blk, x, N, tw, DctSrc<0>{x, N, pre, low}, fetch_split);
  // split: E_k = A + B, E_{N-k} = A - B with A = (Zr_k + Zr_{N-k})/2,
  // B = (c_k (Zi_k + Zi_{N-k}) - s_k (Zr_k - Zr_{N-k}))/2: one thread per pair (k, N-k)
  //
  // Fast form for the two largest lengths of a configuration (compile-time N; the whole line stored, contiguous):
  // thread t owns the pairs k = t + q T, q < H / T -- every LDS and global address is one per-thread base plus a
  // constant (pidx(t + q T) = pidx(t) + q (T + T/16); pidx(N - q T - t) = pidx(-t) + (N - q T) (1 + 1/16), T a
  // multiple of 16) -- and thread 0 also owns k = H.  The generic form below spends 280 vector instructions per
  // thread on 60 of arithmetic, and with two waves per SIMD every instruction is 8 clocks of the line's latency.
  auto split_fast = [&](auto NC, auto STORE) {
    constexpr int NN = decltype(NC)::value;
    constexpr bool kStore = decltype(STORE)::value;
    constexpr int HH = NN / 2, QF = HH / T;
    static_assert(HH % T == 0 && T % 16 == 0 && QF >= 1 && QF < QH, "split_fast: line length / thread count");
    lds2_t X = (lds2_t)x;
    RPDE_TLS(blk, double, ek, QF + 1);
    RPDE_TLS(blk, double, en, QF + 1);
    RPDE_PHASE(blk, tid) {
      const int pa = tid + (tid >> 4);          // pidx(tid)
      const int pb = -tid + ((-tid) >> 4);      // pidx(-tid) (arithmetic shift: floor)
      const double fs = (tid & 1) ? -inv_n : inv_n;   // T is even: k and N - k have the parity of tid
#pragma unroll
      for (int q = 0; q <= QF; ++q) {
        if (q == QF && tid != 0) break;         // k = H: thread 0 only
        const int k = tid + q * T;
        const dbl2 za = X[pa + q * (T + T / 16)];
        const int ib = (q == 0) ? (tid == 0 ? 0 : pb + NN + NN / 16) : pb + (NN - q * T) + (NN - q * T) / 16;
        const dbl2 zb = X[ib];
        const double c = RPDE_T(cs)[2 * q], sn = RPDE_T(cs)[2 * q + 1];
        const double A = 0.5 * (za.x + zb.x), B = 0.5 * (c * (za.y + zb.y) - sn * (za.x - zb.x));
        double e0 = A + B, e1 = A - B;
        if (post) {   // forward scaling (-1)^k / N (ends: half), zero from `cut` on
          const double f = (q == 0 && tid == 0) ? 0.5 * fs : fs;
          e0 = (k < cut) ? e0 * f : 0.0;
          e1 = (NN - k < cut) ? e1 * f : 0.0;
        }
        if constexpr (kStore) {
          gdst[k] = gscale * e0;
          gdst[NN - k] = gscale * e1;
        } else {
          RPDE_T(ek)[q] = e0;
          RPDE_T(en)[q] = e1;
        }
      }
    }
    RPDE_SYNC(blk);
    if constexpr (!kStore) {
      RPDE_PHASE(blk, tid) {
#pragma unroll
        for (int q = 0; q <= QF; ++q) {
          if (q == QF && tid != 0) break;
          const int k = tid + q * T;
          x[k] = RPDE_T(ek)[q];
          x[NN - k] = RPDE_T(en)[q];
        }
      }
      RPDE_SYNC(blk);
    }
  };
  if constexpr (Cfg::FMAX / 4 >= T) {
    const bool whole = gdst == nullptr || (ges == 1 && gn > N);
    if (whole && (N == Cfg::FMAX || N == Cfg::FMAX / 2)) {
      using std::integral_constant;
      if (N == Cfg::FMAX) {
        if (gdst) split_fast(integral_constant<int, Cfg::FMAX>{}, integral_constant<bool, true>{});
        else split_fast(integral_constant<int, Cfg::FMAX>{}, integral_constant<bool, false>{});
      } else {
        if (gdst) split_fast(integral_constant<int, Cfg::FMAX / 2>{}, integral_constant<bool, true>{});
        else split_fast(integral_constant<int, Cfg::FMAX / 2>{}, integral_constant<bool, false>{});
      }
      return;
    }
  }
  {
    RPDE_TLS(blk, double, e, 2 * QH);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QH; ++q) {
        const int k = tid + q * T;
        if (k <= H) {
          const int kn = N - k;
          const dbl2 za = ((lds2_t)x)[pidx(k)];
          const dbl2 zb = ((lds2_t)x)[pidx((k == 0) ? 0 : kn)];
          const double ar = za.x, ai = za.y, br = zb.x, bi = zb.y;
          const double c = RPDE_T(cs)[2 * q], s = RPDE_T(cs)[2 * q + 1];
          const double A = 0.5 * (ar + br), B = 0.5 * (c * (ai + bi) - s * (ar - br));
          double ek = A + B, en = A - B;
          if (post) {   // forward scaling (-1)^k / N (ends: half), zero from `cut` on; N even: k, N-k share the sign
            double f = (k & 1) ? -inv_n : inv_n;
            if (k == 0) f *= 0.5;
            ek = (k < cut) ? ek * f : 0.0;
            en = (kn < cut) ? en * f : 0.0;
          }
          RPDE_T(e)[2 * q] = ek;
          RPDE_T(e)[2 * q + 1] = en;
          if (gdst) {   // fused OP_STORE
            if (k < gn) gdst[(long)k * ges] = gscale * ek;
            if (kn < gn) gdst[(long)kn * ges] = gscale * en;
          }
        }
      }
    }
    RPDE_SYNC(blk);
    if (!gdst) {
      RPDE_PHASE(blk, tid) {
#pragma unroll
        for (int q = 0; q < QH; ++q) {
          const int k = tid + q * T;
          if (k <= H) { x[k] = RPDE_T(e)[2 * q]; x[N - k] = RPDE_T(e)[2 * q + 1]; }
        }
      }
      RPDE_SYNC(blk);
    }
  }
}

// direct O(n^2) DCT-I for line lengths without an FFT plan (small / odd sizes); costab[m] = cos(pi m / N), m < 2N
template <class Cfg>
RPDE_DEVN void dct1_direct(Blk& blk, lds_t x, int N, tab_t pre, tab_t post, tab_t costab) {
  constexpr int T = Cfg::T;
  RPDE_TLS(blk, double, e, Cfg::EPT);
  if (pre) {
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < Cfg::EPT; ++q) { const int k = tid + q * T; if (k <= N) x[k] *= pre[k]; }
    }
    RPDE_SYNC(blk);
  }
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Cfg::EPT; ++q) {
      const int k = tid + q * T;
      if (k <= N) {
        double acc = x[0] + ((k & 1) ? -x[N] : x[N]);
        for (int j = 1; j < N; ++j) acc += 2.0 * x[j] * costab[(int)(((long)j * k) % (2 * N))];
        if (post) acc *= post[k];
        RPDE_T(e)[q] = acc;
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Cfg::EPT; ++q) { const int k = tid + q * T; if (k <= N) x[k] = RPDE_T(e)[q]; }
  }
  RPDE_SYNC(blk);
}

// real FFT, nx reals -> nx/2+1 interleaved complex, unnormalised (forward) ; tw2[k] = (cos, sin)(2 pi k / nx)
template <class Cfg>
RPDE_DEVN void rfft_forward_lds(Blk& blk, lds_t x, int nx, tab_t tw, tab_t tw2) {
  constexpr int T = Cfg::T;
  // per-thread trip counts from the largest transform of the configuration (M <= FMAX complex
  // points, nx = 2 M reals), not from EPT: the 1024-thread configuration has EPT = 18 but M / T = 8
  constexpr int QM = (Cfg::FMAX / T + 1 < Cfg::EPT) ? Cfg::FMAX / T + 1 : Cfg::EPT;          // k <= M
  constexpr int QE = (2 * Cfg::FMAX / T + 1 < Cfg::EPT) ? 2 * Cfg::FMAX / T + 1 : Cfg::EPT;  // e < nx + 2
  const int M = nx / 2;
  {  // z_j = x_{2j} + i x_{2j+1}: move to the padded work layout
    RPDE_TLS(blk, double, v, QE);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QE; ++q) { const int e = tid + q * T; if (e < nx) RPDE_T(v)[q] = x[e]; }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QE; ++q) {
        const int e = tid + q * T;
        if (e < nx) x[2 * pidx(e >> 1) + (e & 1)] = RPDE_T(v)[q];
      }
    }
    RPDE_SYNC(blk);
  }
  fft_dispatch<Cfg>(blk, x, M, tw);
  RPDE_TLS(blk, double, yr, QM);
  RPDE_TLS(blk, double, yi, QM);
  RPDE_PHASE(blk, tid) {
    double tc[QM], ts[QM];   // split twiddles: all fetched before the first use
#pragma unroll
    for (int q = 0; q < QM; ++q) { const int kc = min(tid + q * T, M); tc[q] = tw2[2 * kc]; ts[q] = tw2[2 * kc + 1]; }
#pragma unroll
    for (int q = 0; q < QM; ++q) { RPDE_PIN(tc[q]); RPDE_PIN(ts[q]); }
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      const int k = tid + q * T;
      if (k <= M) {
        const int ka = 2 * pidx((k == M) ? 0 : k);
        const int kb = 2 * pidx((k == 0) ? 0 : M - k);
        const double ar = x[ka], ai = x[ka + 1];
        const double br = x[kb], bi = x[kb + 1];
        const double c = tc[q], s = ts[q];
        const double sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
        RPDE_T(yr)[q] = 0.5 * (sr + c * di - s * dr);
        RPDE_T(yi)[q] = 0.5 * (si - c * dr - s * di);
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      const int k = tid + q * T;
      if (k <= M) { x[2 * k] = RPDE_T(yr)[q]; x[2 * k + 1] = RPDE_T(yi)[q]; }
    }
  }
  RPDE_SYNC(blk);
}

// inverse: nx/2+1 interleaved complex -> nx reals, scaled by 1/nx (imaginary parts of k=0, nx/2 ignored)
template <class Cfg>
RPDE_DEVN void rfft_backward_lds(Blk& blk, lds_t x, int nx, tab_t tw, tab_t tw2) {
  constexpr int T = Cfg::T;
  constexpr int QM = (Cfg::FMAX / T + 1 < Cfg::EPT) ? Cfg::FMAX / T + 1 : Cfg::EPT;
  constexpr int QE = (2 * Cfg::FMAX / T + 1 < Cfg::EPT) ? 2 * Cfg::FMAX / T + 1 : Cfg::EPT;
  const int M = nx / 2;
  {
    RPDE_TLS(blk, double, zr, QM);
    RPDE_TLS(blk, double, zi, QM);
    RPDE_PHASE(blk, tid) {
      double tc[QM], ts[QM];   // split twiddles: all fetched before the first use
#pragma unroll
      for (int q = 0; q < QM; ++q) { const int kc = min(tid + q * T, M); tc[q] = tw2[2 * kc]; ts[q] = tw2[2 * kc + 1]; }
#pragma unroll
      for (int q = 0; q < QM; ++q) { RPDE_PIN(tc[q]); RPDE_PIN(ts[q]); }
#pragma unroll
      for (int q = 0; q < QM; ++q) {
        const int k = tid + q * T;
        if (k < M) {
          const int kb = M - k;
          double ar = x[2 * k], ai = x[2 * k + 1];
          double br = x[2 * kb], bi = -x[2 * kb + 1];  // conj X_{M-k}
          if (k == 0) { ai = 0.0; bi = 0.0; }
          const double c = tc[q], s = ts[q];  // conj(W^k) = c + i s
          const double sr = ar + br, si = ai + bi, dr = ar - br, di = ai - bi;
          // Z_k = ( S + i (c + i s) D ) / 2 ;  store conj(Z_k) for the conjugate-FFT inverse
          const double er = sr + (-(c * di) - s * dr);
          const double ei = si + (c * dr - s * di);
          RPDE_T(zr)[q] = 0.5 * er;
          RPDE_T(zi)[q] = -0.5 * ei;
        }
      }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QM; ++q) {
        const int k = tid + q * T;
        if (k < M) { const int p = 2 * pidx(k); x[p] = RPDE_T(zr)[q]; x[p + 1] = RPDE_T(zi)[q]; }
      }
    }
    RPDE_SYNC(blk);
  }
  fft_dispatch<Cfg>(blk, x, M, tw);
  const double sc = 1.0 / (double)M;
  RPDE_TLS(blk, double, v, QE);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QE; ++q) {
      const int e = tid + q * T;
      if (e < nx) RPDE_T(v)[q] = x[2 * pidx(e >> 1) + (e & 1)] * ((e & 1) ? -sc : sc);
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QE; ++q) { const int e = tid + q * T; if (e < nx) x[e] = RPDE_T(v)[q]; }
  }
  RPDE_SYNC(blk);
}

// ---------------------------------------------------------------------------------------------
// Transforms of ARBITRARY length (Bluestein / chirp-z): the reference transforms any n (rustdct / rustfft / realfft under
// funspace, call sites src/field.rs:103-110; its benches run Chebyshev n = 128, 264, 512, 1024: benches/benchmark_navier.rs:6-7,
// benchmark_transform.rs:6).  With 2 j k = j^2 + k^2 - (k - j)^2 a length-L transform is a convolution with a chirp:
//   sum_j x_j w^(2 j k) = w^(k^2) * sum_j (x_j w^(j^2)) w^(-(k - j)^2),
// evaluated as a circular convolution of power-of-two length M (>= number of distinct lags) with the power-of-two FFT of this
// file: a = x * chirp (zero padded) -> FFT_M -> times the tabulated spectrum of the filter / M, conjugated -> FFT_M -> conjugate
// (= inverse FFT) -> times the chirp.  The work area W = M complex numbers at the padded index pidx() starts at the line itself.
//
// the two FFTs and the point-wise product in between; on return W holds V with (a (*) b)_k = conj(V_k)
template <class Cfg>
RPDE_DEVN void bluestein_convolve(Blk& blk, lds_t x, int M, tab_t tw, tab_t hs) {
  constexpr int T = Cfg::T, ZP = Cfg::ZPT;
  lds2_t W = (lds2_t)x;
  fft_dispatch<Cfg>(blk, x, M, tw);
  RPDE_PHASE(blk, tid) {
    double hr[ZP], hi[ZP];   // filter spectrum: all fetched before the first use
#pragma unroll
    for (int q = 0; q < ZP; ++q) { const int i = min(tid + q * T, M - 1); hr[q] = hs[2 * i]; hi[q] = hs[2 * i + 1]; }
#pragma unroll
    for (int q = 0; q < ZP; ++q) { RPDE_PIN(hr[q]); RPDE_PIN(hi[q]); }
#pragma unroll
    for (int q = 0; q < ZP; ++q) {
      const int i = tid + q * T;
      if (i < M) {
        const dbl2 v = W[pidx(i)];
        W[pidx(i)] = dbl2{v.x * hr[q] - v.y * hi[q], -(v.x * hi[q] + v.y * hr[q])};
      }
    }
  }
  RPDE_SYNC(blk);
  fft_dispatch<Cfg>(blk, x, M, tw);
}
// stage the chirped input (QA values per thread, element i = tid + q T, i < na) into the zero-padded work area
template <class Cfg, int QA>
RPDE_DEV void bluestein_stage(Blk& blk, lds_t x, int na, int M, const double* ar, const double* ai) {
  constexpr int T = Cfg::T, ZP = Cfg::ZPT;
  lds2_t W = (lds2_t)x;
  RPDE_SYNC(blk);   // every thread has read its part of the line
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < ZP; ++q) {
      const int i = tid + q * T;
      dbl2 v = dbl2{0.0, 0.0};
      if (q < QA) { if (i < na) v = dbl2{RPDE_TPK(ar, QA)[q < QA ? q : 0], RPDE_TPK(ai, QA)[q < QA ? q : 0]}; }
      if (i < M) W[pidx(i)] = v;
    }
  }
  RPDE_SYNC(blk);
}

// DCT-I of the N + 1 reals of slot x, any N >= 1:  E_k = x_0 + (-1)^k x_N + 2 sum_{0<j<N} x_j cos(pi j k / N) = Re sum_j g_j x_j w^(2 j k),
// w = exp(i pi / (2 N)); M >= 2 N + 1.  bt = bluestein_dct_tables(N, M); pre / post: scaling tables (may be null) like the direct form
template <class Cfg>
RPDE_DEVN void dct1_bluestein(Blk& blk, lds_t x, int N, int M, tab_t pre, tab_t post, tab_t tw, tab_t bt) {
  constexpr int T = Cfg::T;
  constexpr int QI = Cfg::ZPT / 2 + 1;   // N + 1 <= M / 2 + 1 inputs and outputs
  tab_t chirp = bt, hs = bt + 2 * (N + 1);
  {
    RPDE_TLS(blk, double, ar, QI);
    RPDE_TLS(blk, double, ai, QI);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QI; ++q) {
        const int j = tid + q * T;
        if (j <= N) {
          double v = x[j];
          if (pre) v *= pre[j];
          if (j != 0 && j != N) v *= 2.0;
          RPDE_T(ar)[q] = v * chirp[2 * j];
          RPDE_T(ai)[q] = v * chirp[2 * j + 1];
        }
      }
    }
    bluestein_stage<Cfg, QI>(blk, x, N + 1, M, RPDE_TLS_PTR(ar), RPDE_TLS_PTR(ai));
  }
  bluestein_convolve<Cfg>(blk, x, M, tw, hs);
  RPDE_TLS(blk, double, e, QI);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QI; ++q) {
      const int k = tid + q * T;
      if (k <= N) {
        const dbl2 v = ((lds2_t)x)[pidx(k)];
        double ek = chirp[2 * k] * v.x + chirp[2 * k + 1] * v.y;   // Re (w^(k^2) conj V_k)
        if (post) ek *= post[k];
        RPDE_T(e)[q] = ek;
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QI; ++q) { const int k = tid + q * T; if (k <= N) x[k] = RPDE_T(e)[q]; }
  }
  RPDE_SYNC(blk);
}

// real FFT of any length nx (K = nx / 2, M >= nx + K); bt = bluestein_rfft_tables(nx, M).
//   forward: nx reals -> K + 1 interleaved complex, unnormalised:  X_k = sum_j x_j w^(2 j k), w = exp(-i pi / nx)
//   backward: K + 1 interleaved complex -> nx reals, scaled by 1 / nx:  x_j = Re sum_k g_k X_k u^(2 j k) / nx, u = conj w,
//             g = 1 for k = 0 and 2 k = nx, else 2 (the imaginary parts of those two are ignored, like realfft's c2r)
template <class Cfg, bool FWD>
RPDE_DEVN void rfft_bluestein(Blk& blk, lds_t x, int nx, int M, tab_t tw, tab_t bt) {
  constexpr int T = Cfg::T, ZP = Cfg::ZPT;
  constexpr int QK = ZP / 2 + 1;         // K + 1 <= M / 3 + 1 spectral coefficients
  constexpr int QA = FWD ? ZP : QK, QO = FWD ? QK : ZP;
  const int K = nx / 2;
  tab_t chirp = bt, hs = bt + 2 * nx + (FWD ? 0 : 2 * M);
  {
    RPDE_TLS(blk, double, ar, QA);
    RPDE_TLS(blk, double, ai, QA);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < QA; ++q) {
        const int j = tid + q * T;
        if constexpr (FWD) {
          if (j < nx) {
            const double v = x[j];
            RPDE_T(ar)[q] = v * chirp[2 * j];
            RPDE_T(ai)[q] = -(v * chirp[2 * j + 1]);
          }
        } else {
          if (j <= K) {
            const bool self = (j == 0) || (2 * j == nx);
            const double g = self ? 1.0 : 2.0;
            const double xr = g * x[2 * j], xi = self ? 0.0 : g * x[2 * j + 1];
            const double c = chirp[2 * j], sn = chirp[2 * j + 1];
            RPDE_T(ar)[q] = xr * c - xi * sn;
            RPDE_T(ai)[q] = xr * sn + xi * c;
          }
        }
      }
    }
    bluestein_stage<Cfg, QA>(blk, x, FWD ? nx : K + 1, M, RPDE_TLS_PTR(ar), RPDE_TLS_PTR(ai));
  }
  bluestein_convolve<Cfg>(blk, x, M, tw, hs);
  RPDE_TLS(blk, double, yr, QO);
  RPDE_TLS(blk, double, yi, FWD ? QO : 1);
  const double sc = 1.0 / (double)nx;
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      const int k = tid + q * T;
      if (k < (FWD ? K + 1 : nx)) {
        const dbl2 v = ((lds2_t)x)[pidx(k)];
        const double c = chirp[2 * k], sn = chirp[2 * k + 1];
        if constexpr (FWD) {   // w^(k^2) conj V_k = (c - i s)(Vr - i Vi)
          RPDE_T(yr)[q] = c * v.x - sn * v.y;
          RPDE_T(yi)[q] = -(c * v.y + sn * v.x);
        } else {               // Re (u^(j^2) conj V_j) / nx
          RPDE_T(yr)[q] = sc * (c * v.x + sn * v.y);
        }
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < QO; ++q) {
      const int k = tid + q * T;
      if constexpr (FWD) { if (k <= K) { x[2 * k] = RPDE_T(yr)[q]; x[2 * k + 1] = RPDE_T(yi)[q]; } }
      else { if (k < nx) x[k] = RPDE_T(yr)[q]; }
    }
  }
  RPDE_SYNC(blk);
}

// ---------------------------------------------------------------------------------------------
// chunked scans for stride-2 linear recurrences
//
// Coef concept:   double b(int k)  inhomogeneous term (already multiplied by p_k)
//                 double q(int k)  coefficient of the first predecessor  (k -/+ 2)
//                 double r(int k)  coefficient of the second predecessor (k -/+ 4)   [ORDER 2]
// DIR = +1: ascending (predecessor k-2), DIR = -1: descending (predecessor k+2).
//
// Thread t owns the chunk of C consecutive elements number tau(t) = t (ascending) or T-1-t
// (descending), so that the carry always flows from thread t-1 to thread t.  Phase 1 reduces a
// chunk to an affine map of its inflow state, phase 2 turns the maps into inflow states (prefix
// composition: wave shuffles + one LDS hop on the GPU; a plain loop in the host emulation),
// phase 3 re-runs the chunk with the exact inflow.
template <int ORDER>
struct Affine {            // s -> M s + v   (ORDER 1: scalars m11, v1)
  double m11, m12, m21, m22, v1, v2;
};
template <int ORDER>
RPDE_HD inline Affine<ORDER> affine_identity() { return Affine<ORDER>{1.0, 0.0, 0.0, 1.0, 0.0, 0.0}; }
// apply `first`, then `second`
template <int ORDER>
RPDE_HD inline Affine<ORDER> affine_compose(const Affine<ORDER>& second, const Affine<ORDER>& first) {
  Affine<ORDER> r;
  if constexpr (ORDER == 1) {
    r.m11 = second.m11 * first.m11;
    r.v1 = second.m11 * first.v1 + second.v1;
    r.m12 = r.m21 = 0.0; r.m22 = 1.0; r.v2 = 0.0;
  } else {
    r.m11 = second.m11 * first.m11 + second.m12 * first.m21;
    r.m12 = second.m11 * first.m12 + second.m12 * first.m22;
    r.m21 = second.m21 * first.m11 + second.m22 * first.m21;
    r.m22 = second.m21 * first.m12 + second.m22 * first.m22;
    r.v1 = second.m11 * first.v1 + second.m12 * first.v2 + second.v1;
    r.v2 = second.m21 * first.v1 + second.m22 * first.v2 + second.v2;
  }
  return r;
}

template <int ORDER>
RPDE_HD inline Affine<ORDER> affine_select(bool c, const Affine<ORDER>& a, const Affine<ORDER>& b) {
  Affine<ORDER> r = b;
  r.m11 = c ? a.m11 : b.m11; r.v1 = c ? a.v1 : b.v1;
  if constexpr (ORDER == 2) {
    r.m12 = c ? a.m12 : b.m12; r.m21 = c ? a.m21 : b.m21; r.m22 = c ? a.m22 : b.m22; r.v2 = c ? a.v2 : b.v2;
  }
  return r;
}

#ifndef RPDE_EMU
template <int ORDER>
__device__ __forceinline__ Affine<ORDER> affine_shfl_up(const Affine<ORDER>& a, int off) {
  Affine<ORDER> r = affine_identity<ORDER>();
  r.m11 = __shfl_up(a.m11, off);
  r.v1 = __shfl_up(a.v1, off);
  if constexpr (ORDER == 2) {
    r.m12 = __shfl_up(a.m12, off); r.m21 = __shfl_up(a.m21, off); r.m22 = __shfl_up(a.m22, off);
    r.v2 = __shfl_up(a.v2, off);
  }
  return r;
}
// cross-lane moves of the in-wave scans as DPP (VALU data path; no trip through the LDS crossbar like
// ds_bpermute).  Lanes without a source (row edge, masked row) receive `old` -- the identity of the scan,
// so no select follows.  Controls (CDNA ISA, wave64): row_shr:n = 0x110 + n, wave_shr:1 = 0x138,
// row_bcast:15 = 0x142 (lane 15 of a row to the next row), row_bcast:31 = 0x143 (lane 31 to rows 2, 3)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double x) {
  const long long xb = __double_as_longlong(x), ob = __double_as_longlong(old);
  const int lo = __builtin_amdgcn_update_dpp((int)ob, (int)xb, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(ob >> 32), (int)(xb >> 32), CTRL, ROW_MASK, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
// the same with zeros shifted in (bound_ctrl) and every row enabled: no `old` operand, i.e. no two v_mov in front of the pair
// of DPP moves (shifts inside a row and the wave shift; the row broadcasts keep rows masked and need `old`)
template <int CTRL>
__device__ __forceinline__ double dpp_f64z(double x) {
#ifdef RPDE_DPP_OLD            // experiment build only (tools/r05_call4.sh): the form of round 4
  return dpp_f64<CTRL, 0xF>(0.0, x);
#else
  const long long xb = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)xb, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(xb >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
#endif
}
template <int ORDER, int CTRL, int ROW_MASK>
__device__ __forceinline__ Affine<ORDER> affine_dpp(const Affine<ORDER>& a) {
  Affine<ORDER> r = affine_identity<ORDER>();
  // the entries whose identity value is zero take the zero-fill form where every row takes part (no `old` operand)
  auto z = [](double x) { if constexpr (ROW_MASK == 0xF) return dpp_f64z<CTRL>(x); else return dpp_f64<CTRL, ROW_MASK>(0.0, x); };
  r.m11 = dpp_f64<CTRL, ROW_MASK>(1.0, a.m11);
  r.v1 = z(a.v1);
  if constexpr (ORDER == 2) {
    r.m12 = z(a.m12); r.m21 = z(a.m21);
    r.m22 = dpp_f64<CTRL, ROW_MASK>(1.0, a.m22); r.v2 = z(a.v2);
  }
  return r;
}
// inclusive scan over the 64 lanes: Kogge-Stone inside the rows of 16, then the row totals travel on
template <int ORDER>
__device__ __forceinline__ Affine<ORDER> affine_wave_scan(Affine<ORDER> inc) {
  inc = affine_compose<ORDER>(inc, affine_dpp<ORDER, 0x111, 0xF>(inc));
  inc = affine_compose<ORDER>(inc, affine_dpp<ORDER, 0x112, 0xF>(inc));
  inc = affine_compose<ORDER>(inc, affine_dpp<ORDER, 0x114, 0xF>(inc));
  inc = affine_compose<ORDER>(inc, affine_dpp<ORDER, 0x118, 0xF>(inc));
  inc = affine_compose<ORDER>(inc, affine_dpp<ORDER, 0x142, 0xA>(inc));
  inc = affine_compose<ORDER>(inc, affine_dpp<ORDER, 0x143, 0xC>(inc));
  return inc;
}
__device__ __forceinline__ double sum_wave_scan(double v) {
  v += dpp_f64z<0x111>(v);
  v += dpp_f64z<0x112>(v);
  v += dpp_f64z<0x114>(v);
  v += dpp_f64z<0x118>(v);
  v += dpp_f64<0x142, 0xA>(0.0, v);
  v += dpp_f64<0x143, 0xC>(0.0, v);
  return v;
}
template <int ORDER>
__device__ __forceinline__ Affine<ORDER> affine_shfl_idx(const Affine<ORDER>& a, int src) {
  Affine<ORDER> r = affine_identity<ORDER>();
  r.m11 = __shfl(a.m11, src);
  r.v1 = __shfl(a.v1, src);
  if constexpr (ORDER == 2) {
    r.m12 = __shfl(a.m12, src); r.m21 = __shfl(a.m21, src); r.m22 = __shfl(a.m22, src);
    r.v2 = __shfl(a.v2, src);
  }
  return r;
}
#endif

// `fill(k, b, q, r)` supplies the coefficients of element k (it may read LDS / padded tables
// freely, also for k >= n); all loads of a chunk are issued in one batch before the dependent
// chains start.
// ORDER 2 parks its r coefficients in the LDS area `scr` (T * C doubles, thread-private entries
// [i * T + tid]) between the chunk reduction and the re-run: 2 C fewer live VGPRs across the
// prefix phase, which is what keeps the kernel at 128 VGPRs without scratch memory traffic.
template <class Cfg, int ORDER, int DIR, class Fill>
RPDE_DEVN void scan_recurrence(Blk& blk, lds_t dst, int n, lds_t carry, const Fill fill,
                               lds_t scr = (lds_t) nullptr) {
  constexpr int T = Cfg::T, C = Cfg::C;
  constexpr int W = 6;  // doubles per stored map
  RPDE_TLS(blk, double, bb, C);
  RPDE_TLS(blk, double, qq, C);
  RPDE_TLS(blk, double, rr, C);
  RPDE_TLS(blk, double, cm, 2 * W);     // chunk map per parity, later the inflow state (v1, v2)
  // ---- phase 0 + 1: batch the coefficient loads, then reduce the chunk to an affine map
  RPDE_PHASE(blk, tid) {
    const int lo = ((DIR > 0) ? tid : (T - 1 - tid)) * C;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      double b, q, r = 0.0;
      fill(lo + i, i * T + tid, b, q, r);
      RPDE_T(bb)[i] = b; RPDE_T(qq)[i] = q; RPDE_T(rr)[i] = r;
      if constexpr (ORDER == 2) scr[i * T + tid] = r;
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double z1 = 0.0, z2 = 0.0;           // inhomogeneous run from the zero state
      double a11 = 1.0, a12 = 0.0;         // homogeneous runs from (1,0) and (0,1)
      double a21 = 0.0, a22 = 1.0;         // state = (most recent value, the one before)
#pragma unroll
      for (int i = 0; i < C / 2; ++i) {
        const int e = (DIR > 0) ? (par + 2 * i) : (C - 2 + par - 2 * i);
        const bool ok = lo + e < n;
        const double q = RPDE_T(qq)[e], bk = RPDE_T(bb)[e];
        if constexpr (ORDER == 1) {
          z1 = ok ? bk + q * z1 : z1;
          a11 = ok ? q * a11 : a11;
        } else {
          const double r = RPDE_T(rr)[e];
          const double nz = bk + q * z1 + r * z2;
          const double n1 = q * a11 + r * a21;
          const double n2 = q * a12 + r * a22;
          z2 = ok ? z1 : z2; z1 = ok ? nz : z1;
          a21 = ok ? a11 : a21; a11 = ok ? n1 : a11;
          a22 = ok ? a12 : a22; a12 = ok ? n2 : a12;
        }
      }
      double* m = RPDE_T(cm) + par * W;
      m[0] = a11; m[1] = a12; m[2] = a21; m[3] = a22; m[4] = z1; m[5] = z2;
    }
  }
  // ---- phase 2: inflow state of every chunk = (exclusive prefix of the maps)(0)
#ifdef RPDE_EMU
  {
    Affine<ORDER> run[2] = {affine_identity<ORDER>(), affine_identity<ORDER>()};
    RPDE_PHASE(blk, tid) {  // the emulation visits tid = 0 .. T-1 in order
      for (int par = 0; par < 2; ++par) {
        double* m = RPDE_T(cm) + par * W;
        const Affine<ORDER> mine{m[0], m[1], m[2], m[3], m[4], m[5]};
        m[4] = run[par].v1; m[5] = run[par].v2;
        run[par] = affine_compose<ORDER>(mine, run[par]);
      }
    }
    (void)carry;
  }
#else
  {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = (T + 63) / 64;
    Affine<ORDER> inc[2], exc[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double* m = cm + par * W;
      inc[par] = Affine<ORDER>{m[0], m[1], m[2], m[3], m[4], m[5]};
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {     // the two parities give two independent chains
      inc[par] = affine_wave_scan<ORDER>(inc[par]);
      exc[par] = affine_dpp<ORDER, 0x138, 0xF>(inc[par]);   // wave_shr:1, lane 0 gets the identity
    }
    if constexpr (NW > 1) {
      if (lane == 63) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          lds_t wt = carry + (par * NW + wave) * W;
          wt[0] = inc[par].m11; wt[1] = inc[par].m12; wt[2] = inc[par].m21; wt[3] = inc[par].m22;
          wt[4] = inc[par].v1; wt[5] = inc[par].v2;
        }
      }
      __syncthreads();
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        // prefix over the (at most 16) wave totals: lane l < NW holds total l, a log2(NW)-level shuffle
        // scan composes them, wave w picks the inclusive prefix of wave w-1
        static_assert(NW <= 16, "cross-wave scan: one lane per wave total, log2(NW) shuffle levels");
        clds_t p = carry + (par * NW + (lane < NW ? lane : 0)) * W;
        Affine<ORDER> t = Affine<ORDER>{p[0], p[1], p[2], p[3], p[4], p[5]};
        t = affine_select<ORDER>(lane < NW, t, affine_identity<ORDER>());
#pragma unroll
        for (int off = 1; off < NW; off <<= 1) {
          const Affine<ORDER> prev = affine_shfl_up<ORDER>(t, off);
          t = affine_select<ORDER>(lane >= off, affine_compose<ORDER>(t, prev), t);
        }
        Affine<ORDER> pre = affine_shfl_idx<ORDER>(t, wave > 0 ? wave - 1 : 0);
        pre = affine_select<ORDER>(wave > 0, pre, affine_identity<ORDER>());
        exc[par] = affine_compose<ORDER>(exc[par], pre);
      }
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) { cm[par * W + 4] = exc[par].v1; cm[par * W + 5] = exc[par].v2; }
  }
#endif
  // ---- phase 3: re-run every chunk with its inflow state (registers only), then write
  RPDE_TLS(blk, double, res, C);
  RPDE_PHASE(blk, tid) {
    const int lo = ((DIR > 0) ? tid : (T - 1 - tid)) * C;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double x1 = RPDE_T(cm)[par * W + 4], x2 = RPDE_T(cm)[par * W + 5];
#pragma unroll
      for (int i = 0; i < C / 2; ++i) {
        const int e = (DIR > 0) ? (par + 2 * i) : (C - 2 + par - 2 * i);
        const bool ok = lo + e < n;
        const double q = RPDE_T(qq)[e], bk = RPDE_T(bb)[e];
        if constexpr (ORDER == 1) {
          x1 = ok ? bk + q * x1 : x1;
        } else {
          const double nx1 = bk + q * x1 + scr[e * T + tid] * x2;
          x2 = ok ? x1 : x2; x1 = ok ? nx1 : x1;
        }
        RPDE_T(res)[e] = x1;
      }
    }
  }
  RPDE_SYNC(blk);   // every thread has consumed its inputs (in-place operation is allowed)
  RPDE_PHASE(blk, tid) {
    const int lo = ((DIR > 0) ? tid : (T - 1 - tid)) * C;
#pragma unroll
    for (int i = 0; i < C; ++i)
      if (lo + i < n) dst[lo + i] = RPDE_T(res)[i];
  }
  RPDE_SYNC(blk);
}

// ---------------------------------------------------------------------------------------------
// Chebyshev derivative d_k = d_{k+2} + 2 (k+1) a_{k+1} (funspace gradient, src/field.rs:127-129) as a
// plain suffix sum: the affine maps of scan_recurrence all have the matrix 1, so a chunk is reduced
// to ONE number per parity, the prefix moves one double per step (no map composition) and the
// re-run of the chunk becomes an addition.  0.109 ms against 0.117 ms per 4097 x 4096 launch for the
// generic scan.  (The same idea with TABULATED matrices for the banded solves -- two doubles per
// step times a fetched 2 x 2 window matrix, homogeneous responses fetched instead of re-run -- was
// measured SLOWER than composing the maps on the fly, 0.33 vs 0.27 ms for the Helmholtz solve: the
// extra table fetches from L2 cost more than the arithmetic they replace.  Dropped.)
template <class Cfg>
RPDE_DEVN void scan_cheb_diff(Blk& blk, lds_t dst, clds_t src, int n, lds_t carry, double scale) {
  constexpr int T = Cfg::T, C = Cfg::C, NW = (T + 63) / 64;
  RPDE_TLS(blk, double, zz, C);        // suffix sums of the chunk from a zero inflow, later the result
  RPDE_TLS(blk, double, vv, 2);        // chunk totals per parity, later the chunk inflows
  RPDE_PHASE(blk, tid) {
    const int lo = (T - 1 - tid) * C;  // descending: the carry flows from thread t-1 to thread t
    double bb[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const int k = lo + i;
      const double sv = src[k + 1];
      bb[i] = (k + 1 < n) ? 2.0 * (double)(k + 1) * sv : 0.0;
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double z = 0.0;
#pragma unroll
      for (int i = 0; i < C / 2; ++i) {
        const int e = C - 2 + par - 2 * i;
        z += (lo + e < n) ? bb[e] : 0.0;
        RPDE_T(zz)[e] = z;
      }
      RPDE_T(vv)[par] = z;
    }
  }
#ifdef RPDE_EMU
  for (int par = 0; par < 2; ++par) {
    double run = 0.0;
    for (int t = 0; t < T; ++t) { const double mine = vv_st[(size_t)t * 2 + par]; vv_st[(size_t)t * 2 + par] = run; run += mine; }
  }
  (void)carry;
#else
  {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[2] = {vv[0], vv[1]};
    v[0] = sum_wave_scan(v[0]);
    v[1] = sum_wave_scan(v[1]);
    double S[2] = {0.0, 0.0};
    if constexpr (NW > 1) {
      if (lane == 63) { carry[wave] = v[0]; carry[NW + wave] = v[1]; }
      __syncthreads();
      for (int u = 0; u < wave; ++u) { S[0] += carry[u]; S[1] += carry[NW + u]; }
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) vv[par] = dpp_f64z<0x138>(v[par]) + S[par];   // wave_shr:1
  }
#endif
  RPDE_SYNC(blk);   // every thread has consumed its inputs (in-place operation is allowed)
  RPDE_PHASE(blk, tid) {
    const int lo = (T - 1 - tid) * C;
#pragma unroll
    for (int i = 0; i < C; ++i) {
      const int k = lo + i;
      const double x = (RPDE_T(zz)[i] + RPDE_T(vv)[i & 1]) * ((k == 0) ? 0.5 * scale : scale);
      if (k < n) dst[k] = x;
    }
  }
  RPDE_SYNC(blk);
}

// generic table-driven recurrence: x_k = p_k src_k + q_k x_pred (+ r_k x_pred2).  The tables are
// stored CHUNK-MAJOR for the kernel configuration (entry [i * T + t] belongs to element i of the
// chunk thread t owns, see chunk_major() in kernels.h), so that a wave reads 512 contiguous bytes.
template <bool HASP, bool HASR>
struct FillRec {
  clds_t src; tab_t pt; tab_t qt; tab_t rt;
  RPDE_DEV void operator()(int k, int ti, double& b, double& q, double& r) const {
    const double s = src[k];
    if constexpr (HASP) b = pt[ti] * s; else b = s;
    q = qt[ti];
    if constexpr (HASR) r = rt[ti];
  }
};

// ---------------------------------------------------------------------------------------------
// the interpreter.  Conventions that keep the inner loops free of divergent branches (with one
// or two waves per SIMD there is little to hide a stall behind): a slot is addressable up to
// T*EPT >= slot_len doubles (the LDS allocation is rounded up accordingly), so LDS reads are
// never guarded; device tables carry kTableSlack doubles of zero padding, so table reads are
// never guarded; results are selected and stores are masked.
constexpr int kTableSlack = 5200;

// VAR selects what is compiled in: the second-order scan (OP_REC2) and the register stash
// (OP_PUSH / OP_POPAXPY, 2 EPT VGPRs live across the whole program) are each left out of the
// variants that do not need them
// A contiguous line of n doubles can be moved as 16-byte pairs (k, k + 1), k even, when its first element is
// 16-byte aligned and element n exists whenever n is odd (the pitch of an array is a multiple of 16 doubles).
// Thread t then owns the pairs k = 2 t + 2 T j -- half as many vector-memory and LDS instructions per line.
RPDE_HD inline bool line_vec16(const double* p, int n, long ld) {
  return (((size_t)p) & 15) == 0 && ((n & 1) == 0 || n < ld);
}

template <class Cfg, int VAR = kVarAll>
RPDE_DEV void run_line_program(Blk& blk, const Program& pg) {
  constexpr int T = Cfg::T, EPT = Cfg::EPT;
  constexpr bool FULL = (VAR & kVarRec2) != 0, STASH = (VAR & kVarStash) != 0;
  // paired loads hold two or three rows in registers: not in the 1024-thread configuration (18 elements per
  // thread would spill); there the ops of a pair simply run one after the other
  constexpr bool kPairs = EPT <= 12;
  constexpr bool kVec = (EPT % 2) == 0;   // 16-byte line moves (line_vec16)
  RPDE_TLS(blk, double, stash, STASH ? EPT : 1);
  const int line = blk.line, comp = blk.comp;
  const int SL = pg.slot_len;
  lds_t lds = (lds_t)blk.lds;
  lds_t carry = lds + pg.nslots * SL + (T * EPT - SL) + 8;
#ifndef RPDE_EMU
  if (blk.trc && threadIdx.x == 0) blk.trc[0] = (long long)wall_clock64();
#endif
  for (int ip = 0; ip < pg.nops; ++ip) {
    const Op& op = pg.ops[ip];
    RPDE_MARK(blk, ip);
    lds_t d = lds + op.d * SL;
    clds_t a = lds + op.a * SL;
    clds_t b = lds + op.b * SL;
    const int n = op.n;
    const int gline = line + pg.line0;           // global line: tables and stencil coefficients
    const long toff = op.tabld * gline;
    switch (op.code) {
      case OP_LOAD: {
        const ArrayRef& A = pg.arr[op.arr];
        cgmem_t src = (cgmem_t)(A.p + comp * A.coff + (long)line * A.ld);
        const int es = A.es;
        const bool plain = !op.i0 && es == 1;       // contiguous line: no 64-bit index arithmetic
        if (kPairs && op.b == 1) {
          // pair of plain loads (ProgramBuilder::pair_last_loads): the loads of this op and of the next one are in
          // flight together -- one HBM round trip instead of two; the second may target the same slot (applied in order)
          const Op& o2 = pg.ops[ip + 1];
          const ArrayRef& A2 = pg.arr[o2.arr];
          cgmem_t src2 = (cgmem_t)(A2.p + comp * A2.coff + (long)line * A2.ld);
          lds_t d2 = lds + o2.d * SL;
          const int n2 = o2.n;
          const bool same = o2.d == op.d;
          if (kVec && (SL & 1) == 0 && line_vec16(A.p + comp * A.coff + (long)line * A.ld, n, A.ld) &&
              line_vec16(A2.p + comp * A2.coff + (long)line * A2.ld, n2, A2.ld)) {
            cgmem2_t s1 = (cgmem2_t)src, s2 = (cgmem2_t)src2;
            lds2_t dd = (lds2_t)d, dd2 = (lds2_t)d2;
            RPDE_PHASE(blk, tid) {
              dbl2 v[EPT / 2], w[EPT / 2];
#pragma unroll
              for (int j = 0; j < EPT / 2; ++j) {
                const int k = 2 * tid + 2 * T * j;
                v[j] = (k < n) ? s1[k >> 1] : dbl2{0.0, 0.0};
              }
#pragma unroll
              for (int j = 0; j < EPT / 2; ++j) {
                const int k = 2 * tid + 2 * T * j;
                w[j] = (k < n2) ? s2[k >> 1] : dbl2{0.0, 0.0};
              }
#pragma unroll
              for (int j = 0; j < EPT / 2; ++j) {
                const int k = 2 * tid + 2 * T * j;
                const double x0 = op.s0 * v[j].x, x1 = (k + 1 < n) ? op.s0 * v[j].y : 0.0;
                const double y0 = o2.s0 * w[j].x, y1 = (k + 1 < n2) ? o2.s0 * w[j].y : 0.0;
                const dbl2 old = dd[k >> 1];
                dbl2 r, r2;
                r.x = (op.acc == 2) ? (k < n ? old.x * x0 : old.x) : (op.acc ? (old.x + x0) : x0);
                r.y = (op.acc == 2) ? (k + 1 < n ? old.y * x1 : old.y) : (op.acc ? (old.y + x1) : x1);
                const dbl2 old2 = same ? r : dd2[k >> 1];
                r2.x = (o2.acc == 2) ? (k < n2 ? old2.x * y0 : old2.x) : (o2.acc ? (old2.x + y0) : y0);
                r2.y = (o2.acc == 2) ? (k + 1 < n2 ? old2.y * y1 : old2.y) : (o2.acc ? (old2.y + y1) : y1);
                if (k < SL) {
                  if (same) dd[k >> 1] = r2;
                  else { dd[k >> 1] = r; dd2[k >> 1] = r2; }
                }
              }
            }
            RPDE_SYNC(blk);
            ++ip;
            break;
          }
          RPDE_PHASE(blk, tid) {
            double v[EPT], w[EPT];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              v[q] = (k < n) ? src[k] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              w[q] = (k < n2) ? src2[k] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              const double x = op.s0 * v[q], y = o2.s0 * w[q];
              const double old = d[k];
              const double r = (op.acc == 2) ? (k < n ? old * x : old) : (op.acc ? (old + x) : x);
              const double old2 = same ? r : d2[k];
              const double r2 = (o2.acc == 2) ? (k < n2 ? old2 * y : old2) : (o2.acc ? (old2 + y) : y);
              if (k < SL) {
                if (same) d[k] = r2;
                else { d[k] = r; d2[k] = r2; }
              }
            }
          }
          RPDE_SYNC(blk);
          ++ip;                                       // the second op of the pair is done
          break;
        }
        if (kVec && plain && (SL & 1) == 0 && line_vec16(A.p + comp * A.coff + (long)line * A.ld, n, A.ld)) {
          cgmem2_t src2 = (cgmem2_t)src;
          lds2_t dd = (lds2_t)d;
          RPDE_PHASE(blk, tid) {
            dbl2 v[EPT / 2];
#pragma unroll
            for (int j = 0; j < EPT / 2; ++j) {
              const int k = 2 * tid + 2 * T * j;
              v[j] = (k < n) ? src2[k >> 1] : dbl2{0.0, 0.0};
            }
#pragma unroll
            for (int j = 0; j < EPT / 2; ++j) {
              const int k = 2 * tid + 2 * T * j;
              const double x0 = op.s0 * v[j].x, x1 = (k + 1 < n) ? op.s0 * v[j].y : 0.0;
              const dbl2 old = dd[k >> 1];
              dbl2 r;
              r.x = (op.acc == 2) ? (k < n ? old.x * x0 : old.x) : (op.acc ? (old.x + x0) : x0);
              r.y = (op.acc == 2) ? (k + 1 < n ? old.y * x1 : old.y) : (op.acc ? (old.y + x1) : x1);
              if (k < SL) dd[k >> 1] = r;
            }
          }
          RPDE_SYNC(blk);
          break;
        }
        RPDE_PHASE(blk, tid) {
          double v[EPT];
          if (plain) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              v[q] = (k < n) ? src[k] : 0.0;
            }
          } else {
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              if (op.i0 == 2) {   // (i kappa) * complex: swap re/im inside the pair; sign and wavenumber are applied below (a
                v[q] = (k < n) ? src[(long)(k ^ 1) * es] : 0.0;   // product here would put a wait behind every load)
              } else {
                const long kk = op.i0 ? ((long)(k & 1) * op.i1 + (k >> 1)) : k;   // i0: parity de-interleaved source
                v[q] = (k < n) ? src[kk * es] : 0.0;
              }
            }
          }
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const double wk = (op.i0 == 2) ? ((k & 1) ? (double)(k >> 1) : -(double)(k >> 1)) : 1.0;
            const double x = op.s0 * (wk * v[q]);
            const double old = d[k];
            if (k < SL) d[k] = (op.acc == 2) ? (k < n ? old * x : old) : (op.acc ? (old + x) : x);
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_LOADX: {
        const ArrayRef& A = pg.arr[op.arr];
        const bool has0 = gline < op.i1, has2 = gline >= 2 && (gline - 2) < op.i1;
        cgmem_t s0p = (cgmem_t)(A.p + comp * A.coff + (long)line * A.ld);
        cgmem_t s2p = (cgmem_t)(A.p + comp * A.coff + (long)(line - 2) * A.ld);
        const double c2 = has2 ? ((tab_t)pg.tabs[op.tab])[gline - 2] : 0.0;
        const int es = A.es;
        if (kPairs && op.b == 1) {   // paired with the plain load that follows (see OP_LOAD): three rows in flight together
          const Op& o2 = pg.ops[ip + 1];
          const ArrayRef& A2 = pg.arr[o2.arr];
          cgmem_t src2 = (cgmem_t)(A2.p + comp * A2.coff + (long)line * A2.ld);
          lds_t d2 = lds + o2.d * SL;
          const int n2 = o2.n;
          const bool same = o2.d == op.d;
          RPDE_PHASE(blk, tid) {
            double v0[EPT], v2[EPT], w[EPT];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              v0[q] = (has0 && k < n) ? s0p[(long)k * es] : 0.0;
              v2[q] = (has2 && k < n) ? s2p[(long)k * es] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              w[q] = (k < n2) ? src2[k] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              const double x = op.s0 * (v0[q] + c2 * v2[q]), y = o2.s0 * w[q];
              const double r = op.acc ? (d[k] + x) : x;
              const double old2 = same ? r : d2[k];
              const double r2 = (o2.acc == 2) ? (k < n2 ? old2 * y : old2) : (o2.acc ? (old2 + y) : y);
              if (k < SL) {
                if (same) d[k] = r2;
                else { d[k] = r; d2[k] = r2; }
              }
            }
          }
          RPDE_SYNC(blk);
          ++ip;
          break;
        }
        RPDE_PHASE(blk, tid) {
          double v0[EPT], v2[EPT];
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const long kk = op.i0 ? ((long)(k & 1) * op.i0 + (k >> 1)) : (long)k;   // i0 = half: parity de-interleaved rows
            v0[q] = (has0 && k < n) ? s0p[kk * es] : 0.0;
            v2[q] = (has2 && k < n) ? s2p[kk * es] : 0.0;
          }
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const double x = op.s0 * (v0[q] + c2 * v2[q]);
            if (k < SL) d[k] = op.acc ? (d[k] + x) : x;
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_STORE: {
        const ArrayRef& A = pg.arr[op.arr];
        gmem_t dstp = (gmem_t)(A.p + comp * A.coff + (long)line * A.ld);
        const bool plain = !op.i0 && A.es == 1;
        const bool guard = op.acc != 0 && pg.nanflag != nullptr;
        if (kVec && plain && (SL & 1) == 0 && line_vec16(A.p + comp * A.coff + (long)line * A.ld, n, A.ld)) {
          gmem2_t dst2 = (gmem2_t)dstp;
          clds2_t aa = (clds2_t)a;
          RPDE_PHASE(blk, tid) {
            bool bad = false;
#pragma unroll
            for (int j = 0; j < EPT / 2; ++j) {
              const int k = 2 * tid + 2 * T * j;
              const dbl2 x = aa[k >> 1];
              const double x0 = op.s0 * x.x, x1 = op.s0 * x.y;
              if (k + 1 < n) { dst2[k >> 1] = dbl2{x0, x1}; bad |= (x0 != x0) | (x1 != x1); }
              else if (k < n) { dstp[k] = x0; bad |= (x0 != x0); }
            }
            if (guard && bad) *pg.nanflag = 1;
          }
          RPDE_SYNC(blk);
          break;
        }
        RPDE_PHASE(blk, tid) {
          if (plain) {
            bool bad = false;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              const double x = op.s0 * a[k];
              if (k < n) { dstp[k] = x; bad |= (x != x); }
            }
            // device-side NaN guard: only a thread that actually stored a NaN touches the flag
            if (guard && bad) *pg.nanflag = 1;
          } else {
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              const double x = op.s0 * a[k];
              if (k < n) {
                const long kk = op.i0 ? ((long)(k & 1) * op.i1 + (k >> 1)) : k;
                dstp[kk * A.es] = x;
              }
            }
          }
        }
        RPDE_SYNC(blk);  // the next op may overwrite the slot
      } break;
      case OP_STEN: if constexpr (Cfg::kCheb) {
        tab_t low = (tab_t)(pg.tabs[op.tab] + toff);
        if (op.d != op.a) {   // out of place: no thread overwrites what another one still reads -- one phase
          RPDE_PHASE(blk, tid) {
            double lw[EPT];   // stencil coefficients: all fetched before the first use
#pragma unroll
            for (int q = 0; q < EPT; ++q) lw[q] = low[max(tid + q * T - 2, 0)];
#pragma unroll
            for (int q = 0; q < EPT; ++q) RPDE_PIN(lw[q]);
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              const int k2 = (k >= 2) ? k - 2 : 0;
              const double a0 = a[k], a2 = a[k2], l2 = lw[q];
              double x = (k < n - 2) ? a0 : 0.0;
              x += (k >= 2) ? l2 * a2 : 0.0;
              if (op.acc) x = op.s1 * d[k] + op.s0 * x;   // fused axpby: d = s1 d + s0 S a
              if (k < n) d[k] = x;
            }
          }
          RPDE_SYNC(blk);
          break;
        }
        RPDE_TLS(blk, double, v, EPT);
        RPDE_PHASE(blk, tid) {
          double lw[EPT];
#pragma unroll
          for (int q = 0; q < EPT; ++q) lw[q] = low[max(tid + q * T - 2, 0)];
#pragma unroll
          for (int q = 0; q < EPT; ++q) RPDE_PIN(lw[q]);
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const int k2 = (k >= 2) ? k - 2 : 0;
            const double a0 = a[k], a2 = a[k2], l2 = lw[q];
            double x = (k < n - 2) ? a0 : 0.0;
            x += (k >= 2) ? l2 * a2 : 0.0;
            RPDE_T(v)[q] = x;
          }
        }
        RPDE_SYNC(blk);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) { const int k = tid + q * T; if (k < n) d[k] = RPDE_T(v)[q]; }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_MV3: if constexpr (Cfg::kCheb) {
        tab_t t0 = (tab_t)(pg.tabs[op.tab] + toff);
        tab_t t1 = (tab_t)(pg.tabs[op.tab + 1] + toff);
        tab_t t2 = (tab_t)(pg.tabs[op.tab + 2] + toff);
        RPDE_TLS(blk, double, v, EPT);
        RPDE_PHASE(blk, tid) {
          // the three band coefficients of half of this thread's rows are in flight together (with per-line
          // tables they come from HBM); tables carry slack, so the reads need no predicate
          constexpr int H0 = EPT / 2;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            constexpr int HN = EPT - H0 > H0 ? EPT - H0 : H0;
            const int q0 = h ? H0 : 0, q1 = h ? EPT : H0;
            double c0[HN], c1[HN], c2[HN];
#pragma unroll
            for (int q = q0; q < q1; ++q) { const int k = tid + q * T; c0[q - q0] = t0[k]; c1[q - q0] = t1[k]; c2[q - q0] = t2[k]; }
#pragma unroll
            for (int q = q0; q < q1; ++q) { RPDE_PIN(c0[q - q0]); RPDE_PIN(c1[q - q0]); RPDE_PIN(c2[q - q0]); }
#pragma unroll
            for (int q = q0; q < q1; ++q) {
              const int k = tid + q * T;
              // input line has n + 2 entries; the +4 tap exists for k < n - 2 only (matvec.rs:215-226)
              const double a0 = a[k], a2 = a[k + 2], a4 = a[k + 4];
              double x = a0 * c0[q - q0] + a2 * c1[q - q0];
              x += (k < n - 2) ? a4 * c2[q - q0] : 0.0;
              RPDE_T(v)[q] = x;
            }
          }
        }
        RPDE_SYNC(blk);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) { const int k = tid + q * T; if (k < n) d[k] = RPDE_T(v)[q]; }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_CDIFF: if constexpr (Cfg::kCheb) {
        scan_cheb_diff<Cfg>(blk, d, a, n, carry, op.s0);
      } break;
      case OP_REC1: if constexpr (Cfg::kCheb) {
        tab_t qt = (tab_t)(pg.tabs[op.i0] + toff);
        if (op.tab >= 0) {
          const FillRec<true, false> f{a, (tab_t)(pg.tabs[op.tab] + toff), qt, (tab_t) nullptr};
          if (op.i1 > 0) scan_recurrence<Cfg, 1, +1>(blk, d, n, carry, f);
          else scan_recurrence<Cfg, 1, -1>(blk, d, n, carry, f);
        } else {
          const FillRec<false, false> f{a, (tab_t) nullptr, qt, (tab_t) nullptr};
          if (op.i1 > 0) scan_recurrence<Cfg, 1, +1>(blk, d, n, carry, f);
          else scan_recurrence<Cfg, 1, -1>(blk, d, n, carry, f);
        }
      } break;
      case OP_REC2: {
        if constexpr (FULL && Cfg::kCheb) {
          const FillRec<true, true> f{a, (tab_t)(pg.tabs[op.tab] + toff), (tab_t)(pg.tabs[op.i0] + toff),
                                      (tab_t)(pg.tabs[op.i1] + toff)};
          scan_recurrence<Cfg, 2, -1>(blk, d, n, carry, f, lds + op.b * SL);
        }
      } break;
      case OP_DCT: if constexpr (Cfg::kCheb) {
        // FFT path: tab / i0 >= 0 switch the standard scalings on (evaluated arithmetically, a = cut,
        // s1 = 1/N); direct path: tab / i0 are table indices.
        // fused forms (FFT path only): i1 >= 0: composite->ortho stencil table applied while packing;
        // arr >= 0: results go straight to the global array (b = count, s0 = scale) instead of LDS
        if (pg.fft_n > 0) {
          // i1: -1 no stencil, >= 0 stencil table, -2 Dirichlet stencil (constant -1, no table)
          tab_t low = op.i1 >= 0 ? (tab_t)pg.tabs[op.i1] : (tab_t) nullptr;
          const int sten = op.i1 >= 0 ? 1 : (op.i1 == -2 ? 2 : 0);
          gmem_t gdst = (gmem_t) nullptr;
          int ges = 1;
          if (op.arr >= 0) {
            const ArrayRef& A = pg.arr[op.arr];
            gdst = (gmem_t)(A.p + comp * A.coff + (long)line * A.ld);
            ges = A.es;
          }
          dct1_lds<Cfg>(blk, d, n - 1, op.tab >= 0, op.i0 >= 0, op.a, op.s1, (tab_t)pg.tabs[pg.tw],
                        (tab_t)pg.tabs[pg.tw2], sten, low, gdst, ges, op.s0, op.b);
        } else {
          tab_t pre = op.tab >= 0 ? (tab_t)pg.tabs[op.tab] : (tab_t) nullptr;
          tab_t post = op.i0 >= 0 ? (tab_t)pg.tabs[op.i0] : (tab_t) nullptr;
          if (pg.blu_m > 0) dct1_bluestein<Cfg>(blk, d, n - 1, pg.blu_m, pre, post, (tab_t)pg.tabs[pg.tw], (tab_t)pg.tabs[pg.tw2]);
          else dct1_direct<Cfg>(blk, d, n - 1, pre, post, (tab_t)pg.tabs[pg.tw2]);
        }
      } break;
      case OP_MUL: {
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const double v = op.s0 * a[k] * b[k];
            const double old = d[k];
            if (k < n) d[k] = op.acc ? old + v : v;
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_AXPBY: {
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const double v = op.s0 * a[k] + op.s1 * b[k];
            if (k < n) d[k] = v;
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_PUSH: {
        if constexpr (STASH) {
          RPDE_PHASE(blk, tid) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) RPDE_T(stash)[q] = a[tid + q * T];
          }
          RPDE_SYNC(blk);  // the next op may overwrite the slot
        }
      } break;
      case OP_POPAXPY: {
        if constexpr (STASH) {
          RPDE_PHASE(blk, tid) {
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
              const int k = tid + q * T;
              const double v = op.s0 * d[k] + op.s1 * RPDE_T(stash)[q];
              if (k < n) d[k] = v;
            }
          }
          RPDE_SYNC(blk);
        }
      } break;
      case OP_ZERO: {
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k >= op.i0 && k < op.i1 && k < SL) d[k] = 0.0;
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_TABDIV: {
        tab_t t = (tab_t)(pg.tabs[op.tab] + toff);
        RPDE_PHASE(blk, tid) {
          double den[EPT];   // unpredicated table reads, all issued before the first use; lanes past the end are not stored
#pragma unroll
          for (int q = 0; q < EPT; ++q) den[q] = t[min(tid + q * T, n - 1) >> op.i0];
#pragma unroll
          for (int q = 0; q < EPT; ++q) RPDE_PIN(den[q]);
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            const double v = a[k] / den[q];
            if (k < n) d[k] = v;
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_RFFT_F:
        if (pg.blu_m > 0) rfft_bluestein<Cfg, true>(blk, d, n, pg.blu_m, (tab_t)pg.tabs[pg.tw], (tab_t)pg.tabs[pg.tw2]);
        else rfft_forward_lds<Cfg>(blk, d, n, (tab_t)pg.tabs[pg.tw], (tab_t)pg.tabs[pg.tw2]);
        break;
      case OP_RFFT_B:
        if (pg.blu_m > 0) rfft_bluestein<Cfg, false>(blk, d, n, pg.blu_m, (tab_t)pg.tabs[pg.tw], (tab_t)pg.tabs[pg.tw2]);
        else rfft_backward_lds<Cfg>(blk, d, n, (tab_t)pg.tabs[pg.tw], (tab_t)pg.tabs[pg.tw2]);
        break;
      case OP_CIK: {
        RPDE_TLS(blk, double, vr, EPT);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * T;  // double index
            const int k = e >> 1;
            const double f = op.s0 * (double)k;
            const double partner = a[e ^ 1], self = a[e];
            RPDE_T(vr)[q] = (op.i0 == 1) ? ((e & 1) ? f * partner : -f * partner) : -(f * f) * self;
          }
        }
        RPDE_SYNC(blk);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) { const int e = tid + q * T; if (e < 2 * n) d[e] = RPDE_T(vr)[q]; }
        }
        RPDE_SYNC(blk);
      } break;
      default: break;
    }
  }
  RPDE_MARK(blk, pg.nops);
#ifndef RPDE_EMU
  if (blk.trc && threadIdx.x == 0) { blk.trc[1] = (long long)wall_clock64(); blk.trc[2] = blk.nm; }
#endif
}

}  // namespace rpde
