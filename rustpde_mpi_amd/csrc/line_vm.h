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
#include "platform.h"

namespace rpde {

enum OpCode : int {
  OP_END = 0,
  OP_LOAD,     // d[k] = (acc ? d[k] : 0) + s0 * A[line][k]                  k < n   (zero tail if !acc)
  OP_LOADX,    // d[k] = (acc ? d[k] : 0) + s0 * ([line<i1] A[line][k] + [line>=2] tab[line-2] A[line-2][k])
  OP_STORE,    // A[line][map(k)] = s0 * a[k]   k < n ; i0 = 1: parity de-interleave, half = i1
  OP_STEN,     // d[k] = [k<n-2] a[k] + [k>=2] tab[k-2] * a[k-2]              k < n  (n = ortho length)
  OP_MV3,      // d[k] = t0[k] a[k] + t1[k] a[k+2] + t2[k] a[k+4]             k < n  (tables tab, tab+1, tab+2)
  OP_CDIFF,    // d = s0 * d/dx of the Chebyshev series a (length n)
  OP_REC1,     // first-order stride-2 recurrence, x_k = p_k b_k + q_k x_{k-2 dir}; p = tab (-1: ones), q = i0, dir = i1
  OP_REC2,     // descending second-order: x_k = p_k b_k + q_k x_{k+2} + r_k x_{k+4}; p = tab, q = i0, r = i1
  OP_DCT,      // slot d (scratch d+1): x <- DCT-I(x * pre) * post ; pre = tab (-1 none), post = i0 (-1 none); n = N+1
  OP_MUL,      // d[k] = (acc ? d[k] : 0) + s0 * a[k] * b[k]
  OP_AXPBY,    // d[k] = s0 * a[k] + s1 * b[k]
  OP_ZERO,     // d[k] = 0 for i0 <= k < i1
  OP_TABDIV,   // d[k] = a[k] / tab[k >> i0]   (i0 = 1 for interleaved complex lines)
  OP_RFFT_F,   // slot d: real (n = nx) -> interleaved complex (nx/2+1), unnormalised
  OP_RFFT_B,   // slot d: interleaved complex (nx/2+1) -> real (n = nx), scaled by 1/nx
  OP_CIK,      // complex line: d = (i k s0)^i0 * a  for k < n (complex count)
};

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
  int nlines;     // grid.x
  int ncomp;      // grid.y
  int fft_n;      // complex FFT length used by OP_DCT / OP_RFFT (0: direct O(n^2) DCT)
  int tw;         // table index of the FFT twiddles W_N (N complex: cos, -sin)
  int tw2;        // table index of the split twiddles (cos, sin)(pi k / N) resp. (2 pi k / nx)
  Op ops[kMaxOps];
  ArrayRef arr[kMaxArr];
  const double* tabs[kMaxTab];
};

// ---------------------------------------------------------------------------------------------
// kernel configuration (compile time): T threads, EPT elements per thread
template <int T_, int EPT_, int FMIN_, int FMAX_>
struct LineCfg {
  static constexpr int T = T_;
  static constexpr int EPT = EPT_;
  static constexpr int FMIN = FMIN_;                   // smallest / largest complex FFT length
  static constexpr int FMAX = FMAX_;                   // instantiated in this configuration
  static constexpr int C = (EPT_ + 1) & ~1;           // scan chunk per thread (even)
  static constexpr int G = (T_ + 15) / 16;             // scan groups
  static constexpr int kMaxSlotLen = T_ * EPT_;
  static constexpr int kCarryLen = (T_ + G) * 2 * 6;  // doubles
};

RPDE_HD inline size_t line_lds_doubles(int nslots, int slot_len, int carry_len) {
  return (size_t)nslots * slot_len + carry_len;
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
// Stockham pass of radix R on N interleaved complex numbers in LDS (in place through registers)
template <class Cfg, int N, int R>
RPDE_DEV void fft_pass(Blk& blk, double* w, int Ns, const double* tw) {
  constexpr int T = Cfg::T;
  constexpr int NB = N / R;                        // butterflies
  constexpr int Q = (NB + T - 1) / T;              // butterflies per thread
  static_assert(Q * R <= 16, "FFT too large for this kernel configuration");
  RPDE_TLS(blk, double, xr, 16);
  RPDE_TLS(blk, double, xi, 16);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int j = tid + q * T;
      if (j < NB) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
          RPDE_T(xr)[q * R + t] = w[2 * (j + t * NB)];
          RPDE_T(xi)[q * R + t] = w[2 * (j + t * NB) + 1];
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
        const int k = j % Ns;
        const int tstep = N / (Ns * R);
        double* pr = RPDE_T(xr) + q * R;
        double* pi = RPDE_T(xi) + q * R;
        if (k != 0) {
#pragma unroll
          for (int t = 1; t < R; ++t) {
            const int idx = t * k * tstep;  // < N
            const double c = tw[2 * idx], s = tw[2 * idx + 1];  // W = c + i s (s = -sin)
            const double ar = pr[t], ai = pi[t];
            pr[t] = ar * c - ai * s;
            pi[t] = ar * s + ai * c;
          }
        }
        SmallDft<R>::run(pr, pi);
        const int j0 = (j / Ns) * Ns * R + k;
#pragma unroll
        for (int t = 0; t < R; ++t) {
          w[2 * (j0 + t * Ns)] = pr[t];
          w[2 * (j0 + t * Ns) + 1] = pi[t];
        }
      }
    }
  }
  RPDE_SYNC(blk);
}

template <class Cfg, int N>
RPDE_DEV void fft_lds(Blk& blk, double* w, const double* tw) {
  // radix schedule: as many 16s as possible, then one of {8,4,2}
  constexpr int L = (N >= 8192) ? 13 : (N >= 4096) ? 12 : (N >= 2048) ? 11 : (N >= 1024) ? 10
                  : (N >= 512) ? 9 : (N >= 256) ? 8 : (N >= 128) ? 7 : (N >= 64) ? 6
                  : (N >= 32) ? 5 : (N >= 16) ? 4 : (N >= 8) ? 3 : (N >= 4) ? 2 : 1;
  static_assert((1 << L) == N, "power of two");
  constexpr int n16 = L / 4;
  constexpr int rem = L % 4;
  int Ns = 1;
  if constexpr (rem == 1) { fft_pass<Cfg, N, 2>(blk, w, Ns, tw); Ns *= 2; }
  if constexpr (rem == 2) { fft_pass<Cfg, N, 4>(blk, w, Ns, tw); Ns *= 4; }
  if constexpr (rem == 3) { fft_pass<Cfg, N, 8>(blk, w, Ns, tw); Ns *= 8; }
  if constexpr (n16 >= 1) { fft_pass<Cfg, N, 16>(blk, w, Ns, tw); Ns *= 16; }
  if constexpr (n16 >= 2) { fft_pass<Cfg, N, 16>(blk, w, Ns, tw); Ns *= 16; }
  if constexpr (n16 >= 3) { fft_pass<Cfg, N, 16>(blk, w, Ns, tw); Ns *= 16; }
}

template <class Cfg>
RPDE_DEV void fft_dispatch(Blk& blk, double* w, int n, const double* tw) {
  switch (n) {
#define RPDE_FFT_CASE(NN) \
  case NN: if constexpr (NN >= Cfg::FMIN && NN <= Cfg::FMAX) fft_lds<Cfg, NN>(blk, w, tw); break;
    RPDE_FFT_CASE(2) RPDE_FFT_CASE(4) RPDE_FFT_CASE(8) RPDE_FFT_CASE(16) RPDE_FFT_CASE(32)
    RPDE_FFT_CASE(64) RPDE_FFT_CASE(128) RPDE_FFT_CASE(256) RPDE_FFT_CASE(512)
    RPDE_FFT_CASE(1024) RPDE_FFT_CASE(2048) RPDE_FFT_CASE(4096) RPDE_FFT_CASE(8192)
#undef RPDE_FFT_CASE
    default: break;
  }
}

// ---------------------------------------------------------------------------------------------
// DCT-I of the n = N+1 reals in slot x (work area = x .. x + 2N doubles, i.e. slot d and d+1):
//   E_k = x_0 + (-1)^k x_N + 2 sum_{j=1}^{N-1} x_j cos(pi j k / N)
// through an N-point complex FFT of the even extension (packed two reals per complex).
template <class Cfg>
RPDE_DEV void dct1_lds(Blk& blk, double* x, int N, const double* pre, const double* post,
                       const double* tw, const double* tw2) {
  constexpr int T = Cfg::T;
  {  // pack z_j = e_{2j} + i e_{2j+1}
    RPDE_TLS(blk, double, zr, 16);
    RPDE_TLS(blk, double, zi, 16);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = tid + q * T;
        if (j < N) {
          int m0 = 2 * j, m1 = 2 * j + 1;
          if (m0 > N) m0 = 2 * N - m0;
          if (m1 > N) m1 = 2 * N - m1;
          double a = x[m0], b = x[m1];
          if (pre) { a *= pre[m0]; b *= pre[m1]; }
          RPDE_T(zr)[q] = a;
          RPDE_T(zi)[q] = b;
        }
      }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = tid + q * T;
        if (j < N) { x[2 * j] = RPDE_T(zr)[q]; x[2 * j + 1] = RPDE_T(zi)[q]; }
      }
    }
    RPDE_SYNC(blk);
  }
  fft_dispatch<Cfg>(blk, x, N, tw);
  {  // split: E_k = (Zr_k + Zr_{N-k})/2 + (c_k (Zi_k + Zi_{N-k}) - s_k (Zr_k - Zr_{N-k}))/2
    RPDE_TLS(blk, double, e, Cfg::EPT);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < Cfg::EPT; ++q) {
        const int k = tid + q * T;
        if (k <= N) {
          const int ka = (k == N) ? 0 : k;
          const int kb = (k == 0) ? 0 : N - k;
          const double ar = x[2 * ka], ai = x[2 * ka + 1];
          const double br = x[2 * kb], bi = x[2 * kb + 1];
          const double c = tw2[2 * k], s = tw2[2 * k + 1];
          double v = 0.5 * (ar + br) + 0.5 * (c * (ai + bi) - s * (ar - br));
          if (post) v *= post[k];
          RPDE_T(e)[q] = v;
        }
      }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int q = 0; q < Cfg::EPT; ++q) {
        const int k = tid + q * T;
        if (k <= N) x[k] = RPDE_T(e)[q];
      }
    }
    RPDE_SYNC(blk);
  }
}

// direct O(n^2) DCT-I for line lengths without an FFT plan (small / odd sizes); costab[m] = cos(pi m / N), m < 2N
template <class Cfg>
RPDE_DEV void dct1_direct(Blk& blk, double* x, int N, const double* pre, const double* post,
                          const double* costab) {
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
RPDE_DEV void rfft_forward_lds(Blk& blk, double* x, int nx, const double* tw, const double* tw2) {
  constexpr int T = Cfg::T;
  const int M = nx / 2;
  fft_dispatch<Cfg>(blk, x, M, tw);  // z_j = x_{2j} + i x_{2j+1} is already the interleaved layout
  RPDE_TLS(blk, double, yr, Cfg::EPT);
  RPDE_TLS(blk, double, yi, Cfg::EPT);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Cfg::EPT; ++q) {
      const int k = tid + q * T;
      if (k <= M) {
        const int ka = (k == M) ? 0 : k;
        const int kb = (k == 0) ? 0 : M - k;
        const double ar = x[2 * ka], ai = x[2 * ka + 1];
        const double br = x[2 * kb], bi = x[2 * kb + 1];
        const double c = tw2[2 * k], s = tw2[2 * k + 1];
        const double sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
        RPDE_T(yr)[q] = 0.5 * (sr + c * di - s * dr);
        RPDE_T(yi)[q] = 0.5 * (si - c * dr - s * di);
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Cfg::EPT; ++q) {
      const int k = tid + q * T;
      if (k <= M) { x[2 * k] = RPDE_T(yr)[q]; x[2 * k + 1] = RPDE_T(yi)[q]; }
    }
  }
  RPDE_SYNC(blk);
}

// inverse: nx/2+1 interleaved complex -> nx reals, scaled by 1/nx (imaginary parts of k=0, nx/2 ignored)
template <class Cfg>
RPDE_DEV void rfft_backward_lds(Blk& blk, double* x, int nx, const double* tw, const double* tw2) {
  constexpr int T = Cfg::T;
  const int M = nx / 2;
  RPDE_TLS(blk, double, zr, Cfg::EPT);
  RPDE_TLS(blk, double, zi, Cfg::EPT);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Cfg::EPT; ++q) {
      const int k = tid + q * T;
      if (k < M) {
        const int kb = M - k;
        double ar = x[2 * k], ai = x[2 * k + 1];
        double br = x[2 * kb], bi = -x[2 * kb + 1];  // conj X_{M-k}
        if (k == 0) { ai = 0.0; bi = 0.0; }
        const double c = tw2[2 * k], s = tw2[2 * k + 1];  // conj(W^k) = c + i s
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
    for (int q = 0; q < Cfg::EPT; ++q) {
      const int k = tid + q * T;
      if (k < M) { x[2 * k] = RPDE_T(zr)[q]; x[2 * k + 1] = RPDE_T(zi)[q]; }
    }
  }
  RPDE_SYNC(blk);
  fft_dispatch<Cfg>(blk, x, M, tw);
  const double sc = 1.0 / (double)M;
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int q = 0; q < Cfg::EPT; ++q) {
      const int k = tid + q * T;
      if (k < M) { x[2 * k] *= sc; x[2 * k + 1] *= -sc; }
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
template <class Cfg, int ORDER, int DIR, class Coef>
RPDE_DEV void scan_recurrence(Blk& blk, double* dst, int n, double* carry, const Coef& cf) {
  constexpr int T = Cfg::T, C = Cfg::C, G = Cfg::G;
  constexpr int W = (ORDER == 1) ? 2 : 6;  // doubles per carried affine map
  // thread t owns elements [t*C, t*C + C); scan order tau = t (ascending) or T-1-t (descending)
  // ---- phase 1: per (thread, parity) affine map of the chunk
  RPDE_PHASE(blk, tid) {
    const int lo = tid * C;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double z1 = 0.0, z2 = 0.0;           // inhomogeneous run from zero state
      double a11 = 1.0, a12 = 0.0;         // homogeneous runs: state after chunk from (1,0)
      double a21 = 0.0, a22 = 1.0;         // ... and from (0,1)   [rows: (x1, x2)]
      // state convention: x1 = most recent value, x2 = the one before
#pragma unroll
      for (int i = 0; i < C / 2; ++i) {
        const int k = (DIR > 0) ? (lo + par + 2 * i) : (lo + C - 2 + par - 2 * i);
        if (k < n) {
          const double q = cf.q(k);
          const double bk = cf.b(k);
          if constexpr (ORDER == 1) {
            z1 = bk + q * z1;
            a11 = q * a11;
          } else {
            const double r = cf.r(k);
            const double nz = bk + q * z1 + r * z2; z2 = z1; z1 = nz;
            const double n1 = q * a11 + r * a21; a21 = a11; a11 = n1;   // column from (1,0)
            const double n2 = q * a12 + r * a22; a22 = a12; a12 = n2;   // column from (0,1)
          }
        }
      }
      const int tau = (DIR > 0) ? tid : (T - 1 - tid);
      double* cp = carry + (size_t)(tau * 2 + par) * W;
      if constexpr (ORDER == 1) { cp[0] = a11; cp[1] = z1; }
      else { cp[0] = a11; cp[1] = a12; cp[2] = a21; cp[3] = a22; cp[4] = z1; cp[5] = z2; }
    }
  }
  RPDE_SYNC(blk);
  // ---- phase 2a: exclusive prefix inside groups of 16 (in place), group aggregate behind the carries
  RPDE_PHASE(blk, tid) {
    if (tid < 2 * G) {
      const int g = tid >> 1, par = tid & 1;
      double m11 = 1, m12 = 0, m21 = 0, m22 = 1, v1 = 0, v2 = 0;
      for (int mth = 0; mth < 16; ++mth) {
        const int tau = g * 16 + mth;
        if (tau < T) {
          double* cp = carry + (size_t)(tau * 2 + par) * W;
          if constexpr (ORDER == 1) {
            const double h = cp[0], z = cp[1];
            cp[0] = m11; cp[1] = v1;
            m11 = h * m11; v1 = h * v1 + z;
          } else {
            const double c11 = cp[0], c12 = cp[1], c21 = cp[2], c22 = cp[3], z1 = cp[4], z2 = cp[5];
            cp[0] = m11; cp[1] = m12; cp[2] = m21; cp[3] = m22; cp[4] = v1; cp[5] = v2;
            const double n11 = c11 * m11 + c12 * m21, n12 = c11 * m12 + c12 * m22;
            const double n21 = c21 * m11 + c22 * m21, n22 = c21 * m12 + c22 * m22;
            const double w1 = c11 * v1 + c12 * v2 + z1, w2 = c21 * v1 + c22 * v2 + z2;
            m11 = n11; m12 = n12; m21 = n21; m22 = n22; v1 = w1; v2 = w2;
          }
        }
      }
      double* gp = carry + (size_t)(T * 2 + g * 2 + par) * W;
      if constexpr (ORDER == 1) { gp[0] = m11; gp[1] = v1; }
      else { gp[0] = m11; gp[1] = m12; gp[2] = m21; gp[3] = m22; gp[4] = v1; gp[5] = v2; }
    }
  }
  RPDE_SYNC(blk);
  // ---- phase 2b: exclusive prefix over the group aggregates (in place)
  RPDE_PHASE(blk, tid) {
    if (tid < 2) {
      const int par = tid;
      double m11 = 1, m12 = 0, m21 = 0, m22 = 1, v1 = 0, v2 = 0;
      for (int g = 0; g < G; ++g) {
        double* gp = carry + (size_t)(T * 2 + g * 2 + par) * W;
        if constexpr (ORDER == 1) {
          const double h = gp[0], z = gp[1];
          gp[0] = m11; gp[1] = v1;
          m11 = h * m11; v1 = h * v1 + z;
        } else {
          const double c11 = gp[0], c12 = gp[1], c21 = gp[2], c22 = gp[3], z1 = gp[4], z2 = gp[5];
          gp[0] = m11; gp[1] = m12; gp[2] = m21; gp[3] = m22; gp[4] = v1; gp[5] = v2;
          const double n11 = c11 * m11 + c12 * m21, n12 = c11 * m12 + c12 * m22;
          const double n21 = c21 * m11 + c22 * m21, n22 = c21 * m12 + c22 * m22;
          const double w1 = c11 * v1 + c12 * v2 + z1, w2 = c21 * v1 + c22 * v2 + z2;
          m11 = n11; m12 = n12; m21 = n21; m22 = n22; v1 = w1; v2 = w2;
        }
      }
    }
  }
  RPDE_SYNC(blk);
  // ---- phase 3: re-run every chunk with its inflow state, results through registers
  RPDE_TLS(blk, double, res, C);
  RPDE_PHASE(blk, tid) {
    const int lo = tid * C;
    const int tau = (DIR > 0) ? tid : (T - 1 - tid);
    const int g = tau >> 4;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const double* cp = carry + (size_t)(tau * 2 + par) * W;
      const double* gp = carry + (size_t)(T * 2 + g * 2 + par) * W;
      double x1, x2 = 0.0;
      if constexpr (ORDER == 1) {
        x1 = cp[0] * gp[1] + cp[1];  // inflow = He * Zg + Ze
      } else {
        x1 = cp[0] * gp[4] + cp[1] * gp[5] + cp[4];
        x2 = cp[2] * gp[4] + cp[3] * gp[5] + cp[5];
      }
#pragma unroll
      for (int i = 0; i < C / 2; ++i) {
        const int k = (DIR > 0) ? (lo + par + 2 * i) : (lo + C - 2 + par - 2 * i);
        if (k < n) {
          const double q = cf.q(k);
          const double bk = cf.b(k);
          if constexpr (ORDER == 1) {
            x1 = bk + q * x1;
          } else {
            const double nx1 = bk + q * x1 + cf.r(k) * x2; x2 = x1; x1 = nx1;
          }
          RPDE_T(res)[(DIR > 0) ? (par + 2 * i) : (C - 2 + par - 2 * i)] = x1;
        }
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int lo = tid * C;
#pragma unroll
    for (int i = 0; i < C; ++i)
      if (lo + i < n) dst[lo + i] = RPDE_T(res)[i];
  }
  RPDE_SYNC(blk);
}

struct CoefRec {   // generic table-driven recurrence
  const double* src; const double* pt; const double* qt; const double* rt;
  RPDE_DEV double b(int k) const { return pt ? pt[k] * src[k] : src[k]; }
  RPDE_DEV double q(int k) const { return qt[k]; }
  RPDE_DEV double r(int k) const { return rt[k]; }
};
struct CoefDiff {  // d_k = d_{k+2} + 2 (k+1) a_{k+1}
  const double* src; int n;
  RPDE_DEV double b(int k) const { return (k + 1 < n) ? 2.0 * (double)(k + 1) * src[k + 1] : 0.0; }
  RPDE_DEV double q(int) const { return 1.0; }
  RPDE_DEV double r(int) const { return 0.0; }
};

// ---------------------------------------------------------------------------------------------
// the interpreter
template <class Cfg>
RPDE_DEV void run_line_program(Blk& blk, const Program& pg) {
  constexpr int T = Cfg::T, EPT = Cfg::EPT;
  const int line = blk.line, comp = blk.comp;
  const int SL = pg.slot_len;
  double* lds = blk.lds;
  double* carry = lds + (size_t)pg.nslots * SL;
  for (int ip = 0; ip < pg.nops; ++ip) {
    const Op& op = pg.ops[ip];
    double* d = lds + (size_t)op.d * SL;
    const double* a = lds + (size_t)op.a * SL;
    const double* b = lds + (size_t)op.b * SL;
    const int n = op.n;
    const long toff = op.tabld * line;
    switch (op.code) {
      case OP_LOAD: {
        const ArrayRef& A = pg.arr[op.arr];
        const double* src = A.p + comp * A.coff + (long)line * A.ld;
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < SL) {
              double v = (k < n) ? op.s0 * src[(long)k * A.es] : 0.0;
              d[k] = op.acc ? (d[k] + v) : v;
            }
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_LOADX: {
        const ArrayRef& A = pg.arr[op.arr];
        const bool has0 = line < op.i1, has2 = line >= 2 && (line - 2) < op.i1;
        const double* s0p = A.p + comp * A.coff + (long)line * A.ld;
        const double* s2p = A.p + comp * A.coff + (long)(line - 2) * A.ld;
        const double c2 = has2 ? pg.tabs[op.tab][line - 2] : 0.0;
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < SL) {
              double v = 0.0;
              if (k < n) {
                if (has0) v = s0p[(long)k * A.es];
                if (has2) v += c2 * s2p[(long)k * A.es];
                v *= op.s0;
              }
              d[k] = op.acc ? (d[k] + v) : v;
            }
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_STORE: {
        const ArrayRef& A = pg.arr[op.arr];
        double* dstp = A.p + comp * A.coff + (long)line * A.ld;
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) {
              const long kk = op.i0 ? ((long)(k & 1) * op.i1 + (k >> 1)) : k;
              dstp[kk * A.es] = op.s0 * a[k];
            }
          }
        }
        RPDE_SYNC(blk);  // the next op may overwrite the slot
      } break;
      case OP_STEN: {
        const double* low = pg.tabs[op.tab] + toff;
        RPDE_TLS(blk, double, v, EPT);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) {
              double x = (k < n - 2) ? a[k] : 0.0;
              if (k >= 2) x += low[k - 2] * a[k - 2];
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
      case OP_MV3: {
        const double* t0 = pg.tabs[op.tab] + toff;
        const double* t1 = pg.tabs[op.tab + 1] + toff;
        const double* t2 = pg.tabs[op.tab + 2] + toff;
        RPDE_TLS(blk, double, v, EPT);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) {
              // input line has n + 2 entries; the +4 tap exists for k < n - 2 only (matvec.rs:215-226)
              double x = a[k] * t0[k] + a[k + 2] * t1[k];
              if (k < n - 2) x += a[k + 4] * t2[k];
              RPDE_T(v)[q] = x;
            }
          }
        }
        RPDE_SYNC(blk);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) d[k] = RPDE_T(v)[q];
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_CDIFF: {
        CoefDiff cf{a, n};
        scan_recurrence<Cfg, 1, -1>(blk, d, n, carry, cf);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) d[k] *= (k == 0) ? 0.5 * op.s0 : op.s0;
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_REC1: {
        CoefRec cf{a, op.tab >= 0 ? pg.tabs[op.tab] + toff : nullptr, pg.tabs[op.i0] + toff, nullptr};
        if (op.i1 > 0) scan_recurrence<Cfg, 1, +1>(blk, d, n, carry, cf);
        else scan_recurrence<Cfg, 1, -1>(blk, d, n, carry, cf);
      } break;
      case OP_REC2: {
        CoefRec cf{a, op.tab >= 0 ? pg.tabs[op.tab] + toff : nullptr, pg.tabs[op.i0] + toff,
                   pg.tabs[op.i1] + toff};
        scan_recurrence<Cfg, 2, -1>(blk, d, n, carry, cf);
      } break;
      case OP_DCT: {
        const double* pre = op.tab >= 0 ? pg.tabs[op.tab] : nullptr;
        const double* post = op.i0 >= 0 ? pg.tabs[op.i0] : nullptr;
        if (pg.fft_n > 0) dct1_lds<Cfg>(blk, d, n - 1, pre, post, pg.tabs[pg.tw], pg.tabs[pg.tw2]);
        else dct1_direct<Cfg>(blk, d, n - 1, pre, post, pg.tabs[pg.tw2]);
      } break;
      case OP_MUL: {
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) { const double v = op.s0 * a[k] * b[k]; d[k] = op.acc ? d[k] + v : v; }
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_AXPBY: {
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) d[k] = op.s0 * a[k] + op.s1 * b[k];
          }
        }
        RPDE_SYNC(blk);
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
        const double* t = pg.tabs[op.tab] + toff;
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int k = tid + q * T;
            if (k < n) d[k] = a[k] / t[k >> op.i0];
          }
        }
        RPDE_SYNC(blk);
      } break;
      case OP_RFFT_F:
        rfft_forward_lds<Cfg>(blk, d, n, pg.tabs[pg.tw], pg.tabs[pg.tw2]);
        break;
      case OP_RFFT_B:
        rfft_backward_lds<Cfg>(blk, d, n, pg.tabs[pg.tw], pg.tabs[pg.tw2]);
        break;
      case OP_CIK: {
        RPDE_TLS(blk, double, vr, EPT);
        RPDE_PHASE(blk, tid) {
#pragma unroll
          for (int q = 0; q < EPT; ++q) {
            const int e = tid + q * T;  // double index
            if (e < 2 * n) {
              const int k = e >> 1;
              const double f = op.s0 * (double)k;
              if (op.i0 == 1) RPDE_T(vr)[q] = (e & 1) ? f * a[e - 1] : -f * a[e + 1];
              else RPDE_T(vr)[q] = -(f * f) * a[e];
            }
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
}

}  // namespace rpde
