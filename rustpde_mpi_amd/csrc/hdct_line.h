// Whole-line Chebyshev transform, second form (round 3): DCT-I of N + 1 reals through a complex FFT of HALF the
// length the first form (dct_line.h) runs.  Same arguments (DctLineArgs / ConvLineArgs), same results up to round-off.
//
// dct_line.h packs the even extension of the line (2N reals) into an N-point complex FFT: 16 complex points per thread,
// 64 data registers, one real component at a time through the exchange buffer.  The even extension is also SYMMETRIC;
// using that as well leaves an N/2-point complex FFT:
//
//   y_j = (x_j + x_{N-j}) - 2 sin(pi j / N) (x_j - x_{N-j}),  0 < j < N,   y_0 = x_0 + x_N              (N reals)
//   Y = real FFT of y  =  split of the N/2-point complex FFT of z_i = y_{2i} + i y_{2i+1}
//   E_{2k} = Re Y_k,    E_{2k+1} = E_{2k-1} - Im Y_k,    E_1 = x_0 - x_N + 2 sum_{0<j<N/2} cos(pi j / N) (x_j - x_{N-j})
//
// (E_k = x_0 + (-1)^k x_N + 2 sum x_j cos(pi j k / N): rustdct's `process_dct1` under funspace's chebyshev transforms,
// src/field.rs:103-111.)  Half the butterflies, half the exchange traffic, 8 complex points = 32 data registers per
// thread, both components in the exchange buffer at once (half the barriers per exchange).  The price: the odd
// coefficients are a running sum over Im Y, so round-off grows like sqrt(N) eps instead of log(N) eps -- measured
// 7e-15 relative L2 at N = 4096 (first form: 3e-16); the step-level bar is 1e-10, the operator tests use 2e-12.
//
// N = 16 T points, T threads (256 for N = 4096: one wave per SIMD), M = N/2 = 8 T complex points, passes 8 x 8 x 8 x 4
// (N = 1024, one wave per line: 8 x 8 x 8; N = 256 in the emulation build: 8 x 8 x 2).  The running sum needs no second data layout: thread t owns k = t + u T,
// so row u is scanned across the threads (DPP inside a wave, wave totals through LDS) and the rows are chained.
#pragma once
#include "dct_line.h"

namespace rpde {

template <int N>
struct HdctGeom {
  static constexpr int T = N / 16;         // threads per line
  static constexpr int M = N / 2;          // complex FFT length = 8 T
  static constexpr int PL = M + M / 16;    // doubles per component plane (padded index pidx)
  static constexpr int NW = (T + 63) / 64; // waves
  static constexpr int SCR = 2 * PL + 8;   // scratch: [0, NW) partial sums of E_1, [8, 8 + 8 NW) wave totals of the row scans
                                           // (8 doubles behind the planes stay free: the padded line of rhs_line.h ends there)
  static constexpr int LDS = N + N / 16 + 64;   // doubles per line buffer
};
RPDE_HD inline size_t hdct_lds_doubles(int N) { return (size_t)N + N / 16 + 64; }

struct HdctNoFetch { RPDE_DEV void operator()(int) const {} };
// emit(tid, u, m, e0, e1): E_m = e0 and E_{m+1} = e1 for m = 2 (tid + u T), u < 8;  u = 8 (thread 0): E_N = e0
struct HdctStoreEmit {
  gmem_t dst; double sc;
  RPDE_DEV void operator()(int, int u, int m, double e0, double e1) const {
    if (u == 8) dst[m] = sc * e0;
    else ((gmem2_t)dst)[m >> 1] = dbl2{sc * e0, sc * e1};
  }
};

// staged: the input line is already in the buffer (x[m] at buffer index b = m + 2 -- at b + b / 16 when the transform takes the
// derivative or a table stencil, see `padx` below --, zeros around it) and a barrier has been passed.
// fetch(tid) is called once per thread ahead of the last barriers: the caller's chance to put the global loads its emit
// needs in flight.
// PAIR: the line runs in one half of a workgroup whose other half runs another transform of the same length
// (hdct_pair_line): every barrier is met by both halves, so the barriers of the derivative are passed without it too.
// MODE: what the caller knows at compile time about a.fwd / a.deriv / a.sten (kHdctFwd | kHdctDeriv | kHdctSten2 | kHdctSten1),
// -1 = nothing (every flag is tested at run time).  The step's launches have fixed flags: with them as constants the
// selects of the staging, of the pre-step's prefactors and of the output scaling (2/3 rule) and the code of the branches
// not taken disappear (round 5, profiles/r05_isa_census.md).
constexpr int kHdctFwd = 1, kHdctDeriv = 2, kHdctSten2 = 4, kHdctSten1 = 8;
RPDE_HD inline int hdct_mode_of(const DctLineArgs& a) {
  return (a.fwd ? kHdctFwd : 0) | (a.deriv ? kHdctDeriv : 0) | (a.sten == 2 ? kHdctSten2 : 0) | (a.sten == 1 ? kHdctSten1 : 0);
}
template <int N, class Fetch, class Emit, bool PAIR = false, int MODE = -1>
RPDE_DEV void hdct_core(Blk& blk, const DctLineArgs& a, bool staged, const Fetch& fetch, const Emit& emit) {
  const bool a_fwd = MODE >= 0 ? (MODE & kHdctFwd) != 0 : a.fwd != 0;
  const bool a_deriv = MODE >= 0 ? (MODE & kHdctDeriv) != 0 : a.deriv != 0;
  const int a_sten = MODE >= 0 ? ((MODE & kHdctSten2) ? 2 : (MODE & kHdctSten1) ? 1 : 0) : a.sten;
  using G = HdctGeom<N>;
  constexpr int T = G::T, M = G::M, PL = G::PL, NW = G::NW;
  static_assert(N == 4096 || N == 2048 || N == 1024 || N == 256, "N / 2 = 8 x 8 x 8 x 4, 8 x 8 x 8 x 2, 8 x 8 x 8 or 8 x 8 x 2");
  static_assert(T % 16 == 0, "padded indices assume T a multiple of 16");
  lds_t buf = (lds_t)blk.lds;
  lds2_t buf2 = (lds2_t)blk.lds;
  lds_t pre = buf, pim = buf + PL, scr = buf + G::SCR;
  tab_t tw = (tab_t)a.tw;
  tab_t tw2 = (tab_t)a.tw2;
  const int n_in = a.n_in;
  // x-layout of the staged line: x[m] at buffer index b = m + 2.  A transform that sweeps the line in chunks of 16 per thread
  // (derivative, table stencil) keeps it at the PADDED index b + b / 16: the chunks of neighbouring lanes are 17 doubles apart
  // (no bank conflicts).  Unpadded -- the form of rounds 3 and 4 -- the 32 chunk accesses of a derivative are 16-way conflicts:
  // more than half of the LDS cycles of S1 and of the convection terms (SQ_LDS_BANK_CONFLICT, profiles/r05_lds_counters.txt).
  // Everything else (the pure transforms, the forward transforms) keeps b: pairs stay 16-byte accesses.  The padded line ends
  // at N + 3 + (N + 3) / 16 < SCR.
  // (X(r + 16 m) = X(r) + P m with P = 17 or 16: one runtime term per thread and phase, everything else folds into constants.)
  const bool padx = MODE >= 0 ? (MODE & (kHdctDeriv | kHdctSten1)) != 0 : (a.deriv != 0 || a.sten == 1);
  const int padm = padx ? -1 : 0, P = padx ? 17 : 16;
  auto X = [padm](int b) { return b + ((b >> 4) & padm); };
  RPDE_TLS(blk, double, re, 8);
  RPDE_TLS(blk, double, im, 8);
  // the one table entry a thread needs, (cos, sin)(pi tid / N), goes out first: everything else (the twiddles of the
  // pre-step, of the split) is this angle turned by compile-time constants or doubled
  RPDE_TLS(blk, double, cs0, 2);
  RPDE_PHASE(blk, tid) { RPDE_T(cs0)[0] = tw2[2 * tid]; RPDE_T(cs0)[1] = tw2[2 * tid + 1]; }

  // ---- stage the line two doubles into the buffer: xs[m + 2] = x[m] (m < n_in), zeros in front and behind.  The
  // Dirichlet stencil (sten == 2: c_m = a_m - a_{m-2}) is applied on the way: the second pair of a thread is the first
  // pair of its neighbour, an L1 hit.
  if (!staged) {
    // through a buffer descriptor that ends behind the pair holding the last coefficient: a pair outside the line (the one
    // in front of it, everything behind it) reads zero -- no bounds test, no branch, one 32-bit offset per thread (round 5;
    // the flat form cost fourteen instructions per load: test, exec-mask branch, zeroed registers, 64-bit address)
    const RowBuf rb = row_buf(a.in + (long)blk.line * a.ldi, 8L * ((n_in + 1) & ~1));
    const bool dsten = a_sten == 2;
    RPDE_PHASE(blk, tid) {
      constexpr int QP = (N + 4 + 2 * T - 1) / (2 * T);   // pairs per thread: 2 T QP >= N + 4
      dbl2 v[QP], w[QP];
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int p = tid + q * T;                        // pair p = elements k = 2 p - 2, k + 1 at byte 16 (p - 1)
        v[q] = row_ld2(rb, 16 * (p - 1), 0);
        w[q] = dsten ? row_ld2(rb, 16 * (p - 2), 0) : dbl2{0.0, 0.0};
      }
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int p = tid + q * T, k = 2 * p - 2;
        dbl2 c = v[q], o = w[q];
        if (k + 1 >= n_in) c.y = 0.0;
        if (k - 1 >= n_in) o.y = 0.0;
        c.x -= o.x; c.y -= o.y;
        if (2 * p + 1 < N + 4) {
          if (padx) { const int xq = X(2 * tid) + P * (q * T / 8); buf[xq] = c.x; buf[xq + 1] = c.y; }   // b = 2 p is even: the pair stays inside its group of 16
          else buf2[p] = c;
        }
      }
    }
    RPDE_SYNC(blk);
  }

  // ---- table stencil and / or derivative: the orthonormal coefficients (then their derivative) replace the staged
  // line, thread t owning the contiguous chunk k = 16 t .. 16 t + 15 (the last thread also k = N)
  if (a_sten == 1) {
    tab_t low = (tab_t)a.low;
    RPDE_TLS(blk, double, c, 17);
    RPDE_PHASE(blk, tid) {
      const int k0 = 16 * tid;
      double xs[19], lw[17];
#pragma unroll
      for (int i = 0; i < 19; ++i) xs[i] = buf[P * tid + X(i)];               // xs[i] = a_{k0 + i - 2} at b = k0 + i
#pragma unroll
      for (int i = 0; i < 17; ++i) lw[i] = low[max(k0 + i - 2, 0)];
#pragma unroll
      for (int i = 0; i < 17; ++i) RPDE_T(c)[i] = xs[i + 2] + lw[i] * xs[i];  // c_k = a_k + low_{k-2} a_{k-2}; zeros outside
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
      const int k0 = 16 * tid;
#pragma unroll
      for (int i = 0; i < 16; ++i) buf[P * tid + X(i + 2)] = RPDE_T(c)[i];
      if (tid == T - 1) buf[X(N + 2)] = RPDE_T(c)[16];
    }
    RPDE_SYNC(blk);
  }
  if (PAIR || a_deriv) {
    // d_k = dscale * sum_{j > k, j + k odd} 2 j c_j, d_0 halved (the suffix sums of scan_cheb_diff, line_vm.h): thread t
    // owns the chunk lo = 16 (T - 1 - t), so that the carry flows from thread t - 1 to thread t
    const bool dv = a_deriv;
    RPDE_TLS(blk, double, zz, 16);
    RPDE_TLS(blk, double, vd, 2);
    RPDE_PHASE(blk, tid) {
      RPDE_T(vd)[0] = 0.0; RPDE_T(vd)[1] = 0.0;
      if (dv) {
        const int lo = (T - 1 - tid) * 16;
        double bb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) bb[i] = 2.0 * (double)(lo + i + 1) * buf[P * (T - 1 - tid) + X(i + 3)];   // 2 (k + 1) c_{k+1}, k + 1 <= N
#pragma unroll
        for (int par = 0; par < 2; ++par) {
          double z = 0.0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int e = 14 + par - 2 * i;
            z += bb[e];
            RPDE_T(zz)[e] = z;
          }
          RPDE_T(vd)[par] = z;
        }
      }
    }
#ifdef RPDE_EMU
    for (int par = 0; par < 2; ++par) {
      double run = 0.0;
      for (int t = 0; t < T; ++t) { const double mine = vd_st[(size_t)t * 2 + par]; vd_st[(size_t)t * 2 + par] = run; run += mine; }
    }
#else
    {
      const int tid = (int)threadIdx.x - blk.t0, lane = tid & 63, wave = tid >> 6;
      double v[2] = {vd[0], vd[1]};
      v[0] = sum_wave_scan(v[0]);
      v[1] = sum_wave_scan(v[1]);
      double S[2] = {0.0, 0.0};
      if constexpr (NW > 1) {
        if (lane == 63) { scr[8 + wave] = v[0]; scr[8 + NW + wave] = v[1]; }
        __syncthreads();
        for (int u = 0; u < wave; ++u) { S[0] += scr[8 + u]; S[1] += scr[8 + NW + u]; }
      }
#pragma unroll
      for (int par = 0; par < 2; ++par) vd[par] = dpp_f64z<0x138>(v[par]) + S[par];   // wave_shr:1
    }
#endif
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
      if (dv) {
        const int lo = (T - 1 - tid) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = lo + i;
          buf[P * (T - 1 - tid) + X(i + 2)] = (RPDE_T(zz)[i] + RPDE_T(vd)[i & 1]) * ((k == 0) ? 0.5 * a.dscale : a.dscale);
        }
        if (tid == 0) buf[X(N + 2)] = 0.0;       // d_N = 0
      }
    }
    RPDE_SYNC(blk);
  }

  // ---- y from x, in place: thread t owns the pairs (j, N - j), j = t + q T < M; thread 0 also j = M.  x carries the
  // backward pre-factor f_m = (-1)^m / 2 (both ends 1); j and N - j have the parity of t (T and N are even).
  RPDE_TLS(blk, double, e1p, 1);
  RPDE_PHASE(blk, tid) {
    // (cos, sin)(pi j / N), j = tid + q T: the table entry of q = 0 turned by q pi / 16 (compile-time constants)
    constexpr double kC32[8] = {1.0, 0.98078528040323043, 0.92387953251128674, 0.83146961230254524, 0.70710678118654757,
                                0.55557023301960218, 0.38268343236508978, 0.19509032201612825};
    constexpr double kS32[8] = {0.0, 0.19509032201612825, 0.38268343236508978, 0.55557023301960218, 0.70710678118654757,
                                0.83146961230254524, 0.92387953251128674, 0.98078528040323043};
    double cs[8], sn[8], xa[8], xb[8];
    const double c0 = RPDE_T(cs0)[0], s0 = RPDE_T(cs0)[1];
    const int xj = X(tid + 2), xn = X(N + 2 - tid);            // b = j + 2 and N - j + 2 for q = 0; q T is a multiple of 16
#pragma unroll
    for (int q = 0; q < 8; ++q) { xa[q] = buf[xj + P * (q * T / 16)]; xb[q] = buf[xn - P * (q * T / 16)]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) { cs[q] = c0 * kC32[q] - s0 * kS32[q]; sn[q] = s0 * kC32[q] + c0 * kS32[q]; }
    const double f = a_fwd ? 1.0 : ((tid & 1) ? -0.5 : 0.5);
    double e1 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = tid + q * T;
      const bool end = (q == 0 && tid == 0);               // j = 0 pairs the two ends of the line
      const double fa = (end && !a_fwd) ? 1.0 : f;
      const double A = fa * xa[q], B = fa * xb[q];
      const double s = A + B, d = A - B;
      const double t2 = 2.0 * sn[q] * d;
      buf[xj + P * (q * T / 16)] = end ? s : s - t2;
      if (!end) buf[xn - P * (q * T / 16)] = s + t2;
      e1 += end ? d : 2.0 * cs[q] * d;
    }
    if (tid == 0) buf[X(M + 2)] = (a_fwd ? 2.0 : 1.0) * buf[X(M + 2)];   // y_M = 2 f_M x_M, M even
    RPDE_T(e1p)[0] = e1;
  }
#ifndef RPDE_EMU
  {  // wave totals of E_1's sum wait in the scratch area until the last phase
    const int lane = ((int)threadIdx.x - blk.t0) & 63, wave = ((int)threadIdx.x - blk.t0) >> 6;
    const double tot = sum_wave_scan(e1p[0]);
    if (lane == ((T < 64) ? T - 1 : 63)) scr[wave] = tot;
  }
#endif
  RPDE_SYNC(blk);

  // ---- first pass: z_i = (y_{2i}, y_{2i+1}), i = tid + t T, radix 8, no twiddles
  RPDE_PHASE(blk, tid) {
    clds2_t y2 = (clds2_t)buf;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (padx) {                                          // the pair (y_{2i}, y_{2i+1}) at b = 2 i + 2: inside one group of 16
        const int q = X(2 * tid + 2) + P * (t * T / 8);
        RPDE_T(re)[t] = buf[q];
        RPDE_T(im)[t] = buf[q + 1];
      } else {
        const dbl2 z = y2[tid + t * T + 1];
        RPDE_T(re)[t] = z.x;
        RPDE_T(im)[t] = z.y;
      }
    }
    SmallDft<8>::run(RPDE_T(re), RPDE_T(im));
  }
  // exchange after a radix-R pass with Ns = 2^LGNS (B = 8 / R butterflies per thread, butterfly b of a thread lives
  // in the registers u = b + t B): output t of butterfly j goes to position j0 + t Ns; afterwards register u holds
  // position tid + u T.  Both components at once, one plane each.
  auto exchange = [&](auto LG, auto RR) {
    constexpr int LGNS = decltype(LG)::value, Ns = 1 << LGNS, R = decltype(RR)::value, B = 8 / R;
    constexpr int LGR = (R == 8) ? 3 : (R == 4) ? 2 : 1;
    RPDE_SYNC(blk);                                          // everybody has read what this overwrites
    RPDE_PHASE(blk, tid) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int jb = tid + b * T;
        const int j0 = ((jb >> LGNS) << (LGNS + LGR)) + (jb & (Ns - 1));
        const int b0 = pidx(j0);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const int p = (Ns >= 16) ? b0 + t * Ns + (t * Ns) / 16 : pidx(j0 + t * Ns);
          pre[p] = RPDE_T(re)[b + t * B];
          pim[p] = RPDE_T(im)[b + t * B];
        }
      }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
      const int b0 = pidx(tid);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        RPDE_T(re)[u] = pre[b0 + u * T + (u * T) / 16];
        RPDE_T(im)[u] = pim[b0 + u * T + (u * T) / 16];
      }
    }
  };
  // radix-R pass with twiddles W_M^(t k tstep), k = j mod Ns, tstep = M / (R Ns): powers of the table entry for t = 1
  // (the table holds W_N: W_M^m = W_N^(2m))
  auto pass = [&](auto LG, auto RR) {
    constexpr int LGNS = decltype(LG)::value, Ns = 1 << LGNS, R = decltype(RR)::value, B = 8 / R, tstep = M / (R * Ns);
    RPDE_PHASE(blk, tid) {
      double wc[B], ws[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int k = (tid + b * T) & (Ns - 1);
        wc[b] = tw[4 * (k * tstep)];
        ws[b] = tw[4 * (k * tstep) + 1];
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        double xr[R], xi[R];
#pragma unroll
        for (int t = 0; t < R; ++t) { xr[t] = RPDE_T(re)[b + t * B]; xi[t] = RPDE_T(im)[b + t * B]; }
        double cc = wc[b], cs = ws[b];
#pragma unroll
        for (int t = 1; t < R; ++t) {
          const double ar = xr[t], ai = xi[t];
          xr[t] = ar * cc - ai * cs;
          xi[t] = ar * cs + ai * cc;
          if (t < R - 1) { const double nc = cc * wc[b] - cs * ws[b], ns = cc * ws[b] + cs * wc[b]; cc = nc; cs = ns; }
        }
        SmallDft<R>::run(xr, xi);
#pragma unroll
        for (int t = 0; t < R; ++t) { RPDE_T(re)[b + t * B] = xr[t]; RPDE_T(im)[b + t * B] = xi[t]; }
      }
    }
  };
  using std::integral_constant;
  exchange(integral_constant<int, 0>{}, integral_constant<int, 8>{});
  pass(integral_constant<int, 3>{}, integral_constant<int, 8>{});
  exchange(integral_constant<int, 3>{}, integral_constant<int, 8>{});
  if constexpr (N == 4096) {
    pass(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    exchange(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    pass(integral_constant<int, 9>{}, integral_constant<int, 4>{});
  } else if constexpr (N == 2048) {   // 2049-point lines (round 6: the y-lines of BASELINE config 5)
    pass(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    exchange(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    pass(integral_constant<int, 9>{}, integral_constant<int, 2>{});
  } else if constexpr (N == 1024) {
    pass(integral_constant<int, 6>{}, integral_constant<int, 8>{});
  } else {
    pass(integral_constant<int, 6>{}, integral_constant<int, 2>{});
  }
  // now register u of thread t holds Z_k, k = t + u T (natural order)

  // ---- split of the real FFT: Y_k = ((Z_k + conj Z_{M-k}) - i w_k (Z_k - conj Z_{M-k})) / 2, w_k = exp(-2 pi i k / N);
  // the partner Z_{M-k} comes through the planes.  E_{2k} = Re Y_k, v_k = Im Y_k.
  RPDE_TLS(blk, double, cw, 2);
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int b0 = pidx(tid);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      pre[b0 + u * T + (u * T) / 16] = RPDE_T(re)[u];
      pim[b0 + u * T + (u * T) / 16] = RPDE_T(im)[u];
    }
    RPDE_T(cw)[0] = RPDE_T(cs0)[0] * RPDE_T(cs0)[0] - RPDE_T(cs0)[1] * RPDE_T(cs0)[1];   // (cos, sin)(2 pi tid / N): the doubled angle
    RPDE_T(cw)[1] = 2.0 * RPDE_T(cs0)[0] * RPDE_T(cs0)[1];
    fetch(tid);
  }
  RPDE_SYNC(blk);
  RPDE_TLS(blk, double, ev, 9);
  RPDE_TLS(blk, double, vv, 8);
  RPDE_PHASE(blk, tid) {
    // w_k for k = tid + u T: the entry of u = 0 turned by 2 pi u / 16
    constexpr double kC16[8] = {1.0, 0.92387953251128674, 0.70710678118654757, 0.38268343236508978, 0.0,
                                -0.38268343236508978, -0.70710678118654757, -0.92387953251128674};
    constexpr double kS16[8] = {0.0, 0.38268343236508978, 0.70710678118654757, 0.92387953251128674, 1.0,
                                0.92387953251128674, 0.70710678118654757, 0.38268343236508978};
    const int nb = -tid + ((-tid) >> 4);                       // pidx(-tid): M - k = (M - u T) - tid, M - u T a multiple of 16
    const double c0 = RPDE_T(cw)[0], s0 = RPDE_T(cw)[1];
    RPDE_T(ev)[8] = RPDE_T(re)[0] - RPDE_T(im)[0];             // thread 0: E_N = Re Y_M = Re Z_0 - Im Z_0
#pragma unroll
    for (int h = 0; h < 2; ++h) {                              // two batches of partner reads
      double br[4], bi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = 4 * h + i;
        const int p = (u == 0 && tid == 0) ? 0 : nb + (M - u * T) + (M - u * T) / 16;   // Z_M = Z_0
        br[i] = pre[p];
        bi[i] = pim[p];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = 4 * h + i;
        const double c = c0 * kC16[u] - s0 * kS16[u], sn = s0 * kC16[u] + c0 * kS16[u];
        const double ar = RPDE_T(re)[u], ai = RPDE_T(im)[u];
        const double Pr = ar + br[i], Pi = ai - bi[i], Qr = ar - br[i], Qi = ai + bi[i];
        RPDE_T(ev)[u] = 0.5 * (Pr + c * Qi - sn * Qr);
        RPDE_T(vv)[u] = 0.5 * (Pi - c * Qr - sn * Qi);
      }
#ifndef RPDE_EMU
      __builtin_amdgcn_sched_barrier(0);                       // keep the second batch of reads behind the first batch's arithmetic
#endif
    }
  }
  // ---- S_k = sum_{m <= k} v_m (v_0 = 0): row u = the k of one register index, scanned across the threads, the rows
  // chained; E_1 = the sum of the pre-step; E_{2k+1} = E_1 - S_k.  Results leave through emit as soon as they exist.
  const double fn = 1.0 / (double)N;
  auto finish = [&](int tid, int u, double e0, double eo) {
    const int m = 2 * (tid + u * T);
    if (a_fwd) {   // (-1)^k / N, both ends halved, the 2/3 rule
      e0 = (m < a.cut) ? e0 * ((u == 0 && tid == 0) ? 0.5 * fn : fn) : 0.0;
      eo = (m + 1 < a.cut) ? -eo * fn : 0.0;
    }
    emit(tid, u, m, e0, eo);
  };
  auto finish_end = [&](int tid, double en) {
    if (a_fwd) en = (N < a.cut) ? en * 0.5 * fn : 0.0;         // N is even
    emit(tid, 8, N, en, 0.0);
  };
#ifdef RPDE_EMU
  {
    double run = 0.0, e1 = 0.0;
    for (int u = 0; u < 8; ++u)
      for (int t = 0; t < T; ++t) { double& x = vv_st[(size_t)t * 8 + u]; run += x; x = run; }
    for (int t = 0; t < T; ++t) e1 += e1p_st[(size_t)t];
    (void)scr;
    RPDE_PHASE(blk, tid) {
      for (int u = 0; u < 8; ++u) finish(tid, u, RPDE_T(ev)[u], e1 - RPDE_T(vv)[u]);
      if (tid == 0) finish_end(tid, RPDE_T(ev)[8]);
    }
  }
#else
  {
    const int tid = (int)threadIdx.x - blk.t0, lane = tid & 63, wave = tid >> 6;
    const bool last = lane == ((T < 64) ? T - 1 : 63);
#pragma unroll
    for (int u = 0; u < 8; ++u) {                              // two row scans at a time (registers)
      vv[u] = sum_wave_scan(vv[u]);
      if (last) scr[8 + u * NW + wave] = vv[u];
      if (u & 1) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    double e1 = 0.0;
#pragma unroll
    for (int x = 0; x < NW; ++x) e1 += scr[x];
    double run = e1;                                           // E_1 - (rows before u) - (waves before this one in row u)
    // (a window of wave totals behind NW - 1 zeros, read at a wave-uniform address, would replace the selects below -- tried
    // in round 5: 50 vector instructions fewer per transform, but the S3 kernel spills 44 registers more with it; not kept)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      double before = 0.0, row = 0.0;
#pragma unroll
      for (int x = 0; x < NW; ++x) {
        const double tot = scr[8 + u * NW + x];
        before += (x < wave) ? tot : 0.0;
        row += tot;
      }
      finish(tid, u, ev[u], run - before - vv[u]);
      run -= row;
      if (u & 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (tid == 0) finish_end(tid, ev[8]);
  }
#endif
}

template <int N, int MODE = -1>
RPDE_DEV void hdct_bwd_line(Blk& blk, const DctLineArgs& a) {
  hdct_core<N, HdctNoFetch, HdctStoreEmit, false, MODE>(blk, a, false, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a.out + (long)blk.line * a.ldo), a.scale});
}

// the two flag combinations the time step launches (backward of a Dirichlet state line, and of its derivative) run code
// compiled for them (hdct_core's MODE); everything else the generic form
template <int N>
RPDE_DEV void hdct_bwd_line_by_mode(Blk& blk, const DctLineArgs& a) {
#ifdef RPDE_EMU
  const int mode = hdct_mode_of(a);
#else
  const int mode = __builtin_amdgcn_readfirstlane(hdct_mode_of(a));
#endif
  if (mode == kHdctSten2) hdct_bwd_line<N, kHdctSten2>(blk, a);
  else if (mode == (kHdctSten2 | kHdctDeriv)) hdct_bwd_line<N, kHdctSten2 | kHdctDeriv>(blk, a);
  else hdct_bwd_line<N>(blk, a);
}

// S1 of the step: the physical values AND the physical derivative of one state line (a0: the series, a1: deriv = 1 of the
// same input) by ONE workgroup of 2 T threads.  The line is loaded once (the Dirichlet stencil on the way) into two
// exchange buffers; then the two halves of the workgroup run the two transforms side by side, each in its buffer --
// half the global reads of two transforms in a row, a latency chain of one transform instead of two.
template <int N>
RPDE_DEV void hdct_pair_line(int line, double* lds, const DctLineArgs& a0, const DctLineArgs& a1) {
  constexpr int T = N / 16, T2 = 2 * T;
  const long LB = (long)hdct_lds_doubles(N);
  {
    Blk all{line, 0, T2, lds};
    lds2_t bufa = (lds2_t)lds;
    lds_t la = (lds_t)lds, lb = (lds_t)(lds + LB);
    const bool dsten = a0.sten == 2;
    const int n_in = a0.n_in;
    const RowBuf rb = row_buf(a0.in + (long)line * a0.ldi, 8L * ((n_in + 1) & ~1));   // as in hdct_core: outside the line reads zero
    RPDE_PHASE(all, tid) {
      constexpr int QP = (N + 4 + 2 * T2 - 1) / (2 * T2);   // pairs per thread: 2 T2 QP >= N + 4
      dbl2 v[QP], w[QP];
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int p = tid + q * T2;
        v[q] = row_ld2(rb, 16 * (p - 1), 0);
        w[q] = dsten ? row_ld2(rb, 16 * (p - 2), 0) : dbl2{0.0, 0.0};
      }
#pragma unroll
      for (int q = 0; q < QP; ++q) {
        const int p = tid + q * T2, k = 2 * p - 2;
        dbl2 c = v[q], o = w[q];
        if (k + 1 >= n_in) c.y = 0.0;
        if (k - 1 >= n_in) o.y = 0.0;
        c.x -= o.x; c.y -= o.y;
        if (2 * p + 1 < N + 4) {
          // the derivative's buffer (and both under a table stencil) in the padded x-layout of hdct_core
          const int xq = 2 * p + ((2 * p) >> 4);
          if (a0.sten == 1) { la[xq] = c.x; la[xq + 1] = c.y; } else bufa[p] = c;
          lb[xq] = c.x; lb[xq + 1] = c.y;
        }
      }
    }
    RPDE_SYNC(all);
  }
#ifdef RPDE_EMU
  {
    Blk b0{line, 0, T, lds};
    Blk b1{line, 0, T, lds + LB};
    if (a0.sten != 1) {
      hdct_core<N, HdctNoFetch, HdctStoreEmit, true, 0>(b0, a0, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a0.out + (long)line * a0.ldo), a0.scale});
      hdct_core<N, HdctNoFetch, HdctStoreEmit, true, kHdctDeriv>(b1, a1, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a1.out + (long)line * a1.ldo), a1.scale});
    } else {
      hdct_core<N, HdctNoFetch, HdctStoreEmit, true>(b0, a0, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a0.out + (long)line * a0.ldo), a0.scale});
      hdct_core<N, HdctNoFetch, HdctStoreEmit, true>(b1, a1, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a1.out + (long)line * a1.ldo), a1.scale});
    }
  }
#else
  {
    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >= T ? 1 : 0);
    Blk blk{line, 0, T, lds + half * LB, nullptr, 0, half * T};
    // hdct_pair_ok: both backward, the first the series, the second its derivative; the stencil is applied by the staging above
    if (a0.sten != 1) {
      if (half) hdct_core<N, HdctNoFetch, HdctStoreEmit, true, kHdctDeriv>(blk, a1, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a1.out + (long)line * a1.ldo), a1.scale});
      else hdct_core<N, HdctNoFetch, HdctStoreEmit, true, 0>(blk, a0, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a0.out + (long)line * a0.ldo), a0.scale});
    } else {
      const DctLineArgs& a = half ? a1 : a0;
      hdct_core<N, HdctNoFetch, HdctStoreEmit, true>(blk, a, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(a.out + (long)line * a.ldo), a.scale});
    }
  }
#endif
}
RPDE_HD inline bool hdct_pair_ok(const DctLineArgs& a0, const DctLineArgs& a1) {
  return a0.in == a1.in && a0.ldi == a1.ldi && a0.n_in == a1.n_in && a0.sten == a1.sten && a0.low == a1.low && a0.N == a1.N &&
         a0.nlines == a1.nlines && !a0.fwd && !a1.fwd && !a0.deriv && a1.deriv;
}

// One y-line of a convection term on this core (see conv_line in dct_line.h for the mathematics): the physical factors
// u, v (, bx, by) of a thread's 17 points are fetched while the transform that needs them is still in its last passes.
template <int N, int MEAN = 0>
RPDE_DEV void hconv_line(Blk& blk, const ConvLineArgs& c) {
  constexpr int T = N / 16;
  lds_t buf = (lds_t)blk.lds;
  lds2_t buf2 = (lds2_t)blk.lds;
  const long off = (long)blk.line * c.ld, offl = (long)blk.line * conv_lift_pitch(c);
  const bool lift = c.bx != nullptr;
#ifdef RPDE_CONV_GUARD_OFF
  const bool on = true;                // (A/B build: the form of rounds 4 - 6, 400 - 470 registers)
#else
  const bool on = c.up != nullptr;     // always true: see conv_line (dct_line.h) -- the factor loads behind a run-time condition: 154 registers
#endif
  RPDE_TLS(blk, double, acc, 17);
  DctLineArgs a1{c.fx, c.ld, c.n_in, nullptr, 0, c.nlines, N, 2, c.tw, c.tw2, 1.0};
  // (the pairs m = 2 (tid + u T) of the physical factors through buffer descriptors: line_vm.h RowBuf, as conv_line)
  const long rowb = 8L * (N + 2);
  {
    cgmem_t up = (cgmem_t)(c.up + off), bx = (cgmem_t)(lift ? c.bx + offl : c.up + off);
    const RowBuf rup = row_buf(c.up + off, rowb), rbx = row_buf(lift ? c.bx + offl : c.up + off, rowb);
    cgmem_t um = (cgmem_t)((MEAN ? c.um : c.up) + off);
    const RowBuf rum = row_buf((MEAN ? c.um : c.up) + off, rowb);
    hdct_core<N>(blk, a1, false, HdctNoFetch{}, [&](int tid, int u, int m, double e0, double e1) {
      if (!on) { if (u == 8) RPDE_T(acc)[16] = e0; else { RPDE_T(acc)[2 * u] = e0; RPDE_T(acc)[2 * u + 1] = e1; } return; }
      if constexpr (MEAN == 2) {   // (um + up) (d/dx f + bx) (nonlin_eq.rs:59-134)
        if (u == 8) { RPDE_T(acc)[16] = (um[m] + up[m]) * (e0 + bx[m]); return; }
        const dbl2 f = row_ld2(rup, 16 * tid, 16 * u * T), g = row_ld2(rbx, 16 * tid, 16 * u * T), h = row_ld2(rum, 16 * tid, 16 * u * T);
        RPDE_T(acc)[2 * u] = (h.x + f.x) * (e0 + g.x);
        RPDE_T(acc)[2 * u + 1] = (h.y + f.y) * (e1 + g.y);
        return;
      }
      if constexpr (MEAN == 1 || MEAN == 3) {   // um d/dx f + up bx (lnse_eq.rs:59-110; MEAN 3: um = -U, lnse_adj_eq.rs:16-94)
        if (u == 8) { RPDE_T(acc)[16] = um[m] * e0 + up[m] * bx[m]; return; }
        const dbl2 f = row_ld2(rup, 16 * tid, 16 * u * T), g = row_ld2(rbx, 16 * tid, 16 * u * T), h = row_ld2(rum, 16 * tid, 16 * u * T);
        RPDE_T(acc)[2 * u] = h.x * e0 + f.x * g.x;
        RPDE_T(acc)[2 * u + 1] = h.y * e1 + f.y * g.y;
        return;
      }
      if (u == 8) { RPDE_T(acc)[16] = up[m] * (lift ? e0 + bx[m] : e0); return; }
      const dbl2 f = row_ld2(rup, 16 * tid, 16 * u * T);
      dbl2 g = dbl2{0.0, 0.0};
      if (lift) g = row_ld2(rbx, 16 * tid, 16 * u * T);
      RPDE_T(acc)[2 * u] = f.x * (e0 + g.x);
      RPDE_T(acc)[2 * u + 1] = f.y * (e1 + g.y);
    });
  }
  RPDE_SYNC(blk);
  DctLineArgs a2 = a1;
  a2.in = c.f0; a2.deriv = 1; a2.dscale = c.dscale;
  {
    cgmem_t vp = (cgmem_t)(c.vp + off), by = (cgmem_t)(lift ? c.by + offl : c.vp + off);
    const RowBuf rvp = row_buf(c.vp + off, rowb), rby = row_buf(lift ? c.by + offl : c.vp + off, rowb);
    cgmem_t vm = (cgmem_t)((MEAN ? c.vm : c.vp) + off);
    const RowBuf rvm = row_buf((MEAN ? c.vm : c.vp) + off, rowb);
    hdct_core<N>(blk, a2, false, HdctNoFetch{}, [&](int tid, int u, int m, double e0, double e1) {
      if (!on) { if (u == 8) RPDE_T(acc)[16] += e0; else { RPDE_T(acc)[2 * u] += e0; RPDE_T(acc)[2 * u + 1] += e1; } return; }
      if constexpr (MEAN == 2) {   // + (vm + vp) (d/dy f + by)
        if (u == 8) { RPDE_T(acc)[16] += (vm[m] + vp[m]) * (e0 + by[m]); return; }
        const dbl2 f = row_ld2(rvp, 16 * tid, 16 * u * T), g = row_ld2(rby, 16 * tid, 16 * u * T), h = row_ld2(rvm, 16 * tid, 16 * u * T);
        RPDE_T(acc)[2 * u] += (h.x + f.x) * (e0 + g.x);
        RPDE_T(acc)[2 * u + 1] += (h.y + f.y) * (e1 + g.y);
        return;
      }
      if constexpr (MEAN == 3) {   // + vm d/dy f + vp by + tp cz (the adjoint term's third mean-gradient product)
        cgmem_t tp = (cgmem_t)(c.tp + off), cz = (cgmem_t)(c.cz + off);
        const RowBuf rtp = row_buf(c.tp + off, rowb), rcz = row_buf(c.cz + off, rowb);
        if (u == 8) { RPDE_T(acc)[16] += vm[m] * e0 + vp[m] * by[m] + tp[m] * cz[m]; return; }
        const dbl2 f = row_ld2(rvp, 16 * tid, 16 * u * T), g = row_ld2(rby, 16 * tid, 16 * u * T), h = row_ld2(rvm, 16 * tid, 16 * u * T);
        const dbl2 p = row_ld2(rtp, 16 * tid, 16 * u * T), q = row_ld2(rcz, 16 * tid, 16 * u * T);
        RPDE_T(acc)[2 * u] += h.x * e0 + f.x * g.x + p.x * q.x;
        RPDE_T(acc)[2 * u + 1] += h.y * e1 + f.y * g.y + p.y * q.y;
        return;
      }
      if constexpr (MEAN == 1) {   // + vm d/dy f + vp by
        if (u == 8) { RPDE_T(acc)[16] += vm[m] * e0 + vp[m] * by[m]; return; }
        const dbl2 f = row_ld2(rvp, 16 * tid, 16 * u * T), g = row_ld2(rby, 16 * tid, 16 * u * T), h = row_ld2(rvm, 16 * tid, 16 * u * T);
        RPDE_T(acc)[2 * u] += h.x * e0 + f.x * g.x;
        RPDE_T(acc)[2 * u + 1] += h.y * e1 + f.y * g.y;
        return;
      }
      if (u == 8) { RPDE_T(acc)[16] += vp[m] * (lift ? e0 + by[m] : e0); return; }
      const dbl2 f = row_ld2(rvp, 16 * tid, 16 * u * T);
      dbl2 g = dbl2{0.0, 0.0};
      if (lift) g = row_ld2(rby, 16 * tid, 16 * u * T);
      RPDE_T(acc)[2 * u] += f.x * (e0 + g.x);
      RPDE_T(acc)[2 * u + 1] += f.y * (e1 + g.y);
    });
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {   // the sum as the staged input line of the forward transform: x[m] at buf[m + 2]
#pragma unroll
    for (int u = 0; u < 8; ++u) buf2[tid + u * T + 1] = dbl2{RPDE_T(acc)[2 * u], RPDE_T(acc)[2 * u + 1]};
    if (tid == 0) { buf[N + 2] = RPDE_T(acc)[16]; buf[0] = 0.0; buf[1] = 0.0; buf[N + 3] = 0.0; }
  }
  RPDE_SYNC(blk);
  DctLineArgs a3{nullptr, 0, N + 1, nullptr, 0, c.nlines, N, 0, c.tw, c.tw2, 1.0};
  a3.fwd = 1; a3.cut = c.cut;
  hdct_core<N>(blk, a3, true, HdctNoFetch{}, HdctStoreEmit{(gmem_t)(c.out + (long)blk.line * c.ldo), 1.0});
}

}  // namespace rpde
