// Whole-line real FFTs of the periodic step (round 4): the Fourier axis of Navier2D::new_periodic (funspace fourier_r2c,
// src/navier_stokes/navier.rs:336-428, called through src/field.rs:103-110) on lines of N = 4096 / 1024 reals that stay
// in registers, like the Chebyshev lines of hdct_line.h: T = N / 16 threads, M = N / 2 = 8 T complex points, eight per
// thread, radix passes 8 x 8 x 8 x 4 (N = 4096), 8 x 8 x 8 (1024), 8 x 8 x 8 x 8 (8192), 8 x 8 x 8 x 8 x 2 (16384: 1024 threads,
// 140 KB of LDS, one workgroup per CU), 8 x 8 x 2 (256: emulation build), both components in the exchange buffer at once.
//
//   backward (complex -> real, rustfft inverse / n: oracle/bases.py:131, the line program's OP_RFFT_B):
//     Z_k = ((X_k + conj X_{M-k}) + i conj(w_k) (X_k - conj X_{M-k})) / 2,  w_k = exp(-2 pi i k / N),  0 <= k < M
//     (y_{2i}, y_{2i+1}) = conj(FFT_M(conj Z))_i / M
//     The thread that transforms Z_k owns k = t + u T: its own X_k and the partner X_{M-k} come straight from global memory
//     (both runs are contiguous across the lanes), no staging.  `cik`: X_k times i k kscale first (the x-derivative).
//   forward (real -> complex, unnormalised: oracle/bases.py:121, OP_RFFT_F):
//     z_i = (y_{2i}, y_{2i+1}) from global memory, Z = FFT_M(z), Y_k = ((Z_k + conj Z_{M-k}) - i w_k (Z_k - conj Z_{M-k})) / 2
//     (the partner through the planes), Y_M = Re Z_0 - Im Z_0.  Results leave through `emit`: the right-hand side of a
//     Helmholtz solve is assembled where the coefficient exists (S3 of the periodic step).
// Same source for the HIP kernels and the host emulation.
#pragma once
#include "hdct_line.h"

namespace rpde {

struct RfftLineArgs {
  const double* in; long ldi;      // backward: lines of N / 2 + 1 interleaved complex numbers; forward: lines of N reals
  double* out; long ldo;           // backward: lines of N reals; forward: lines of N / 2 + 1 interleaved complex numbers
  int nlines;
  int N;                           // reals per line: 16384, 8192, 4096 or 1024 (256: emulation build)
  const double* tw;                // fft_twiddles(N / 2): (cos, -sin)(2 pi k / (N / 2))             (AxisTables::tw of the Fourier axis)
  const double* tw2;               // rfft_split_twiddles(N): (cos, sin)(2 pi k / N), k = 0 .. N / 2   (AxisTables::tw2)
  double scale = 1.0;              // multiplies the result (backward: on top of 1 / N)
  int cik = 0; double kscale = 0;  // backward: transform i k kscale X_k instead of X_k (OP_CIK with power 1)
};
RPDE_HD inline bool rfft_line_ok(const RfftLineArgs& a) {
  return (a.N == 16384 || a.N == 8192 || a.N == 4096 || a.N == 1024 || a.N == 256) && (((size_t)a.in | (size_t)a.out) & 15) == 0 && (a.ldi & 1) == 0 && (a.ldo & 1) == 0;
}

// the passes behind the first radix-8 pass of an M = N / 2 point FFT (hdct_core's, with the twiddles of a Fourier axis:
// tw holds W_M itself).  re / im: the eight points of every thread (thread-local storage of the caller).
template <int N>
RPDE_DEV void rfft_passes(Blk& blk, double* re_b, double* im_b, lds_t pre, lds_t pim, tab_t tw) {
  constexpr int T = N / 16, M = N / 2;
  static_assert(N == 16384 || N == 8192 || N == 4096 || N == 1024 || N == 256, "N / 2 = 8 x 8 x 8 x 8 x 2, 8 x 8 x 8 x 8, 8 x 8 x 8 x 4, 8 x 8 x 8 or 8 x 8 x 2");
  auto exchange = [&](auto LG, auto RR) {
    constexpr int LGNS = decltype(LG)::value, Ns = 1 << LGNS, R = decltype(RR)::value, B = 8 / R;
    constexpr int LGR = (R == 8) ? 3 : (R == 4) ? 2 : 1;
    RPDE_SYNC(blk);                                          // everybody has read what this overwrites
    RPDE_PHASE(blk, tid) {
      const double* re = RPDE_TPK(re_b, 8);
      const double* im = RPDE_TPK(im_b, 8);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int jb = tid + b * T;
        const int j0 = ((jb >> LGNS) << (LGNS + LGR)) + (jb & (Ns - 1));
        const int b0 = pidx(j0);
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const int p = (Ns >= 16) ? b0 + t * Ns + (t * Ns) / 16 : pidx(j0 + t * Ns);
          pre[p] = re[b + t * B];
          pim[p] = im[b + t * B];
        }
      }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
      double* re = RPDE_TPK(re_b, 8);
      double* im = RPDE_TPK(im_b, 8);
      const int b0 = pidx(tid);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        re[u] = pre[b0 + u * T + (u * T) / 16];
        im[u] = pim[b0 + u * T + (u * T) / 16];
      }
    }
  };
  auto pass = [&](auto LG, auto RR) {
    constexpr int LGNS = decltype(LG)::value, Ns = 1 << LGNS, R = decltype(RR)::value, B = 8 / R, tstep = M / (R * Ns);
    RPDE_PHASE(blk, tid) {
      double* re = RPDE_TPK(re_b, 8);
      double* im = RPDE_TPK(im_b, 8);
      double wc[B], ws[B];
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int k = (tid + b * T) & (Ns - 1);
        wc[b] = tw[2 * (k * tstep)];
        ws[b] = tw[2 * (k * tstep) + 1];
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        double xr[R], xi[R];
#pragma unroll
        for (int t = 0; t < R; ++t) { xr[t] = re[b + t * B]; xi[t] = im[b + t * B]; }
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
        for (int t = 0; t < R; ++t) { re[b + t * B] = xr[t]; im[b + t * B] = xi[t]; }
      }
    }
  };
  using std::integral_constant;
  exchange(integral_constant<int, 0>{}, integral_constant<int, 8>{});
  pass(integral_constant<int, 3>{}, integral_constant<int, 8>{});
  exchange(integral_constant<int, 3>{}, integral_constant<int, 8>{});
  if constexpr (N == 16384 || N == 8192) {
    pass(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    exchange(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    pass(integral_constant<int, 9>{}, integral_constant<int, 8>{});
    if constexpr (N == 16384) {
      exchange(integral_constant<int, 9>{}, integral_constant<int, 8>{});
      pass(integral_constant<int, 12>{}, integral_constant<int, 2>{});
    }
  } else if constexpr (N == 4096) {
    pass(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    exchange(integral_constant<int, 6>{}, integral_constant<int, 8>{});
    pass(integral_constant<int, 9>{}, integral_constant<int, 4>{});
  } else if constexpr (N == 1024) {
    pass(integral_constant<int, 6>{}, integral_constant<int, 8>{});
  } else {
    pass(integral_constant<int, 6>{}, integral_constant<int, 2>{});
  }
  // now register u of thread t holds point t + u T of the result (natural order)
}

// emit(tid, u, i, y0, y1): y_{2i} = y0, y_{2i+1} = y1 for i = tid + u T, u < 8
struct RfftStoreReal {
  gmem_t dst;
  RPDE_DEV void operator()(int, int, int i, double y0, double y1) const { ((gmem2_t)dst)[i] = dbl2{y0, y1}; }
};

template <int N, class Emit>
RPDE_DEV void rfft_bwd_core(Blk& blk, const RfftLineArgs& a, const Emit& emit) {
  constexpr int T = N / 16, M = N / 2, PL = M + M / 16;
  lds_t buf = (lds_t)blk.lds;
  lds_t pre = buf, pim = buf + PL;
  tab_t tw = (tab_t)a.tw;
  tab_t tw2 = (tab_t)a.tw2;
  cgmem2_t src = (cgmem2_t)(a.in + (long)blk.line * a.ldi);
  RPDE_TLS(blk, double, re, 8);
  RPDE_TLS(blk, double, im, 8);
  RPDE_PHASE(blk, tid) {
    dbl2 xa[8], xb[8], cs[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = tid + u * T;
      xa[u] = src[k];
      xb[u] = src[M - k];
      cs[u] = ((cgmem2_t)tw2)[k];
    }
    const double f = 0.5 * a.scale / (double)M;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = tid + u * T;
      double ar = xa[u].x, ai = xa[u].y, br = xb[u].x, bi = -xb[u].y;   // b = conj X_{M-k}
      if (a.cik) {                                                       // X -> i k kscale X; conj(i g X_{M-k}) = -i g conj X_{M-k}
        const double fa = a.kscale * (double)k, fb = a.kscale * (double)(M - k);
        const double tr = -fa * ai, ti = fa * ar;
        ar = tr; ai = ti;
        const double sr = fb * bi, si = -fb * br;
        br = sr; bi = si;
      }
      if (k == 0) { ai = 0.0; bi = 0.0; }                                // imaginary parts of X_0 and X_M are ignored (irfft)
      const double c = cs[u].x, s = cs[u].y;                             // conj(w_k) = c + i s
      const double sr = ar + br, si = ai + bi, dr = ar - br, di = ai - bi;
      const double er = sr + (-(c * di) - s * dr);
      const double ei = si + (c * dr - s * di);
      RPDE_T(re)[u] = f * er;                                            // conj Z_k, scaled
      RPDE_T(im)[u] = -f * ei;
    }
    SmallDft<8>::run(RPDE_T(re), RPDE_T(im));
  }
  rfft_passes<N>(blk, RPDE_TLS_PTR(re), RPDE_TLS_PTR(im), pre, pim, tw);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int u = 0; u < 8; ++u) emit(tid, u, tid + u * T, RPDE_T(re)[u], -RPDE_T(im)[u]);
  }
}

template <int N>
RPDE_DEV void rfft_bwd_line(Blk& blk, const RfftLineArgs& a) {
  rfft_bwd_core<N>(blk, a, RfftStoreReal{(gmem_t)(a.out + (long)blk.line * a.ldo)});
}

// emit(tid, u, k, yr, yi): Y_k for k = tid + u T (u < 8); u = 8 (thread 0): k = M
struct RfftStoreCplx {
  gmem_t dst; double sc;
  RPDE_DEV void operator()(int, int, int k, double yr, double yi) const { ((gmem2_t)dst)[k] = dbl2{sc * yr, sc * yi}; }
};

template <int N, class Emit>
RPDE_DEV void rfft_fwd_core(Blk& blk, const RfftLineArgs& a, const Emit& emit) {
  constexpr int T = N / 16, M = N / 2, PL = M + M / 16;
  lds_t buf = (lds_t)blk.lds;
  lds_t pre = buf, pim = buf + PL;
  tab_t tw = (tab_t)a.tw;
  tab_t tw2 = (tab_t)a.tw2;
  cgmem2_t src = (cgmem2_t)(a.in + (long)blk.line * a.ldi);
  RPDE_TLS(blk, double, re, 8);
  RPDE_TLS(blk, double, im, 8);
  RPDE_PHASE(blk, tid) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const dbl2 z = src[tid + u * T];
      RPDE_T(re)[u] = z.x;
      RPDE_T(im)[u] = z.y;
    }
    SmallDft<8>::run(RPDE_T(re), RPDE_T(im));
  }
  rfft_passes<N>(blk, RPDE_TLS_PTR(re), RPDE_TLS_PTR(im), pre, pim, tw);
  RPDE_TLS(blk, double, cw, 16);
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int b0 = pidx(tid);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      pre[b0 + u * T + (u * T) / 16] = RPDE_T(re)[u];
      pim[b0 + u * T + (u * T) / 16] = RPDE_T(im)[u];
      const dbl2 w = ((cgmem2_t)tw2)[tid + u * T];
      RPDE_T(cw)[2 * u] = w.x; RPDE_T(cw)[2 * u + 1] = w.y;
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int nb = -tid + ((-tid) >> 4);                       // pidx(-tid): M - k = (M - u T) - tid, M - u T a multiple of 16
    if (tid == 0) emit(tid, 8, M, RPDE_T(re)[0] - RPDE_T(im)[0], 0.0);   // Y_M = Re Z_0 - Im Z_0
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int p = (u == 0 && tid == 0) ? 0 : nb + (M - u * T) + (M - u * T) / 16;   // Z_M = Z_0
      const double br = pre[p], bi = pim[p];
      const double c = RPDE_T(cw)[2 * u], sn = RPDE_T(cw)[2 * u + 1];    // w_k = c - i sn
      const double ar = RPDE_T(re)[u], ai = RPDE_T(im)[u];
      const double Pr = ar + br, Pi = ai - bi, Qr = ar - br, Qi = ai + bi;
      emit(tid, u, tid + u * T, 0.5 * (Pr + c * Qi - sn * Qr), 0.5 * (Pi - c * Qr - sn * Qi));
    }
  }
}

template <int N>
RPDE_DEV void rfft_fwd_line(Blk& blk, const RfftLineArgs& a) {
  rfft_fwd_core<N>(blk, a, RfftStoreCplx{(gmem_t)(a.out + (long)blk.line * a.ldo), a.scale});
}

// S1 of the periodic step: the physical values AND the physical x-derivative of one spectral state line by ONE workgroup of
// 2 T threads, one transform per half (a1 = a0 with cik = 1), as hdct_pair_line does for Chebyshev lines.
template <int N>
RPDE_DEV void rfft_pair_line(int line, double* lds, const RfftLineArgs& a0, const RfftLineArgs& a1) {
  constexpr int T = N / 16;
  const long LB = (long)hdct_lds_doubles(N);
#ifdef RPDE_EMU
  { Blk b0{line, 0, T, lds}; rfft_bwd_line<N>(b0, a0); }
  { Blk b1{line, 0, T, lds + LB}; rfft_bwd_line<N>(b1, a1); }
#else
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >= T ? 1 : 0);
  Blk blk{line, 0, T, lds + half * LB, nullptr, 0, half * T};
  rfft_bwd_line<N>(blk, half ? a1 : a0);
#endif
}

// The same two transforms one after the other in ONE buffer: lines of 8192 / 16384 reals, whose exchange buffer (70 / 140 KB)
// does not fit twice into the LDS of a CU next to a second workgroup.  The line is read twice (the second time from the L2).
template <int N>
RPDE_DEV void rfft_seq2_line(Blk& blk, const RfftLineArgs& a0, const RfftLineArgs& a1) {
  rfft_bwd_line<N>(blk, a0);
  RPDE_SYNC(blk);
  rfft_bwd_line<N>(blk, a1);
}

// S3 of the periodic step (navier_eq.rs solve_velx / solve_vely / solve_temp with the Fourier axis: the x part of HholtzAdi is
// a division by a diagonal, src/solver/hholtz_adi.rs:149-169 with sdma.rs:36-46): forward real FFT of a line of the
// convection term, 2/3 rule, right-hand side, diagonal factor -- one kernel, the coefficient never leaves its register.
//   rhs_k = -dt conv_k [k < cut] + (S_y state)_k + extra_k,  out_k = rhs_k / diag_k
//   which 0 (velx): extra = i k pk P_k                      (pk = -dt / sx: minus dt times the x-derivative of the pressure)
//   which 1 (vely): extra = -dt GY_k + dt (S_y temp)_k + dt TBC_k
//   which 2 (temp): extra = ctbc TBC2_k
// S_y v: rows j and j - 2 of a YX array with the y stencil, (S_y v)_j = v_j + low[j-2] v_{j-2} (low == nullptr: the array holds
// orthonormal rows already -- the "hc" temperature).
struct FourRhsArgs {
  RfftLineArgs f;                  // in: lines of the convection term (N reals), out: right-hand sides (N / 2 + 1 complex)
  int which, cut;
  double dt;
  int line0;                       // global row index of line 0 (pencil-sharded engines)
  int rows;                        // rows of the composite y space (row j exists for j < rows)
  long ld;                         // pitch of state / p / gy / tsrc / tbc (doubles)
  const double* state; const double* low;
  const double* p; double pk;      // which 0
  const double* gy;                // which 1
  const double* tsrc; const double* tlow;   // which 1
  const double* tbc; double ctbc;  // which 1: TBC with ctbc = dt; which 2: TBC2 with ctbc = dt * ka
  const double* diag;              // [N / 2 + 1]
  int tbc_cols = -1;               // >= 0: only the first tbc_cols doubles of a tbc row can be non-zero (a lift that does not depend on
                                   // x: mode 0 only, its Laplacian possibly nothing): the other modes are not read (Navier2DEngine::analyse_lift)
};
RPDE_HD inline bool four_rhs_ok(const FourRhsArgs& a) {
  return rfft_line_ok(a.f) && (a.ld & 1) == 0 && (((size_t)a.state | (size_t)a.p | (size_t)a.gy | (size_t)a.tsrc | (size_t)a.tbc) & 15) == 0;
}

template <int N>
RPDE_DEV void four_rhs_line(Blk& blk, const FourRhsArgs& a) {
  const int line = blk.line, gline = a.line0 + line;
  const long off = (long)line * a.ld, off2 = (long)(line - 2) * a.ld;
  const bool has0 = gline < a.rows, has2 = gline >= 2 && gline - 2 < a.rows;
  tab_t lowt = (tab_t)a.low;
  tab_t tlowt = (tab_t)a.tlow;
  const double c2 = (has2 && a.low) ? lowt[gline - 2] : 0.0;
  const double t2 = (has2 && a.tlow) ? tlowt[gline - 2] : 0.0;
  cgmem2_t st0 = (cgmem2_t)(a.state + off), st2 = (cgmem2_t)(a.state + off2);
  cgmem2_t pp = (cgmem2_t)(a.p ? a.p + off : a.state + off), gy = (cgmem2_t)(a.gy ? a.gy + off : a.state + off);
  cgmem2_t ts0 = (cgmem2_t)(a.tsrc ? a.tsrc + off : a.state + off), ts2 = (cgmem2_t)(a.tsrc ? a.tsrc + off2 : a.state + off2);
  cgmem2_t tb = (cgmem2_t)(a.tbc ? a.tbc + off : a.state + off);
  tab_t dg = (tab_t)a.diag;
  gmem2_t dst = (gmem2_t)(a.f.out + (long)line * a.f.ldo);
  const int which = a.which;
  const int tm = (a.tbc_cols >= 0) ? (a.tbc_cols + 1) / 2 : (1 << 30);   // modes of the tbc row that are read
  rfft_fwd_core<N>(blk, a.f, [&](int, int, int k, double yr, double yi) {
    double vr = (k < a.cut) ? -a.dt * yr : 0.0, vi = (k < a.cut) ? -a.dt * yi : 0.0;
    if (has0) { const dbl2 s = st0[k]; vr += s.x; vi += s.y; }
    if (c2 != 0.0) { const dbl2 s = st2[k]; vr += c2 * s.x; vi += c2 * s.y; }
    if (which == 0) {
      const dbl2 q = pp[k];
      const double g = a.pk * (double)k;
      vr -= g * q.y; vi += g * q.x;
    } else if (which == 1) {
      const dbl2 g = gy[k], b = (k < tm) ? tb[k] : dbl2{0.0, 0.0};
      vr += -a.dt * g.x + a.ctbc * b.x; vi += -a.dt * g.y + a.ctbc * b.y;
      if (has0) { const dbl2 s = ts0[k]; vr += a.dt * s.x; vi += a.dt * s.y; }
      if (t2 != 0.0) { const dbl2 s = ts2[k]; vr += a.dt * t2 * s.x; vi += a.dt * t2 * s.y; }
    } else {
      const dbl2 b = (k < tm) ? tb[k] : dbl2{0.0, 0.0};
      vr += a.ctbc * b.x; vi += a.ctbc * b.y;
    }
    const double d = dg[k];
    dst[k] = dbl2{vr / d, vi / d};
  });
}

}  // namespace rpde
