// DCT-I of TWO real lines at once (included by line_vm.h).
//
// The lines x^A, x^B (N + 1 reals each, N = 2^L, 8 <= N <= 4096, in two consecutive LDS slots) are treated as
// one complex sequence c = x^A + i x^B.  The DCT-I is real-linear, so Re / Im of the complex
// transform are the transforms of the two lines -- no separation step.  The complex DCT-I is split the
// numerically stable way (FFTW's split-radix form of REDFT00, no O(sqrt n) error growth like the
// real-FFT trick of Numerical Recipes' cosft1):
//   even outputs  E_{2m}   = DCT-I of the folded sequence  s_j = g_j + g_{N-j}   (half the size: recurse)
//   odd outputs   E_{2m+1} = DCT-III of                    d_j = g_j - g_{N-j}   (size M = N/2)
// and a complex DCT-III of size M is ONE complex FFT of size M (Makhoul's permutation, transposed):
//   y_0 = d_0,  y_j = e^{-i pi j / 2M} (d_j + i d_{M-j}),   U = FFT_M(y),   O_{2n} = U_n,  O_{2n+1} = U_{M-1-n}.
// Levels l = 0 .. L-1 have M_l = N / 2^{l+1}; their FFTs (2048, 1024, ..., 1 points for N = 4096: 4095 in
// total) run CONCURRENTLY, eight points per thread and pass: N/8 threads for two lines, i.e. half the
// butterflies, half the LDS traffic and ~2/3 of the barrier phases per line of the packed even-extension
// FFT in dct1_lds.  Everything lives in the two slots the single-line transform uses as its work area.
//
// Layout: Y = complex array over the two slots with the padded index pidx(); level l owns
// Y[N - N_l, N - N_l / 2) (N_l = N / 2^l), the folded sequence g^{l+1} sits behind it in Y[N - N_l / 2, N].
// Semantics (rustdct process_dct1 under funspace's chebyshev forward / backward, src/field.rs:103-110):
//   E_k = x_0 + (-1)^k x_N + 2 sum_{j=1}^{N-1} x_j cos(pi j k / N),  inputs scaled by `pre`, outputs by `post` as in dct1_lds.
#pragma once

namespace rpde {

struct Cx { double re, im; };
RPDE_HD inline Cx cadd(Cx a, Cx b) { return Cx{a.re + b.re, a.im + b.im}; }
RPDE_HD inline Cx csub(Cx a, Cx b) { return Cx{a.re - b.re, a.im - b.im}; }
RPDE_HD inline Cx cmul(Cx a, Cx b) { return Cx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// one orbit of eight entries of a folded sequence g of N_r + 1 entries under the three reflections
// j -> N_r - j, N_r/2 - j, N_r/4 - j: three fold levels in registers
struct Orbit {
  Cx d0[4];   // level a differences at j, H-j, Q-j, Q+j     (H = N_r/2, Q = N_r/4)
  Cx d1[2];   // level b differences at j, Q-j
  Cx d2;      // level c difference at j
  Cx g3;      // folded three times, at j
};
RPDE_HD inline Orbit fold_orbit(const Cx (&g)[8]) {   // g: entries j, N-j, H-j, H+j, Q-j, 3Q+j, Q+j, 3Q-j
  Orbit o;
  const Cx s0 = cadd(g[0], g[1]), s1 = cadd(g[2], g[3]), s2 = cadd(g[4], g[5]), s3 = cadd(g[6], g[7]);
  o.d0[0] = csub(g[0], g[1]); o.d0[1] = csub(g[2], g[3]); o.d0[2] = csub(g[4], g[5]); o.d0[3] = csub(g[6], g[7]);
  const Cx t0 = cadd(s0, s1), t1 = cadd(s2, s3);
  o.d1[0] = csub(s0, s1); o.d1[1] = csub(s2, s3);
  o.g3 = cadd(t0, t1);
  o.d2 = csub(t0, t1);
  return o;
}

// y_k (and y_{M-k}) of one DCT-III level from the pair (d_k, d_{M-k}); conj(tw2[m]) = e^{-i pi m / N}
template <class Y>
RPDE_DEV void emit_pair(Y& y, tab_t tw2, int off, int M, int lev, int k, Cx dk, Cx dmk) {
  if (k >= M) return;                                   // d_M does not exist (degenerate orbits)
  if (k == 0) { y.put(off, dk); return; }
  const int mk = M - k;
  {
    const double c = tw2[2 * (k << lev)], s = tw2[2 * (k << lev) + 1];
    const Cx z{dk.re - dmk.im, dk.im + dmk.re};          // d_k + i d_{M-k}
    y.put(off + k, Cx{c * z.re + s * z.im, c * z.im - s * z.re});
  }
  if (mk != k) {
    const double c = tw2[2 * (mk << lev)], s = tw2[2 * (mk << lev) + 1];
    const Cx z{dmk.re - dk.im, dmk.im + dk.re};
    y.put(off + mk, Cx{c * z.re + s * z.im, c * z.im - s * z.re});
  }
}

struct YArr {   // the complex work array over the two slots
  lds2_t p;
  RPDE_DEV Cx get(int i) const { const dbl2 v = p[pidx(i)]; return Cx{v.x, v.y}; }
  RPDE_DEV void put(int i, Cx v) { p[pidx(i)] = dbl2{v.re, v.im}; }
};

// what one orbit j < E8 = N_r / 8 feeds: levels a and b complete (their pairs (k, M - k) lie inside the
// orbit), the sequence folded three times, and the RAW level-c difference d_j -- its partner d_{E8-j}
// belongs to another thread, so level c is twiddled in place one phase later (fix_level_c)
template <class Y>
RPDE_DEV void emit_orbit(Y& y, tab_t tw2, int N, int Nr, int lev, int j, const Orbit& o) {
  const int H = Nr >> 1, Q = Nr >> 2, E8 = Nr >> 3;
  const int offa = N - Nr, offb = N - H, offc = N - Q, offg = N - E8;
  emit_pair(y, tw2, offa, H, lev, j, o.d0[0], o.d0[1]);
  emit_pair(y, tw2, offa, H, lev, Q - j, o.d0[2], o.d0[3]);
  emit_pair(y, tw2, offb, Q, lev + 1, j, o.d1[0], o.d1[1]);
  y.put(offg + j, o.g3);
  y.put(offc + j, o.d2);
}
// the orbit of j = E8 has four distinct entries (E8, 7 E8, 3 E8, 5 E8): x = their folds
// {d^a_{E8}, d^a_{3 E8}, d^b_{E8}, g3_{E8}}; there is no d^c_{E8}
struct Orbit4 { Cx x[4]; };
RPDE_HD inline Orbit4 fold_orbit4(Cx g0, Cx g1, Cx g2, Cx g3) {   // entries E8, N - E8, H - E8, H + E8
  Orbit4 o;
  const Cx s0 = cadd(g0, g1), s1 = cadd(g2, g3), t = cadd(s0, s1);
  o.x[0] = csub(g0, g1); o.x[1] = csub(g2, g3); o.x[2] = csub(s0, s1); o.x[3] = cadd(t, t);
  return o;
}
template <class Y>
RPDE_DEV void emit_orbit4(Y& y, tab_t tw2, int N, int Nr, int lev, const Orbit4& o) {
  const int H = Nr >> 1, Q = Nr >> 2, E8 = Nr >> 3;
  emit_pair(y, tw2, N - Nr, H, lev, E8, o.x[0], o.x[1]);
  emit_pair(y, tw2, N - H, Q, lev + 1, E8, o.x[2], o.x[2]);
  y.put(N - E8 + E8, o.x[3]);
}
template <class Y>
RPDE_DEV void fix_level_c(Y& y, tab_t tw2, int N, int Nr, int lev, int k) {   // 1 <= k <= E8 / 2
  const int Q = Nr >> 2, E8 = Nr >> 3, offc = N - Q;
  const Cx dk = y.get(offc + k), dmk = y.get(offc + E8 - k);
  emit_pair(y, tw2, offc, E8, lev + 2, k, dk, dmk);
}

RPDE_HD inline void orbit_indices(int Nr, int j, int (&i)[8]) {
  const int H = Nr >> 1, Q = Nr >> 2;
  i[0] = j; i[1] = Nr - j; i[2] = H - j; i[3] = H + j; i[4] = Q - j; i[5] = 3 * Q + j; i[6] = Q + j; i[7] = 3 * Q - j;
}

// level parameters of FFT thread t of N/8: size M (0: this thread owns the tiny levels M = 4, 2, 1),
// offset of the level's region and the thread's butterfly slot b in [0, M/8)
RPDE_HD inline void pair_thread_level(int N, int t, int& M, int& off, int& b) {
  const int tn = (N >> 3) - t;                  // 1 .. N/8
  const int c = (tn <= 1) ? 1 : 1 << (32 - RPDE_CLZ(tn - 1));   // N_l / 8: the power of two >= tn
  M = 4 * c;
  off = N - 8 * c;
  b = c - tn;
  if (c == 1) M = 0;
}

// one Stockham pass of a level on the eight points of a thread: 8 / R butterflies of radix R = 2^LGR
template <int LGR>
RPDE_DEV void pair_pass(YArr& Y, tab_t tw, int N, int off, int b, int st, int lgNs, const double* pr, const double* pi) {
  constexpr int R = 1 << LGR, NQ = 8 >> LGR;
  const int Ns = 1 << lgNs;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = b + q * st, k = j & (Ns - 1);
    double vr[R], vi[R];
#pragma unroll
    for (int t = 0; t < R; ++t) { vr[t] = pr[q + NQ * t]; vi[t] = pi[q + NQ * t]; }
    if (lgNs > 0) {
      // W_{Ns R}^{t k} = W_N^{t k N / (Ns R)}: powers of the table entry for t = 1
      const long e = (long)k * (N >> (lgNs + LGR));
      const double c1 = tw[2 * e], s1 = tw[2 * e + 1];
      double wc = c1, ws = s1;
#pragma unroll
      for (int t = 1; t < R; ++t) {
        const double ar = vr[t], ai = vi[t];
        vr[t] = ar * wc - ai * ws; vi[t] = ar * ws + ai * wc;
        const double nc = wc * c1 - ws * s1; ws = wc * s1 + ws * c1; wc = nc;
      }
    }
    SmallDft<R>::run(vr, vi);
    const int j0 = ((j >> lgNs) << (lgNs + LGR)) + k;
#pragma unroll
    for (int t = 0; t < R; ++t) Y.put(off + j0 + t * Ns, Cx{vr[t], vi[t]});
  }
}

template <class Cfg>
RPDE_DEVN void dct1_pair_lds(Blk& blk, lds_t xa, int SL, int N, bool pre, bool post, int cut, double inv_n,
                             tab_t tw, tab_t tw2) {
  constexpr int T = Cfg::T;
  YArr Y{(lds2_t)xa};
  clds_t A = xa, B = xa + SL;
  const int NT = N >> 3;                        // FFT threads
  RPDE_TLS(blk, Orbit, orb, 1);
  RPDE_TLS(blk, Orbit4, orx, 1);
  // ---- round 0: three fold levels straight from the two real lines, one orbit per thread (thread 0 also
  // takes the short orbit of j = N/8)
  RPDE_PHASE(blk, tid) {
    if (tid < NT) {
      auto src = [&](int m) {
        double f = 1.0;
        if (pre) f = (m == 0 || m == N) ? 1.0 : ((m & 1) ? -0.5 : 0.5);
        return Cx{A[m] * f, B[m] * f};
      };
      int idx[8];
      orbit_indices(N, tid, idx);
      Cx g[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = src(idx[k]);
      RPDE_T(orb)[0] = fold_orbit(g);
      if (tid == 0) RPDE_T(orx)[0] = fold_orbit4(src(NT), src(N - NT), src((N >> 1) - NT), src((N >> 1) + NT));
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    if (tid < NT) {
      emit_orbit(Y, tw2, N, N, 0, tid, RPDE_T(orb)[0]);
      if (tid == 0) emit_orbit4(Y, tw2, N, N, 0, RPDE_T(orx)[0]);
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    if (tid >= 1 && tid <= (NT >> 1)) fix_level_c(Y, tw2, N, N, 0, tid);
  }
  // ---- further rounds inside wavefront 0 (N_r <= 512: at most 64 orbits + the short one), three levels
  // each while N_r >= 8; the level-c fix of round 0 above touches another region of Y
  int Nr = N >> 3, lev = 3;
  for (; Nr >= 8; Nr >>= 3, lev += 3) {
    const int E8 = Nr >> 3, og = N - Nr;
    RPDE_PHASE(blk, tid) {
      if (tid < E8) {
        int idx[8];
        orbit_indices(Nr, tid, idx);
        Cx g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = Y.get(og + idx[k]);
        RPDE_T(orb)[0] = fold_orbit(g);
        if (tid == 0)
          RPDE_T(orx)[0] = fold_orbit4(Y.get(og + E8), Y.get(og + Nr - E8), Y.get(og + (Nr >> 1) - E8), Y.get(og + (Nr >> 1) + E8));
      }
    }
    RPDE_WSYNC();
    RPDE_PHASE(blk, tid) {
      if (tid < E8) {
        emit_orbit(Y, tw2, N, Nr, lev, tid, RPDE_T(orb)[0]);
        if (tid == 0) emit_orbit4(Y, tw2, N, Nr, lev, RPDE_T(orx)[0]);
      }
    }
    RPDE_WSYNC();
    RPDE_PHASE(blk, tid) {
      if (tid >= 1 && tid <= (E8 >> 1)) fix_level_c(Y, tw2, N, Nr, lev, tid);
    }
    RPDE_WSYNC();
  }
  // ---- serial tail (thread 0): the levels left when N_r is 4 or 2, then the last fold: E_0 and E_N
  RPDE_PHASE(blk, tid) {
    if (tid == 0) {
      Cx a, b, c, t0, t1;
      int lv = lev;
      if (Nr == 4) {        // level M = 2 at N - 4; the folded sequence (a, b, c) stays in registers
        const Cx g0 = Y.get(N - 4), g1 = Y.get(N - 3), g2 = Y.get(N - 2), g3 = Y.get(N - 1), g4 = Y.get(N);
        const Cx d1 = csub(g1, g3);
        Y.put(N - 4, csub(g0, g4));
        emit_pair(Y, tw2, N - 4, 2, lv, 1, d1, d1);
        a = cadd(g0, g4); b = cadd(g1, g3); c = cadd(g2, g2);
        ++lv;
      } else if (Nr == 2) {
        a = Y.get(N - 2); b = Y.get(N - 1); c = Y.get(N);
      }
      if (Nr >= 2) {        // level M = 1 at N - 2
        Y.put(N - 2, csub(a, c));
        t0 = cadd(a, c); t1 = cadd(b, b);
      } else {
        t0 = Y.get(N - 1); t1 = Y.get(N);
      }
      Y.put(N - 1, cadd(t0, t1));     // E_0
      Y.put(N, csub(t0, t1));         // E_N
    }
  }
  RPDE_SYNC(blk);
  // ---- the FFTs of all levels, eight points per thread and pass (Stockham, forward sign)
  RPDE_TLS(blk, double, xr, 8);
  RPDE_TLS(blk, double, xi, 8);
  for (int pass = 0; pass < 4; ++pass) {
    RPDE_PHASE(blk, tid) {
      if (tid < NT) {
        int M, off, b;
        pair_thread_level(N, tid, M, off, b);
        if (M >= 8) {
          const int st = M >> 3;
#pragma unroll
          for (int m = 0; m < 8; ++m) { const Cx v = Y.get(off + b + m * st); RPDE_T(xr)[m] = v.re; RPDE_T(xi)[m] = v.im; }
        } else if (pass == 0) {     // tiny levels: M = 4 at N-8, M = 2 at N-4, M = 1 at N-2
#pragma unroll
          for (int m = 0; m < 7; ++m) { const Cx v = Y.get(N - 8 + m); RPDE_T(xr)[m] = v.re; RPDE_T(xi)[m] = v.im; }
        }
      }
    }
    RPDE_SYNC(blk);
    RPDE_PHASE(blk, tid) {
      if (tid < NT) {
        int M, off, b;
        pair_thread_level(N, tid, M, off, b);
        double* pr = RPDE_T(xr);
        double* pi = RPDE_T(xi);
        if (M >= 8) {
          const int lg = 31 - RPDE_CLZ(M);
          const int rem = lg % 3, npass = lg / 3 + (rem ? 1 : 0);
          if (pass < npass) {
            const bool small = rem && pass == 0;
            const int lgR = small ? rem : 3;
            const int lgNs = small ? 0 : (rem + 3 * (pass - (rem ? 1 : 0)));
            if (lgR == 3) pair_pass<3>(Y, tw, N, off, b, M >> 3, lgNs, pr, pi);
            else if (lgR == 2) pair_pass<2>(Y, tw, N, off, b, M >> 3, lgNs, pr, pi);
            else pair_pass<1>(Y, tw, N, off, b, M >> 3, lgNs, pr, pi);
          }
        } else if (pass == 0) {
          double ar[4] = {pr[0], pr[1], pr[2], pr[3]}, ai[4] = {pi[0], pi[1], pi[2], pi[3]};
          SmallDft<4>::run(ar, ai);
#pragma unroll
          for (int t = 0; t < 4; ++t) Y.put(N - 8 + t, Cx{ar[t], ai[t]});
          Y.put(N - 4, Cx{pr[4] + pr[5], pi[4] + pi[5]});
          Y.put(N - 3, Cx{pr[4] - pr[5], pi[4] - pi[5]});
        }
      }
    }
    RPDE_SYNC(blk);
  }
  // ---- outputs: O_{2n} = U_n, O_{2n+1} = U_{M-1-n} at k = (2m + 1) 2^l; two real lines back into the slots
  RPDE_TLS(blk, double, er, 9);
  RPDE_TLS(blk, double, ei, 9);
  RPDE_PHASE(blk, tid) {
    if (tid < NT) {
      int M, off, b;
      pair_thread_level(N, tid, M, off, b);
      if (M >= 8) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { const Cx v = Y.get(off + 8 * b + m); RPDE_T(er)[m] = v.re; RPDE_T(ei)[m] = v.im; }
      } else {
#pragma unroll
        for (int m = 0; m < 9; ++m) { const Cx v = Y.get(N - 8 + m); RPDE_T(er)[m] = v.re; RPDE_T(ei)[m] = v.im; }
      }
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    if (tid < NT) {
      int M, off, b;
      pair_thread_level(N, tid, M, off, b);
      lds_t oa = xa, ob = xa + SL;
      auto store = [&](int k, double re, double im) {
        if (post) {
          double f = (k & 1) ? -inv_n : inv_n;
          if (k == 0 || k == N) f *= 0.5;
          if (k >= cut) f = 0.0;
          re *= f; im *= f;
        }
        oa[k] = re; ob[k] = im;
      };
      auto out_level = [&](int Ml, int u, double re, double im) {   // U_u of the level with Ml points
        const int m = (2 * u < Ml) ? 2 * u : 2 * (Ml - 1 - u) + 1;
        store((2 * m + 1) * (N / (2 * Ml)), re, im);
      };
      if (M >= 8) {
#pragma unroll
        for (int m = 0; m < 8; ++m) out_level(M, 8 * b + m, RPDE_T(er)[m], RPDE_T(ei)[m]);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) out_level(4, u, RPDE_T(er)[u], RPDE_T(ei)[u]);
#pragma unroll
        for (int u = 0; u < 2; ++u) out_level(2, u, RPDE_T(er)[4 + u], RPDE_T(ei)[4 + u]);
        out_level(1, 0, RPDE_T(er)[6], RPDE_T(ei)[6]);
        store(0, RPDE_T(er)[7], RPDE_T(ei)[7]);
        store(N, RPDE_T(er)[8], RPDE_T(ei)[8]);
      }
    }
  }
  RPDE_SYNC(blk);
  (void)T;
}

}  // namespace rpde
