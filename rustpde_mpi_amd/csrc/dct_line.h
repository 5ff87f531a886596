// Backward Chebyshev transform of contiguous lines as a kernel of its own (experiment of round 2):
//
//   out[line][k] = scale * DCT-I( f_m * (a_m - [sten] a_{m-2}) )_k ,   k = 0 .. N,   f_m = (-1)^m / 2 (ends: 1)
//
// i.e. composite (Dirichlet) or orthonormal Chebyshev coefficients of a line -> its N + 1 physical values: what the
// line program  LOAD ; DCT(stencil, backward scaling, fused store)  computes (funspace `backward`, src/field.rs:108-111).
//
// Why a second form: a DCT line in the line VM is a latency chain of ONE workgroup (512 threads, 128 VGPRs, 76 KB of
// LDS: two workgroups per CU and no room for a third).  Here a line belongs to N / 16 threads (256 for N = 4096:
// one wave per SIMD) that keep their 16 complex points in registers through all radix-16 passes and use LDS only
// as an exchange buffer for ONE real component at a time: N (1 + 1/16) doubles = 34.8 KB, so FOUR workgroups
// share a CU at 128 VGPRs.  More barriers per line (an exchange costs four), twice the lines in flight.
//
// The arithmetic is that of dct1_lds: even extension packed two reals per complex, N-point complex FFT (Stockham
// index pattern, radix 16), split step with the twiddles (cos, sin)(pi k / N).
#pragma once
#include "line_vm.h"

namespace rpde {

struct DctLineArgs {
  const double* in; long ldi; int n_in;   // input lines: n_in <= N + 1 coefficients each (the rest counts as zero)
  double* out; long ldo;                  // output lines: N + 1 values each
  int nlines;
  int N;                                  // 256 or 4096
  int sten;                               // 0: orthonormal input; 2: Dirichlet composite input (c_m = a_m - a_{m-2});
                                          // 1: composite input with the stencil table `low` (c_m = a_m + low[m-2] a_{m-2})
  const double* tw;                       // N complex FFT twiddles (cos, -sin)(2 pi k / N)      (AxisTables::tw)
  const double* tw2;                      // split twiddles (cos, sin)(pi k / N), k = 0 .. N     (AxisTables::tw2)
  double scale;
  const double* low = nullptr;            // sten == 1: AxisTables::low (length >= N - 1)
  int deriv = 0;                          // 1: transform dscale * d/dx of the orthonormal series instead of the series
  double dscale = 1.0;                    //    (funspace `gradient` along the line, src/field.rs:127-129)
  int fwd = 0;                            // 1: forward transform (funspace `forward` of the orthonormal base, src/field.rs:103-106):
  int cut = 1 << 30;                      //    N + 1 physical values in, coefficients (-1)^k E_k / N (ends halved) out, zero from `cut` on
};

RPDE_HD inline size_t dct_line_lds_doubles(int N) { return (size_t)N + N / 16 + 16; }   // the padded x-layout ends at N + 3 + (N + 3) / 16; wave totals of the derivative behind it
template <int N> struct DctGeom { static constexpr int LDS = N + N / 16 + 16; };
// the 16-byte staging loads need an aligned line start and, for an odd count, one readable element behind the line
// (N = 1024 runs on the half-length core, hdct_line.h, only)
RPDE_HD inline bool dct_line_ok(const DctLineArgs& a) {
  return (a.N == 256 || a.N == 1024 || a.N == 2048 || a.N == 4096) && a.n_in >= 1 && a.n_in <= a.N + 1 && (((size_t)a.in) & 15) == 0 && (a.ldi & 1) == 0 &&
         ((a.n_in & 1) == 0 || a.n_in < a.ldi) && (a.sten == 0 || a.sten == 2 || (a.sten == 1 && a.low != nullptr));
}

// what the split phase does with a result: emit(tid, slot, k, E_k) -- slot 2 t / 2 t + 1 for the pair (k, N - k) of a thread's
// t-th point, slot 16 for k = N / 2 (thread 0).  The plain transform stores to the output line.
struct DctStoreEmit {
  gmem_t dst; double sc;
  RPDE_DEV void operator()(int, int, int k, double v) const { dst[k] = sc * v; }
};

// staged: the input line is already in the buffer (x[m] at buf[m + 2], zeros around it) and a barrier has been passed
// PADX: what the caller knows at compile time about the x-layout (1: derivative / table stencil, 0: neither; -1: look at the flags)
template <int N, class Emit, int PADX = -1>
RPDE_DEV void dct_line_core(Blk& blk, const DctLineArgs& a, bool staged, const Emit& emit) {
  constexpr int T = N / 16;
  static_assert(N == 4096 || N == 256, "N = 16^2 or 16^3");
  static_assert(T % 16 == 0, "padded indices assume T a multiple of 16");
  lds_t buf = (lds_t)blk.lds;
  lds2_t buf2 = (lds2_t)blk.lds;
  const int line = blk.line;
  tab_t tw = (tab_t)a.tw;
  tab_t tw2 = (tab_t)a.tw2;
  const int n_in = a.n_in;
  const bool sten = a.sten == 2;
  // x-layout of the staged line (x[m] at buffer index b = m + 2): padded to b + b / 16 when the line is swept in chunks of 16
  // per thread (table stencil / derivative) -- see hdct_core (hdct_line.h): unpadded, those chunk accesses are 16-way bank conflicts
  // (X(r + 16 m) = X(r) + P m, P = 17 or 16: one runtime term per thread and phase, the rest folds into constants)
  const bool padx = PADX >= 0 ? PADX != 0 : (a.sten == 1 || a.deriv != 0);
  const int padm = padx ? -1 : 0, P = padx ? 17 : 16;
  auto X = [padm](int b) { return b + ((b >> 4) & padm); };
  RPDE_TLS(blk, double, re, 16);
  RPDE_TLS(blk, double, im, 16);

  // ---- stage the line two doubles into the buffer: xs[m + 2] = x[m] (m < n_in), zeros in front and behind, so that
  // the stencil tap x[m - 2] and the tail m >= n_in need no selects.  Pair p holds xs[2p], xs[2p + 1] = x[2p - 2], x[2p - 1].
  if (!staged) {
  // through a buffer descriptor that ends behind the pair holding the last coefficient: a pair outside the line reads zero,
  // no bounds test, no branch, one 32-bit offset per thread (round 5, as hdct_core)
  const RowBuf rb = row_buf(a.in + (long)line * a.ldi, 8L * ((n_in + 1) & ~1));
  // a Dirichlet line whose DERIVATIVE is wanted gets its stencil c_m = a_m - a_{m-2} here, on the way in, like hdct_core's
  // lines (the second pair of a thread is its neighbour's first: an L1 hit) -- rounds 2 - 4 sent it through the table-stencil
  // block below with a table of -1: one more pass of the line through LDS and two more barriers per convection term
  const bool dsten = a.sten == 2 && a.deriv != 0;
  RPDE_PHASE(blk, tid) {
    constexpr int QP = (N + 4 + 2 * T - 1) / (2 * T);   // pairs per thread: 2 T QP >= N + 4
    dbl2 v[QP], o[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int p = tid + q * T;                          // pair p = elements 2 p - 2, 2 p - 1 at byte 16 (p - 1)
      v[q] = row_ld2(rb, 16 * (p - 1), 0);
      o[q] = dsten ? row_ld2(rb, 16 * (p - 2), 0) : dbl2{0.0, 0.0};
    }
#pragma unroll
    for (int q = 0; q < QP; ++q) {
      const int p = tid + q * T, k = 2 * p - 2;
      dbl2 w = v[q], u = o[q];
      if (k + 1 >= n_in) w.y = 0.0;
      if (k - 1 >= n_in) u.y = 0.0;
      w.x -= u.x; w.y -= u.y;
      if (2 * p + 1 < N + 4) {
        if (padx) { const int xq = X(2 * tid) + P * (q * T / 8); buf[xq] = w.x; buf[xq + 1] = w.y; }   // b = 2 p is even: the pair stays inside its group of 16
        else buf2[p] = w;
      }
    }
  }
  RPDE_SYNC(blk);
  }

  // ---- table stencil and / or derivative: the orthonormal coefficients (then their derivative) replace the staged
  // line, thread t owning the contiguous chunk k = 16 t .. 16 t + 15 (the last thread also k = N)
  bool sten_in_read = sten;
  if (a.sten == 1 || a.deriv) {
    sten_in_read = false;
    if (a.sten == 1 || (a.sten == 2 && staged)) {        // (a staged Dirichlet line has not had its stencil yet)
      tab_t low = (tab_t)a.low;
      RPDE_TLS(blk, double, c, 17);
      RPDE_PHASE(blk, tid) {
        const int k0 = 16 * tid;
        double xs[19], lw[17];
#pragma unroll
        for (int i = 0; i < 19; ++i) xs[i] = buf[P * tid + X(i)];               // xs[i] = a_{k0 + i - 2} at b = k0 + i
#pragma unroll
        for (int i = 0; i < 17; ++i) lw[i] = (a.sten == 2) ? -1.0 : low[max(k0 + i - 2, 0)];
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
    if (a.deriv) {
      // d_k = dscale * sum_{j > k, j + k odd} 2 j c_j, d_0 halved (the suffix sums of scan_cheb_diff, line_vm.h): thread t
      // owns the chunk lo = 16 (T - 1 - t), so that the carry flows from thread t - 1 to thread t
      RPDE_TLS(blk, double, zz, 16);
      RPDE_TLS(blk, double, vv, 2);
      lds_t carry = buf + N + N / 16 + 8;      // behind the padded line
      RPDE_PHASE(blk, tid) {
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
        constexpr int NW = (T + 63) / 64;
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
        for (int par = 0; par < 2; ++par) vv[par] = dpp_f64<0x138, 0xF>(0.0, v[par]) + S[par];   // wave_shr:1
      }
#endif
      RPDE_SYNC(blk);
      RPDE_PHASE(blk, tid) {
        const int lo = (T - 1 - tid) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int k = lo + i;
          buf[P * (T - 1 - tid) + X(i + 2)] = (RPDE_T(zz)[i] + RPDE_T(vv)[i & 1]) * ((k == 0) ? 0.5 * a.dscale : a.dscale);
        }
        if (tid == 0) buf[X(N + 2)] = 0.0;       // d_N = 0
      }
      RPDE_SYNC(blk);
    }
  }

  // ---- inputs of the first pass: z_i = (v_{2i}, v_{2i+1}) for i < N/2, (v_{2N-2i}, v_{2N-2i-1}) behind, i = tid + t T
  RPDE_PHASE(blk, tid) {
    const bool sten = sten_in_read;
    clds2_t xs2 = (clds2_t)buf;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int i = tid + t * T;               // m0 = 2 i: x[m0], x[m0+1] = pair i + 1; x[m0-2], x[m0-1] = pair i
      dbl2 c, p;
      if (padx) { const int xq = X(2 * tid + 2) + P * (t * T / 8); c = dbl2{buf[xq], buf[xq + 1]}; p = dbl2{0.0, 0.0}; }   // (padx: the stencil has been applied)
      else { c = xs2[i + 1]; p = xs2[i]; }
      const double v0 = sten ? c.x - p.x : c.x, v1 = sten ? c.y - p.y : c.y;
      const double f0 = (t == 0 && tid == 0) ? 1.0 : 0.5;      // m = 0: the end of the line
      RPDE_T(re)[t] = a.fwd ? v0 : f0 * v0;
      RPDE_T(im)[t] = a.fwd ? v1 : -0.5 * v1;
    }
#pragma unroll
    for (int t = 8; t < 16; ++t) {
      const int m0 = 2 * N - 2 * (tid + t * T);                // even, 2 <= m0 <= N
      // b = m0 + 2 = (2 N + 2 - 2 tid) - 2 t T: 2 t T is a multiple of 16; the four taps are four runtime terms per thread
      const int mt = P * (t * T / 8);
      const double x0 = buf[X(2 * N + 2 - 2 * tid) - mt], x1 = buf[X(2 * N + 1 - 2 * tid) - mt], t0 = buf[X(2 * N - 2 * tid) - mt],
                   t1 = buf[X(2 * N - 1 - 2 * tid) - mt];      // x[m0], x[m0-1], x[m0-2], x[m0-3]
      const double v0 = sten ? x0 - t0 : x0, v1 = sten ? x1 - t1 : x1;
      const double f0 = (t == 8 && tid == 0) ? 1.0 : 0.5;      // m = N: the other end
      RPDE_T(re)[t] = a.fwd ? v0 : f0 * v0;
      RPDE_T(im)[t] = a.fwd ? v1 : -0.5 * v1;
    }
    SmallDft<16>::run(RPDE_T(re), RPDE_T(im));                 // first pass: no twiddles
  }

  // exchange after the pass with Ns = 2^LGNS: output t of butterfly j belongs to position j0 + t Ns, input t of
  // butterfly j of the next pass is position j + t T; one real component at a time through the padded buffer
  auto exchange = [&](auto LG) {
    constexpr int LGNS = decltype(LG)::value, Ns = 1 << LGNS;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      RPDE_SYNC(blk);                                          // everybody has read what this overwrites
      RPDE_PHASE(blk, tid) {
        const int j0 = ((tid >> LGNS) << (LGNS + 4)) + (tid & (Ns - 1));
        const int b0 = pidx(j0);
        double* z = half ? RPDE_T(im) : RPDE_T(re);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int p = (Ns >= 16) ? b0 + t * Ns + (t * Ns) / 16 : pidx(j0 + t * Ns);
          buf[p] = z[t];
        }
      }
      RPDE_SYNC(blk);
      RPDE_PHASE(blk, tid) {
        const int b = pidx(tid);
        double* z = half ? RPDE_T(im) : RPDE_T(re);
#pragma unroll
        for (int t = 0; t < 16; ++t) z[t] = buf[b + t * T + (t * T) / 16];
      }
    }
  };
  // pass with twiddles W_N^(t k tstep), k = j mod Ns, tstep = N / (16 Ns): powers of the table entry for t = 1
  auto pass = [&](auto LG) {
    constexpr int LGNS = decltype(LG)::value, Ns = 1 << LGNS, tstep = N / (16 * Ns);
    RPDE_PHASE(blk, tid) {
      const int k = tid & (Ns - 1);
      const double wc = tw[2 * (k * tstep)], ws = tw[2 * (k * tstep) + 1];
      double cc = wc, cs = ws;
      double* xr = RPDE_T(re);
      double* xi = RPDE_T(im);
#pragma unroll
      for (int t = 1; t < 16; ++t) {
        const double ar = xr[t], ai = xi[t];
        xr[t] = ar * cc - ai * cs;
        xi[t] = ar * cs + ai * cc;
        if (t < 15) { const double nc = cc * wc - cs * ws, ns = cc * ws + cs * wc; cc = nc; cs = ns; }
      }
      SmallDft<16>::run(xr, xi);
    }
  };
  exchange(std::integral_constant<int, 0>{});
  pass(std::integral_constant<int, 4>{});
  if constexpr (N == 4096) {
    exchange(std::integral_constant<int, 4>{});
    pass(std::integral_constant<int, 8>{});
  }
  // now thread j holds Z_k for k = j + t T (natural order)

  // ---- split: E_k = A + B, E_{N-k} = A - B, A = (Zr_k + Zr_{N-k}) / 2, B = (c_k (Zi_k + Zi_{N-k}) - s_k (Zr_k - Zr_{N-k})) / 2.
  // Thread j produces the pairs of k = j + t T, t < 8 (k < N/2); the partner Z_{N-k} comes through the buffer, again
  // one real component at a time.  Thread 0 also owns k = 0 <-> N (its own partner) and k = N/2.
  RPDE_TLS(blk, double, pr, 8);
  RPDE_TLS(blk, double, cs8, 16);
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int b = pidx(tid);
#pragma unroll
    for (int t = 0; t < 16; ++t) buf[b + t * T + (t * T) / 16] = RPDE_T(re)[t];
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int nb = -tid + ((-tid) >> 4);                       // pidx(-tid): N - k = (N - t T) - tid, N - t T a multiple of 16
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int p = (t == 0 && tid == 0) ? 0 : nb + (N - t * T) + (N - t * T) / 16;
      RPDE_T(pr)[t] = buf[p];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {                              // split twiddles of this thread's k: in flight across the next barriers
      const int k = tid + t * T;
      RPDE_T(cs8)[2 * t] = tw2[2 * k];
      RPDE_T(cs8)[2 * t + 1] = tw2[2 * k + 1];
    }
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int b = pidx(tid);
#pragma unroll
    for (int t = 0; t < 16; ++t) buf[b + t * T + (t * T) / 16] = RPDE_T(im)[t];
  }
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {
    const int nb = -tid + ((-tid) >> 4);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = tid + t * T;
      const int p = (t == 0 && tid == 0) ? 0 : nb + (N - t * T) + (N - t * T) / 16;
      const double pi = buf[p];
      const double zr = RPDE_T(re)[t], zi = RPDE_T(im)[t], c = RPDE_T(cs8)[2 * t], s = RPDE_T(cs8)[2 * t + 1];
      const double A = 0.5 * (zr + RPDE_T(pr)[t]), B = 0.5 * (c * (zi + pi) - s * (zr - RPDE_T(pr)[t]));
      double e0 = A + B, e1 = A - B;
      if (a.fwd) {   // (-1)^k / N, both ends halved, the 2/3 rule; k and N - k have the parity of tid (T and N are even)
        double f = (tid & 1) ? -1.0 / (double)N : 1.0 / (double)N;
        if (t == 0 && tid == 0) f *= 0.5;
        e0 = (k < a.cut) ? e0 * f : 0.0;
        e1 = (N - k < a.cut) ? e1 * f : 0.0;
      }
      emit(tid, 2 * t, k, e0);
      emit(tid, 2 * t + 1, N - k, e1);
    }
    if (tid == 0) {   // k = N/2 = 8 T: its own partner
      const double c = tw2[2 * (N / 2)], s = tw2[2 * (N / 2) + 1];
      const double zr = RPDE_T(re)[8], zi = RPDE_T(im)[8];
      const double A = 0.5 * (zr + zr), B = 0.5 * (c * (zi + zi) - s * (zr - zr));
      double e0 = A + B;
      if (a.fwd) e0 = (N / 2 < a.cut) ? e0 * (1.0 / (double)N) : 0.0;       // N / 2 is even for N >= 4
      emit(tid, 16, N / 2, e0);
    }
  }
}

template <int N>
RPDE_DEV void dct_bwd_line(Blk& blk, const DctLineArgs& a) {
  dct_line_core<N>(blk, a, false, DctStoreEmit{(gmem_t)(a.out + (long)blk.line * a.ldo), a.scale});
}

// One y-line of a convection term (src/navier_stokes/functions.rs:56-72, navier_eq.rs conv_velx / conv_vely / conv_temp):
//   out = forward_y[ u (A + bx) + v (B + by) ] with the 2/3 rule,  A = backward_y(fx),  B = backward_y(dscale d/dy f0),
// fx = d/dx f and f0 = f in (physical x, Dirichlet-composite y), u / v / bx / by physical along the line.  Three transforms
// per line in registers; the first product waits in 17 registers per thread while the second transform runs (136 VGPRs:
// three workgroups per CU); the sum is staged in the exchange buffer as the input of the forward transform.
struct ConvLineArgs {
  const double* fx; const double* f0; const double* up; const double* vp; const double* bx; const double* by;   // bx / by may be null
  long ld; int n_in;              // all inputs share the pitch; n_in composite coefficients, N + 1 physical values
  double* out; long ldo;
  int nlines, N;
  const double* tw; const double* tw2;
  double dscale; int cut;
  long ldl = -1;                  // pitch of bx / by (-1: ld).  0: every line reads line 0 -- a lift that does not depend on x ("rbc":
                                  // linear in y) has the same gradient on every y-line, and two of a convection term's six input arrays
                                  // then come out of the L2 instead of HBM (Navier2DEngine::analyse_lift)
  // The linearised convection term of Navier2DLnse (lnse_eq.rs:59-110): um, vm = the physical mean velocities (pitch ld), bx, by = the
  // physical gradient of the mean field the term belongs to; out = DCT_y[ um d/dx f + vm d/dy f + up bx + vp by ] (conv_line<N, true>).
  const double* um = nullptr; const double* vm = nullptr;
  // nonlin != 0 (Navier2DNonLin, nonlin_eq.rs:59-134): (um + up) (d/dx f + bx) + (vm + vp) (d/dy f + by) -- the classic term with the
  // mean velocities added to the perturbation's (conv_line<N, 2>)
  int nonlin = 0;
  // The adjoint LNSE term (lnse_adj_eq.rs:16-94; conv_line<N, 3>): um, vm hold MINUS the mean velocities, up / vp / tp the physical adjoint
  // fields u*, v*, T*, bx / by / cz the mean gradients d_j U, d_j V, d_j T of the equation's direction j (zero arrays for the temperature
  // equation): out = DCT_y[ um d/dx f + vm d/dy f + up bx + vp by + tp cz ] = MINUS the reference's conv_*_adjoint (the step subtracts dt out)
  const double* tp = nullptr; const double* cz = nullptr;
};
RPDE_HD inline long conv_lift_pitch(const ConvLineArgs& c) { return c.ldl >= 0 ? c.ldl : c.ld; }
RPDE_HD inline bool conv_line_ok(const ConvLineArgs& c) {
  const DctLineArgs a{c.fx, c.ld, c.n_in, nullptr, 0, c.nlines, c.N, 2, c.tw, c.tw2, 1.0};
  DctLineArgs b = a;
  b.in = c.f0;
  return dct_line_ok(a) && dct_line_ok(b);   // N = 16^k: three transforms on the full-length core; N = 1024: hconv_line (hdct_line.h), one wave per line
}

template <int N, int MEAN = 0>   // 0: Navier2D, 1: Navier2DLnse (linearised), 2: Navier2DNonLin
RPDE_DEV void conv_line(Blk& blk, const ConvLineArgs& c) {
  constexpr int T = N / 16;
  lds_t buf = (lds_t)blk.lds;
  const long off = (long)blk.line * c.ld;
  cgmem_t up = (cgmem_t)(c.up + off), vp = (cgmem_t)(c.vp + off);
  cgmem_t um = (cgmem_t)((MEAN ? c.um : c.up) + off), vm = (cgmem_t)((MEAN ? c.vm : c.vp) + off);
  const long offl = (long)blk.line * conv_lift_pitch(c);
  cgmem_t bx = (cgmem_t)(c.bx ? c.bx + offl : nullptr), by = (cgmem_t)(c.by ? c.by + offl : nullptr);
  const bool lift = c.bx != nullptr;
  // the physical factors of a thread's 17 points through buffer descriptors (line_vm.h RowBuf): slot 2 t is k = tid + t T, slot
  // 2 t + 1 is N - k = (T - tid) + (15 - t) T -- one per-thread offset each (8 tid, 8 (T - tid)) and the block t in a scalar
  // register, where flat loads spent a 64-bit vector add per point and array (an eighth of the kernel's vector instructions)
  const long rowb = 8L * (N + 1);
  const RowBuf rup = row_buf(c.up + off, rowb), rvp = row_buf(c.vp + off, rowb);
  const RowBuf rbx = row_buf(lift ? c.bx + offl : c.up + off, rowb), rby = row_buf(lift ? c.by + offl : c.vp + off, rowb);
  const RowBuf rum = row_buf((MEAN ? c.um : c.up) + off, rowb), rvm = row_buf((MEAN ? c.vm : c.vp) + off, rowb);
  cgmem_t tp = (cgmem_t)((MEAN == 3 ? c.tp : c.up) + off), cz = (cgmem_t)((MEAN == 3 ? c.cz : c.up) + off);
  const RowBuf rtp = row_buf((MEAN == 3 ? c.tp : c.up) + off, rowb), rcz = row_buf((MEAN == 3 ? c.cz : c.up) + off, rowb);
  auto pick = [&](const RowBuf& r, cgmem_t flat, int tid, int slot, int k) {
    if (slot == 16) return flat[k];                          // k = N / 2, thread 0 only
    return (slot & 1) ? row_ld1(r, 8 * (T - tid), 8 * (15 - (slot >> 1)) * T) : row_ld1(r, 8 * tid, 8 * (slot >> 1) * T);
  };
  RPDE_TLS(blk, double, acc, 17);
  // The factor loads of an emit sit behind a RUN-TIME condition that always holds (`on`; MEAN: `lift`): with unconditional loads the
  // compiler lifts the factor streams of all 17 slots above the transform's last passes -- 167 registers for the classic term (u
  // unconditional, the lift behind `lift`), 174 spilled with `lift` known at compile time, 114 - 258 spilled for the linearised term;
  // behind the condition the loads stay where the emit is: 136 registers, none spilled, for both (round 6).
  // Measured (profiles/r06_experiments/call18_*): on THIS core at 4097-point lines the guarded form is slower at three workgroups per CU
  // (0.72 against 0.65 ms: the early loads were its latency hiding) and equal at four (128 registers, 8 spilled) -- the classic term keeps
  // its unconditional loads here (RPDE_CONV4096_GUARD: the A/B build); on the half-length core (hconv_line: 400 -> 154 registers, three
  // waves per SIMD instead of one) the guard is the default and took 28 % off the term.
#ifdef RPDE_CONV4096_GUARD
  const bool on = c.up != nullptr;
#else
  const bool on = true;
#endif
  DctLineArgs a1{c.fx, c.ld, c.n_in, nullptr, 0, c.nlines, N, 2, c.tw, c.tw2, 1.0};
  auto e1 = [&](int tid, int slot, int k, double v) {
    if constexpr (MEAN == 2) RPDE_T(acc)[slot] = lift ? (pick(rum, um, tid, slot, k) + pick(rup, up, tid, slot, k)) * (v + pick(rbx, bx, tid, slot, k)) : v;
    else if constexpr (MEAN == 1 || MEAN == 3) RPDE_T(acc)[slot] = lift ? pick(rum, um, tid, slot, k) * v + pick(rup, up, tid, slot, k) * pick(rbx, bx, tid, slot, k) : v;
    else RPDE_T(acc)[slot] = on ? pick(rup, up, tid, slot, k) * (lift ? v + pick(rbx, bx, tid, slot, k) : v) : v;
  };
  dct_line_core<N, decltype(e1), 0>(blk, a1, false, e1);
  RPDE_SYNC(blk);
  DctLineArgs a2 = a1;
  a2.in = c.f0; a2.deriv = 1; a2.dscale = c.dscale;
  auto e2 = [&](int tid, int slot, int k, double v) {
    if constexpr (MEAN == 2) RPDE_T(acc)[slot] += lift ? (pick(rvm, vm, tid, slot, k) + pick(rvp, vp, tid, slot, k)) * (v + pick(rby, by, tid, slot, k)) : v;
    else if constexpr (MEAN == 3) RPDE_T(acc)[slot] += lift ? pick(rvm, vm, tid, slot, k) * v + pick(rvp, vp, tid, slot, k) * pick(rby, by, tid, slot, k) + pick(rtp, tp, tid, slot, k) * pick(rcz, cz, tid, slot, k) : v;
    else if constexpr (MEAN == 1) RPDE_T(acc)[slot] += lift ? pick(rvm, vm, tid, slot, k) * v + pick(rvp, vp, tid, slot, k) * pick(rby, by, tid, slot, k) : v;
    else RPDE_T(acc)[slot] += on ? pick(rvp, vp, tid, slot, k) * (lift ? v + pick(rby, by, tid, slot, k) : v) : v;
  };
  dct_line_core<N, decltype(e2), 1>(blk, a2, false, e2);
  RPDE_SYNC(blk);
  RPDE_PHASE(blk, tid) {   // the sum as the staged input line of the forward transform
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = tid + t * T;
      buf[k + 2] = RPDE_T(acc)[2 * t];
      buf[N - k + 2] = RPDE_T(acc)[2 * t + 1];
    }
    if (tid == 0) { buf[N / 2 + 2] = RPDE_T(acc)[16]; buf[0] = 0.0; buf[1] = 0.0; buf[N + 3] = 0.0; }
  }
  RPDE_SYNC(blk);
  DctLineArgs a3{nullptr, 0, N + 1, nullptr, 0, c.nlines, N, 0, c.tw, c.tw2, 1.0};
  a3.fwd = 1; a3.cut = c.cut;
  dct_line_core<N, DctStoreEmit, 0>(blk, a3, true, DctStoreEmit{(gmem_t)(c.out + (long)blk.line * c.ldo), 1.0});
}

}  // namespace rpde
