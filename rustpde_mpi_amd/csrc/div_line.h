// S5 of the confined step as one whole-line kernel: per x-line (one y row j of the YX arrays)
//
//   div = d/dx to_ortho_x( S_y velx )  +  to_ortho_x( d/dy vely )            navier_eq.rs:59-73 (`div`), src/field.rs:113-129
//   g   = B2_x div                                                             Poisson::solve_par, x preconditioner (poisson.rs:206-212,
//                                                                              MatVecFdma rows, src/solver/matvec.rs:207-228)
//
// S_y = the y stencil of the velocity base applied across the lines (rows j and j - 2 of velx); d/dy vely comes from the
// column scan in front of this stage (orthonormal rows).  `div` is stored (N + 1 orthonormal coefficients per line: the
// pressure update reads it), `g` goes out parity de-interleaved for the eigen-transform GEMM (even coefficients first, the
// odd ones `half` columns further).  The line program of the stage (engine.cc S5) needs 512 threads and two LDS slots;
// here 256 threads and one padded line buffer (four workgroups per CU): the element-wise parts in pairs (thread t owns
// k = 2 (t + u T), coalesced 16-byte accesses), the derivative as a chunked suffix sum (thread t owns the 16 coefficients
// of chunk T - 1 - t, so the carry flows from thread t - 1 to thread t; DPP scan in a wave, wave totals through LDS).
#pragma once
#include "rhs_line.h"

namespace rpde {

struct DivLineArgs {
  const double* u = nullptr;      // velx state, composite coefficients (N - 1 per line), rows j and j - 2 are read
  const double* dyv = nullptr;    // d/dy vely, composite-x coefficients (N - 1 per line) of the orthonormal-y rows
  double* div = nullptr;          // N + 1 orthonormal coefficients per line
  double* g = nullptr;            // N - 1 coefficients per line, parity de-interleaved
  long ld = 0;                    // all arrays share the pitch
  int nlines = 0, line0 = 0;      // local lines, global index of the first one
  int N = 0, my = 0, half = 0;    // rows >= my of the velocity do not exist (their stencil tap of row j - 2 does)
  double dscale = 1.0;            // 1 / scale_x
  const double* lowy = nullptr;   // y stencil S[j, j - 2] (indexed with j - 2)
  const double *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;   // B2 rows (natural order, zero behind N - 1 entries)
};
RPDE_HD inline bool div_line_ok(const DivLineArgs& a) {
  return (a.N == 256 || a.N == 1024 || a.N == 4096) && a.u && a.dyv && a.div && a.g && a.lowy && a.p0 && a.p1 && a.p2 &&
         (((size_t)a.u | (size_t)a.dyv | (size_t)a.div | (size_t)a.p0 | (size_t)a.p1 | (size_t)a.p2) & 15) == 0 && (a.ld & 1) == 0 &&
         a.ld > a.N + 1 && a.half > 0;
}

template <int N>
RPDE_DEV void div_line(Blk& blk, const DivLineArgs& a) {
  using G = HdctGeom<N>;
  constexpr int T = G::T, NW = G::NW;
  lds_t buf = (lds_t)blk.lds;
  lds_t scr = buf + G::SCR;
  const int line = blk.line, gline = line + a.line0;
  const long off = (long)line * a.ld;
  const bool has0 = gline < a.my, has2 = gline >= 2;
  const double cy = has2 ? ((tab_t)a.lowy)[gline - 2] : 0.0;
  const int n = N - 1;
  auto pidx = [](int k) { return k + (k >> 4) + 2; };       // padded position of coefficient k (stride 17 per chunk of 16)

  // ---- x stencil (Dirichlet: c_k = a_k - a_{k-2}) of S_y velx and of d/dy vely.  The pair in front of a thread's pair is
  // its neighbour's: it is loaded a second time (an L1 hit), like rhs_line.h does.  c goes into the buffer, the stencilled
  // d/dy vely waits in registers (same ownership in the phase that needs it).
  RPDE_TLS(blk, double, e, 17);
  RPDE_PHASE(blk, tid) {
    cgmem2_t u0 = (cgmem2_t)(a.u + off), u2 = (cgmem2_t)(a.u + (has2 ? off - 2 * a.ld : off)), dv = (cgmem2_t)(a.dyv + off);
    auto pair = [&](cgmem2_t p, int m, bool on) {            // elements m, m + 1 of a row of n coefficients, zeros outside
      dbl2 v = (on && m >= 0 && m < n) ? p[m >> 1] : dbl2{0.0, 0.0};
      if (m + 1 >= n) v.y = 0.0;
      return v;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // four pairs at a time: 24 loads in flight
      dbl2 x0[4], x2[4], xm0[4], xm2[4], b0[4], bm[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 2 * (tid + (4 * h + i) * T);
        x0[i] = pair(u0, m, has0); xm0[i] = pair(u0, m - 2, has0);
        x2[i] = pair(u2, m, has2); xm2[i] = pair(u2, m - 2, has2);
        b0[i] = pair(dv, m, true); bm[i] = pair(dv, m - 2, true);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        RPDE_PIN(x0[i].x); RPDE_PIN(x0[i].y); RPDE_PIN(xm0[i].x); RPDE_PIN(xm0[i].y); RPDE_PIN(x2[i].x); RPDE_PIN(x2[i].y);
        RPDE_PIN(xm2[i].x); RPDE_PIN(xm2[i].y); RPDE_PIN(b0[i].x); RPDE_PIN(b0[i].y); RPDE_PIN(bm[i].x); RPDE_PIN(bm[i].y);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = 4 * h + i, m = 2 * (tid + u * T);
        const double cx = (x0[i].x + cy * x2[i].x) - (xm0[i].x + cy * xm2[i].x);
        const double cyv = (x0[i].y + cy * x2[i].y) - (xm0[i].y + cy * xm2[i].y);
        buf[pidx(m)] = cx;
        buf[pidx(m + 1)] = cyv;                               // m + 1 stays inside the group of 16
        RPDE_T(e)[2 * u] = b0[i].x - bm[i].x;
        RPDE_T(e)[2 * u + 1] = b0[i].y - bm[i].y;
      }
#ifndef RPDE_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (tid == 0) {                                         // k = N: no own coefficient, the tap of k - 2 only
      const dbl2 s0 = pair(u0, N - 2, has0), s2 = pair(u2, N - 2, has2), bb = pair(dv, N - 2, true);
      buf[pidx(N)] = -(s0.x + cy * s2.x);
      RPDE_T(e)[16] = -bb.x;
      buf[pidx(N + 1)] = 0.0;                               // the derivative reads c_{k+1} up to k = N
    }
  }
  RPDE_SYNC(blk);

  // ---- d_k = dscale sum_{j > k, j + k odd} 2 j c_j (d_0 halved, d_N = 0): suffix sums per parity
  RPDE_TLS(blk, double, zz, 16);
  RPDE_TLS(blk, double, vd, 2);
  RPDE_PHASE(blk, tid) {
    const int lo = (T - 1 - tid) * 16;
    double bb[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bb[i] = 2.0 * (double)(lo + i + 1) * buf[pidx(lo + i + 1)];   // 2 (k + 1) c_{k+1}, k + 1 <= N
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double z = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ei = 14 + par - 2 * i;
        z += bb[ei];
        RPDE_T(zz)[ei] = z;
      }
      RPDE_T(vd)[par] = z;
    }
  }
#ifdef RPDE_EMU
  (void)scr;
  for (int par = 0; par < 2; ++par) {
    double run = 0.0;
    for (int t = 0; t < T; ++t) { const double mine = vd_st[(size_t)t * 2 + par]; vd_st[(size_t)t * 2 + par] = run; run += mine; }
  }
#else
  {
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double v[2] = {vd[0], vd[1]};
    v[0] = sum_wave_scan(v[0]);
    v[1] = sum_wave_scan(v[1]);
    double S[2] = {0.0, 0.0};
    if constexpr (NW > 1) {
      if (lane == 63) { scr[8 + wave] = v[0]; scr[8 + NW + wave] = v[1]; }
      __syncthreads();
      for (int x = 0; x < wave; ++x) { S[0] += scr[8 + x]; S[1] += scr[8 + NW + x]; }
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) vd[par] = dpp_f64<0x138, 0xF>(0.0, v[par]) + S[par];   // wave_shr:1
  }
#endif
  RPDE_SYNC(blk);                                           // everybody has read c
  RPDE_PHASE(blk, tid) {
    const int lo = (T - 1 - tid) * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = lo + i;
      buf[pidx(k)] = (RPDE_T(zz)[i] + RPDE_T(vd)[i & 1]) * ((k == 0) ? 0.5 * a.dscale : a.dscale);
    }
    if (tid == 0) {
#pragma unroll
      for (int k = N; k <= N + 5; ++k) buf[pidx(k)] = 0.0;    // d_N = 0; zeros behind it for the taps of the B2 rows
    }
  }
  RPDE_SYNC(blk);

  // ---- div = d + stencilled d/dy vely: out in pairs, and back into the buffer for the B2 rows
  RPDE_PHASE(blk, tid) {
    gmem2_t dst = (gmem2_t)(a.div + off);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T), p = pidx(m);
      const dbl2 v = dbl2{buf[p] + RPDE_T(e)[2 * u], buf[p + 1] + RPDE_T(e)[2 * u + 1]};
      dst[m >> 1] = v;
      buf[p] = v.x; buf[p + 1] = v.y;
    }
    if (tid == 0) {
      const double vn = buf[pidx(N)] + RPDE_T(e)[16];
      ((gmem_t)(a.div + off))[N] = vn;
      buf[pidx(N)] = vn;
    }
  }
  RPDE_SYNC(blk);

  // ---- g_i = p0_i div_i + p1_i div_{i+2} + p2_i div_{i+4}, i < N - 1, parity de-interleaved
  RPDE_PHASE(blk, tid) {
    cgmem2_t t0 = (cgmem2_t)a.p0, t1 = (cgmem2_t)a.p1, t2 = (cgmem2_t)a.p2;
    gmem_t ge = (gmem_t)(a.g + off), go = (gmem_t)(a.g + off + a.half);
    dbl2 c0[8], c1[8], c2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int q = tid + u * T; c0[u] = t0[q]; c1[u] = t1[q]; c2[u] = t2[q]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { RPDE_PIN(c0[u].x); RPDE_PIN(c0[u].y); RPDE_PIN(c1[u].x); RPDE_PIN(c1[u].y); RPDE_PIN(c2[u].x); RPDE_PIN(c2[u].y); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T);
      const double gx = c0[u].x * buf[pidx(m)] + c1[u].x * buf[pidx(m + 2)] + c2[u].x * buf[pidx(m + 4)];
      const double gy = c0[u].y * buf[pidx(m + 1)] + c1[u].y * buf[pidx(m + 3)] + c2[u].y * buf[pidx(m + 5)];
      if (m < n) ge[m >> 1] = gx;
      if (m + 1 < n) go[m >> 1] = gy;
    }
  }
}

}  // namespace rpde
