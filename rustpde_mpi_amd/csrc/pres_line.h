// S9 of the confined step as one whole-line kernel: the pressure update (navier_eq.rs:137-143, `update_pres`), per x-line
// (one y row j of the YX arrays)
//
//   pres += to_ortho_x( S_y pseu ) / dt  -  nu div            S_y: the y stencil of the pressure space across the lines
//   gx    = dscale d/dx pres                                  (rows j and j - 2 of the pseudo-pressure); d/dx p is kept for
//                                                             the next step's S3 (funspace `gradient`, src/field.rs:127-129)
//
// The pseudo-pressure arrives from the second eigen-transform with its x coefficients parity de-interleaved (even ones first,
// the odd ones `half` columns further: G2 stores its two blocks side by side).  The line program of the stage (engine.cc
// S9: OP_LOADX, OP_STEN, two accumulating loads, store, OP_CDIFF, store) runs 512 threads and two barriers per op; here 256
// threads per 4097-point line (64 for 1025) and one padded line buffer like div_line.h: the element-wise part in pairs
// (thread t owns k = 2 (t + u T)), the derivative as a chunked suffix sum (thread t owns the 16 coefficients of chunk
// T - 1 - t; DPP scan in a wave, wave totals through LDS).
#pragma once
#include "div_line.h"

namespace rpde {

struct PresLineArgs {
  const double* ps = nullptr;     // pseudo-pressure rows (N - 1 composite coefficients per line, parity blocks side by side)
  const double* div = nullptr;    // divergence, N + 1 orthonormal coefficients per line
  double* pres = nullptr;         // pressure, N + 1 orthonormal coefficients per line, updated in place
  double* gx = nullptr;           // d/dx pres, N + 1 coefficients per line
  long ld = 0;                    // all arrays share the pitch
  int nlines = 0, line0 = 0;      // local lines, global index of the first one
  int N = 0, my = 0, half = 0;    // rows >= my of the pseudo-pressure do not exist (their stencil tap of row j - 2 does)
  double sdt = 0.0, nu = 0.0, dscale = 1.0;   // 1 / dt, viscosity, 1 / scale_x
  const double* lowy = nullptr;   // y stencil S[j, j - 2] of the pressure space (indexed with j - 2)
  const double* lowx = nullptr;   // x stencil S[k, k - 2] (indexed with k - 2), N - 1 entries
  int* nanflag = nullptr;         // raised when a NaN is stored (Integrate::exit on the device); may be null
};
RPDE_HD inline bool pres_line_ok(const PresLineArgs& a) {
  return (a.N == 256 || a.N == 1024 || a.N == 4096) && a.ps && a.div && a.pres && a.gx && a.lowy && a.lowx &&
         (((size_t)a.ps | (size_t)a.div | (size_t)a.pres | (size_t)a.gx | (size_t)a.lowx) & 15) == 0 && (a.ld & 1) == 0 &&
         a.ld > a.N + 1 && a.half > 0 && (a.half & 1) == 0;
}

template <int N>
RPDE_DEV void pres_line(Blk& blk, const PresLineArgs& a) {
  using G = HdctGeom<N>;
  constexpr int T = G::T, NW = G::NW;
  lds_t buf = (lds_t)blk.lds;
  lds_t scr = buf + G::SCR;
  const int line = blk.line, gline = line + a.line0;
  const long off = (long)line * a.ld;
  const bool has0 = gline < a.my, has2 = gline >= 2 && gline - 2 < a.my;
  const double cy = has2 ? ((tab_t)a.lowy)[gline - 2] : 0.0;
  const int n = N - 1;                                      // composite coefficients of a pseudo-pressure row
  auto pidx = [](int k) { return k + (k >> 4) + 2; };       // padded position of coefficient k (stride 17 per chunk of 16)

  // ---- pres_k += ([k < n] a_k + [k >= 2] lowx_{k-2} a_{k-2}) - nu div_k,  a = (row j + cy row j - 2) / dt.  The pair in
  // front of a thread's pair is its neighbour's (an L1 hit, like div_line.h); a row that does not exist is read through
  // the pointer of row j with a zero factor.
  RPDE_PHASE(blk, tid) {
    // (rows my, my + 1 have no row of their own but a row j - 2; rows 0, 1 the other way round: every line has one of the two)
    cgmem_t e0 = (cgmem_t)(a.ps + (has0 ? off : off - 2 * a.ld)), e2 = (cgmem_t)(a.ps + (has2 ? off - 2 * a.ld : off));
    cgmem_t o0 = e0 + a.half, o2 = e2 + a.half;
    cgmem2_t lx = (cgmem2_t)a.lowx, dv = (cgmem2_t)(a.div + off);
    gmem2_t pr = (gmem2_t)(a.pres + off);
    const double f0 = has0 ? a.sdt : 0.0, f2 = has2 ? a.sdt * cy : 0.0;
    bool bad = false;
    constexpr int NB = 2;                                   // pairs per batch: 11 loads per pair in flight (four pairs spill at 128 registers)
#pragma unroll
    for (int h = 0; h < 8 / NB; ++h) {
      double xe0[NB], xo0[NB], xe2[NB], xo2[NB], me0[NB], mo0[NB], me2[NB], mo2[NB];
      dbl2 lw[NB], d[NB], p[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int q = tid + (NB * h + i) * T;               // pair index: k = 2 q, 2 q + 1
        const int qm = q > 0 ? q - 1 : 0;                   // the pair in front (k - 2, k - 1); q = 0 has none (factor 0 below)
        xe0[i] = e0[q]; xo0[i] = o0[q]; xe2[i] = e2[q]; xo2[i] = o2[q];
        me0[i] = e0[qm]; mo0[i] = o0[qm]; me2[i] = e2[qm]; mo2[i] = o2[qm];
        lw[i] = lx[qm];
        d[i] = dv[q]; p[i] = pr[q];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        RPDE_PIN(xe0[i]); RPDE_PIN(xo0[i]); RPDE_PIN(xe2[i]); RPDE_PIN(xo2[i]); RPDE_PIN(me0[i]); RPDE_PIN(mo0[i]); RPDE_PIN(me2[i]);
        RPDE_PIN(mo2[i]); RPDE_PIN(lw[i].x); RPDE_PIN(lw[i].y); RPDE_PIN(d[i].x); RPDE_PIN(d[i].y); RPDE_PIN(p[i].x); RPDE_PIN(p[i].y);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int u = NB * h + i, q = tid + u * T, m = 2 * q;
        // a_m, a_{m+1} (the odd one of the last pair, k = N - 1, does not exist) and the two in front of them
        const double ae = f0 * xe0[i] + f2 * xe2[i];
        const double ao = (m + 1 < n) ? f0 * xo0[i] + f2 * xo2[i] : 0.0;
        const double be = (q > 0) ? f0 * me0[i] + f2 * me2[i] : 0.0;
        const double bo = (q > 0) ? f0 * mo0[i] + f2 * mo2[i] : 0.0;
        const double ce = ae + lw[i].x * be, co = ao + lw[i].y * bo;
        dbl2 v;
        v.x = (ce - a.nu * d[i].x) + p[i].x;
        v.y = (co - a.nu * d[i].y) + p[i].y;
        pr[q] = v;
        bad |= (v.x != v.x) | (v.y != v.y);
        buf[pidx(m)] = v.x;
        buf[pidx(m + 1)] = v.y;                              // m + 1 stays inside the group of 16
      }
#ifndef RPDE_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (tid == 0) {                                         // k = N: no own coefficient, the stencil tap of k - 2 = N - 2 only
      cgmem_t e0 = (cgmem_t)(a.ps + (has0 ? off : off - 2 * a.ld)), e2 = (cgmem_t)(a.ps + (has2 ? off - 2 * a.ld : off));
      const double f0 = has0 ? a.sdt : 0.0, f2 = has2 ? a.sdt * cy : 0.0;
      const double bn = f0 * e0[(N - 2) >> 1] + f2 * e2[(N - 2) >> 1];
      const double vn = (((tab_t)a.lowx)[N - 2] * bn - a.nu * ((cgmem_t)(a.div + off))[N]) + ((cgmem_t)(a.pres + off))[N];
      ((gmem_t)(a.pres + off))[N] = vn;
      bad |= (vn != vn);
      buf[pidx(N)] = vn;
      buf[pidx(N + 1)] = 0.0;                               // the derivative reads c_{k+1} up to k = N
    }
    if (bad && a.nanflag) *a.nanflag = 1;
  }
  RPDE_SYNC(blk);

  // ---- d_k = dscale sum_{j > k, j + k odd} 2 j c_j (d_0 halved, d_N = 0): suffix sums per parity (div_line.h)
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
    if (tid == 0) buf[pidx(N)] = 0.0;                       // d_N = 0
  }
  RPDE_SYNC(blk);

  // ---- the derivative leaves in pairs, coalesced
  RPDE_PHASE(blk, tid) {
    gmem2_t dst = (gmem2_t)(a.gx + off);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T), p = pidx(m);
      dst[m >> 1] = dbl2{buf[p], buf[p + 1]};
    }
    if (tid == 0) ((gmem_t)(a.gx + off))[N] = buf[pidx(N)];
  }
}

}  // namespace rpde
