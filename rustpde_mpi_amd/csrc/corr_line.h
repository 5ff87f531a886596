// S8 of the confined step as one whole-line kernel: the x part of the velocity correction (navier_eq.rs:117-125),
// per x-line (one y row of the YX arrays)
//
//   velx += from_ortho_xD( dscale d/dx to_ortho_xN(a) ),   dscale = -1 / scale_x        (branch 0, input a = in[0])
//   vely += from_ortho_xD(              to_ortho_xN(b) )                                  (branch 1, input b = in[1])
//
// (funspace `to_ortho` / `gradient` / `from_ortho`, src/field.rs:113-129.)  The line program of the stage (engine.cc S8)
// runs stencil -> suffix-sum derivative -> S^T -> two first-order sweeps with 512 threads and two LDS slots.  Here the
// operators are folded the way the y part of the correction folds them (hostmath.h build_colcorr_tables): for the
// Dirichlet stencil the right-hand side of the projection, S^T d = d_k - d_{k+2}, is the LOCAL term 2 (k + 1) c_{k+1} of
// the derivative's recurrence, and only row 0 keeps a sum over the line -- a rank-one term kappa * h_k with
// kappa = sum_j w_j a_j.  Both branches are then "three taps + first-order sweep up + first-order sweep down":
//
//   rhs_k = t0_k r_k + t1_k r_{k+2} + t2_k r_{k+4},   r_k = in_{k - shift}      (shift 1 with the derivative, 2 without)
//   y_k = rhs_k + q1_k y_{k-2},    x_k = p2_k y_k + q2_k x_{k+2} + r2_k x_{k+4}
//
// 256 threads per 4097-point line (64 for 1025), one padded line buffer (35 KB: four workgroups per CU); the sweeps are
// the chunked scans of rhs_line.h (thread t owns k = 16 t .. 16 t + 15; chunk -> affine map of its inflow, prefix
// composition across the threads, exact re-run).  Per element the arithmetic is that of the column form of the same
// operation (colscan.h colhh_block with the tables of build_colcorr_tables).
#pragma once
#include "rhs_line.h"

namespace rpde {

struct CorrLineTabs {     // chunk-major for 16 elements per thread (rhs_line.h chunk_major16): ascending t0 t1 t2 q1, descending p2 q2 r2
  const double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *q1 = nullptr, *p2 = nullptr, *q2 = nullptr, *r2 = nullptr;
};
struct CorrLineArgs {
  const double* in[2] = {nullptr, nullptr};   // pseudo-pressure after the y part of the correction: N - 1 composite (Neumann) coefficients per line
  double* out[2] = {nullptr, nullptr};        // velx, vely (N - 1 composite coefficients per line), updated in place
  long ld = 0;                                 // all arrays share the pitch
  int nlines = 0, N = 0;
  CorrLineTabs tab[2];
  const double *w = nullptr, *h = nullptr;     // rank-one term of branch 0, natural order, zero behind N - 1 entries
  int* nanflag = nullptr;                      // raised when a NaN is stored (Integrate::exit on the device); may be null
};
RPDE_HD inline bool corr_line_ok(const CorrLineArgs& a) {
  bool ok = (a.N == 256 || a.N == 1024 || a.N == 4096) && (a.ld & 1) == 0 && a.ld > a.N + 1 && a.w && a.h;
  for (int b = 0; b < 2; ++b)
    ok = ok && a.in[b] && a.out[b] && (((size_t)a.in[b]) & 15) == 0 && (((size_t)a.out[b]) & 15) == 0 && a.tab[b].t0 && a.tab[b].r2;
  return ok;
}

// one branch: BR = 0 with the rank-one term (shift 1), BR = 1 without (shift 2)
template <int N, int BR>
RPDE_DEV void corr_line_branch(Blk& blk, const CorrLineArgs& a) {
  using G = HdctGeom<N>;
  constexpr int T = G::T, NW = G::NW, W = 6, SH = BR == 0 ? 1 : 2;
  constexpr int KS = 2 * NW * W;                          // scratch behind the prefix composition's: partial sums of kappa
  lds_t buf = (lds_t)blk.lds;
  lds_t scr = buf + G::SCR;
  const long off = (long)blk.line * a.ld;
  const int n = N - 1;
  const CorrLineTabs& tb = a.tab[BR];

  // ---- the input line into the padded buffer, shifted: r_k = in_{k - SH} at index k + k / 16 + 2, zeros around it
  RPDE_TLS(blk, double, kp, 1);
  RPDE_PHASE(blk, tid) {
    // (rows and tables through buffer descriptors: one 32-bit offset per thread, the block u in a scalar register -- line_vm.h)
    const RowBuf src = row_buf(a.in[BR] + off, 8L * N), wt = row_buf((BR == 0) ? a.w : a.in[BR] + off, 8L * N);
    dbl2 v[8], ww[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = row_ld2(src, 16 * tid, 16 * u * T);            // the pair m = 2 (tid + u T) <= N - 2: inside the row (ld > N + 1)
      if (BR == 0) ww[u] = row_ld2(wt, 16 * tid, 16 * u * T);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { RPDE_PIN(v[u].x); RPDE_PIN(v[u].y); if (BR == 0) { RPDE_PIN(ww[u].x); RPDE_PIN(ww[u].y); } }
    double dot = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T);
      const double x0 = v[u].x, x1 = (m + 1 < n) ? v[u].y : 0.0;   // in_{N-1} does not exist
      if (BR == 0) dot += ww[u].x * x0 + ww[u].y * x1;
      const int k0 = m + SH, k1 = m + 1 + SH;
      buf[k0 + (k0 >> 4) + 2] = x0;
      buf[k1 + (k1 >> 4) + 2] = x1;
    }
    RPDE_T(kp)[0] = dot;
    if (tid == 0) {
#pragma unroll
      for (int k = 0; k < SH; ++k) buf[k + (k >> 4) + 2] = 0.0;
#pragma unroll
      for (int k = N + SH; k <= N + 4; ++k) buf[k + (k >> 4) + 2] = 0.0;
    }
  }
#ifndef RPDE_EMU
  if (BR == 0) {   // wave totals of kappa wait in the scratch area until the output phase
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const double tot = sum_wave_scan(kp[0]);
    if (lane == ((T < 64) ? T - 1 : 63)) scr[KS + wave] = tot;
  }
#endif
  RPDE_SYNC(blk);

  // ---- three taps + forward sweep, thread t owns k = 16 t .. 16 t + 15 (ascending: the carry flows t - 1 -> t)
  RPDE_TLS(blk, double, y, 16);
  RPDE_TLS(blk, double, cm, 2 * W);
  RPDE_TLS(blk, double, qa, 16);
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * tid;
    const ChunkTab t0 = chunk_tab(tb.t0, T), t1 = chunk_tab(tb.t1, T), t2 = chunk_tab((BR == 1) ? tb.t2 : tb.t1, T), q1 = chunk_tab(tb.q1, T);
    double r[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) { const int k = k0 + i; r[i] = buf[k + (k >> 4) + 2]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) RPDE_T(qa)[i] = chunk_ld(q1, tid, i, T);
#pragma unroll
    for (int i = 0; i < 16; ++i) RPDE_PIN(RPDE_T(qa)[i]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // the taps of eight elements at a time (registers)
      double c0[8], c1[8], c2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int j = 8 * h + i; c0[i] = chunk_ld(t0, tid, j, T); c1[i] = chunk_ld(t1, tid, j, T); c2[i] = (BR == 1) ? chunk_ld(t2, tid, j, T) : 0.0; }
#pragma unroll
      for (int i = 0; i < 8; ++i) { RPDE_PIN(c0[i]); RPDE_PIN(c1[i]); if (BR == 1) RPDE_PIN(c2[i]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = 8 * h + i, k = k0 + j;
        double b = c0[i] * r[j] + c1[i] * r[j + 2];
        if (BR == 1) b += c2[i] * r[j + 4];                 // the derivative branch has two taps
        RPDE_T(y)[j] = (k < n) ? b : 0.0;
      }
#ifndef RPDE_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {                     // chunk -> affine map of its inflow (first order)
      double z = 0.0, m11 = 1.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ei = par + 2 * i;
        const bool ok = k0 + ei < n;
        const double q = RPDE_T(qa)[ei];
        z = ok ? RPDE_T(y)[ei] + q * z : z;
        m11 = ok ? q * m11 : m11;
      }
      double* m = RPDE_T(cm) + par * W;
      m[0] = m11; m[1] = 0.0; m[2] = 0.0; m[3] = 1.0; m[4] = z; m[5] = 0.0;
    }
  }
#ifdef RPDE_EMU
  chunk_prefix<1, T>(blk, scr, cm_st);
#else
  chunk_prefix<1, T>(blk, scr, cm);
#endif
  RPDE_SYNC(blk);                                           // everybody has read the right-hand side
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * tid;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double x1 = RPDE_T(cm)[par * W + 4];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ei = par + 2 * i;
        const bool ok = k0 + ei < n;
        x1 = ok ? RPDE_T(y)[ei] + RPDE_T(qa)[ei] * x1 : x1;
        RPDE_T(y)[ei] = x1;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int k = k0 + i; buf[k + (k >> 4) + 2] = RPDE_T(y)[i]; }   // y for the descending sweep
  }
  RPDE_SYNC(blk);

  // ---- backward sweep, descending: thread t owns the chunk of thread T - 1 - t (the carry flows t - 1 -> t again)
  RPDE_TLS(blk, double, bb, 16);
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * (T - 1 - tid);
    const ChunkTab p2 = chunk_tab(tb.p2, T), q2 = chunk_tab(tb.q2, T), r2 = chunk_tab(tb.r2, T);
    {
      double pp[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) pp[i] = chunk_ld(p2, tid, i, T);
#pragma unroll
      for (int i = 0; i < 16; ++i) RPDE_PIN(pp[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int k = k0 + i; RPDE_T(bb)[i] = pp[i] * buf[k + (k >> 4) + 2]; }
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double qq[8], rr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int ei = 14 + par - 2 * i; qq[i] = chunk_ld(q2, tid, ei, T); rr[i] = chunk_ld(r2, tid, ei, T); }
#pragma unroll
      for (int i = 0; i < 8; ++i) { RPDE_PIN(qq[i]); RPDE_PIN(rr[i]); }
      double z1 = 0.0, z2 = 0.0, a11 = 1.0, a12 = 0.0, a21 = 0.0, a22 = 1.0;   // state = (most recent value, the one before)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ei = 14 + par - 2 * i;
        const bool ok = k0 + ei < n;
        const double q = qq[i], r = rr[i];
        const double nz = RPDE_T(bb)[ei] + q * z1 + r * z2;
        const double n1 = q * a11 + r * a21, n2 = q * a12 + r * a22;
        z2 = ok ? z1 : z2; z1 = ok ? nz : z1;
        a21 = ok ? a11 : a21; a11 = ok ? n1 : a11;
        a22 = ok ? a12 : a22; a12 = ok ? n2 : a12;
      }
      double* m = RPDE_T(cm) + par * W;
      m[0] = a11; m[1] = a12; m[2] = a21; m[3] = a22; m[4] = z1; m[5] = z2;
#ifndef RPDE_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
#ifdef RPDE_EMU
  chunk_prefix<2, T>(blk, scr, cm_st);
#else
  chunk_prefix<2, T>(blk, scr, cm);
#endif
  RPDE_SYNC(blk);                                           // everybody has read y
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * (T - 1 - tid);
    const ChunkTab q2 = chunk_tab(tb.q2, T), r2 = chunk_tab(tb.r2, T);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double qq[8], rr[8];                                  // again (L1 / L2): not kept across the prefix
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int ei = 14 + par - 2 * i; qq[i] = chunk_ld(q2, tid, ei, T); rr[i] = chunk_ld(r2, tid, ei, T); }
#pragma unroll
      for (int i = 0; i < 8; ++i) { RPDE_PIN(qq[i]); RPDE_PIN(rr[i]); }
      double x1 = RPDE_T(cm)[par * W + 4], x2 = RPDE_T(cm)[par * W + 5];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int ei = 14 + par - 2 * i;
        const bool ok = k0 + ei < n;
        const double nx1 = RPDE_T(bb)[ei] + qq[i] * x1 + rr[i] * x2;
        x2 = ok ? x1 : x2; x1 = ok ? nx1 : x1;
        RPDE_T(bb)[ei] = x1;
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { const int k = k0 + i; buf[k + (k >> 4) + 2] = RPDE_T(bb)[i]; }
  }
  RPDE_SYNC(blk);

  // ---- velocity += x (+ kappa h), in pairs, coalesced
  double kappa = 0.0;
  if (BR == 0) {
#ifdef RPDE_EMU
    for (int t = 0; t < T; ++t) kappa += kp_st[(size_t)t];
#else
#pragma unroll
    for (int x = 0; x < NW; ++x) kappa += scr[KS + x];
#endif
  }
  RPDE_PHASE(blk, tid) {
    const RowBuf dst = row_buf(a.out[BR] + off, 8L * N), ht = row_buf((BR == 0) ? a.h : a.out[BR] + off, 8L * N);
    dbl2 o[8], hh[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      o[u] = row_ld2(dst, 16 * tid, 16 * u * T);
      if (BR == 0) hh[u] = row_ld2(ht, 16 * tid, 16 * u * T);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { RPDE_PIN(o[u].x); RPDE_PIN(o[u].y); if (BR == 0) { RPDE_PIN(hh[u].x); RPDE_PIN(hh[u].y); } }
    bool bad = false;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T);
      const int p = m + (m >> 4) + 2;
      dbl2 v = dbl2{buf[p], buf[p + 1]};                    // m + 1 stays inside the group of 16
      if (BR == 0) { v.x += kappa * hh[u].x; v.y += kappa * hh[u].y; }
      v.x += o[u].x; v.y += o[u].y;
      if (m + 1 < n) { row_st2(dst, 16 * tid, 16 * u * T, v); bad |= (v.x != v.x) | (v.y != v.y); }
      else if (m < n) { row_st1(dst, 16 * tid, 16 * u * T, v.x); bad |= (v.x != v.x); }
    }
    if (bad && a.nanflag) *a.nanflag = 1;
  }
}

template <int N>
RPDE_DEV void corr_line(Blk& blk, const CorrLineArgs& a) {
  corr_line_branch<N, 0>(blk, a);
  RPDE_SYNC(blk);                                           // the output phase has read the buffer
  corr_line_branch<N, 1>(blk, a);
}

}  // namespace rpde
