// S6 of the confined step as one whole-line kernel: per eigen row r of the x operator (a y-line of the XY array behind G1)
//
//   g = B2_y f                                          Poisson::solve_par, y preconditioner (src/solver/poisson.rs:222-229,
//                                                       MatVecFdma rows, src/solver/matvec.rs:207-228)
//   (A_y + lam_r C_y) p = g                             one four-diagonal system per eigenvalue, swept at setup
//                                                       (FdmaTensor::solve, src/solver/fdma_tensor.rs:219-233; Fdma::fdma,
//                                                       src/solver/fdma.rs:101-118: forward and backward substitution)
//
// The line program of the stage (engine.cc S6: load, OP_MV3, OP_REC1, OP_REC2, store) runs 512 threads with two LDS slots
// (two workgroups per CU) and reads the row's four factor tables in its own chunking (10 elements per thread).  Here: 256
// threads per 4097-point line (64 for 1025), ONE padded line buffer (35 KB: four workgroups per CU), the sweeps as the
// chunked scans of rhs_line.h / corr_line.h (thread t owns k = 16 t .. 16 t + 15; chunk -> affine map of its inflow, prefix
// composition across the threads, exact re-run) with the row's factors in a second, 16-element chunk-major copy
// (PoissonOp::rows16).  Per element the arithmetic is that of the reference's sequential sweeps.  The stage stays bound
// by its tables: four factor rows per line on top of the line itself (6 x 8 bytes per point).
//
// DERIVE (round 6): of the four factor rows only p2 = 1 / d (d: the swept diagonal) needs the sequential sweep of the setup.  The
// row's matrix is  c1 B + mu A  (B = peye . S: low = 0, up1 = 1, up2 = 0; A = pinv . S; mu = lam_r + alpha) and the sweep leaves
//   l_k = low_k / d_{k-2},   up1_k <- up1_k - l_k up2_{k-2},   up2_k unchanged               (Fdma sweep, src/solver/fdma.rs:73-82)
// so   q1_k = -l_k = -(mu aL_k) p2_{k-2},   r2_k = -(mu aU2_k) p2_k,   q2_k = -((c1 [k+2<n] + mu aU1_k) - l_k (mu aU2_{k-2})) p2_k
// follow from p2, mu and four one-dimensional tables of A and B that every line shares (L2).  The kernel reads ONE factor row per
// line instead of four (3 x 8 bytes per point instead of 6) and spends nine multiply-adds per point on the others; products with
// the rounded reciprocal p2 stand where the setup divides by d: the factors differ from the tabulated ones in their last bit, the
// solution by what that is worth (tests/test_emu_parity.py test_s6_derived_factors; RPDE_S6_DERIVE=0: the four tables).
#pragma once
#include "rhs_line.h"

namespace rpde {

struct ProwLineArgs {
  const double* in = nullptr;     // G1's product: N - 1 coefficients per line (eigen row), XY layout
  double* out = nullptr;          // the solved rows, same shape
  long ld = 0;                    // both arrays share the pitch
  int nlines = 0, line0 = 0;      // local lines, index of the first one in the factor tables
  int N = 0;
  const double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;   // B2 rows of the y axis, chunk-major ascending (rhs_line.h chunk_major16)
  const double *q1 = nullptr;                                  // per line (pitch tabld): forward substitution, chunk-major ascending
  const double *p2 = nullptr, *q2 = nullptr, *r2 = nullptr;   // per line: back substitution, chunk-major DESCENDING
  long tabld = 0;                 // = N doubles: 16 T entries per line
  int keep = 1;                   // 4097-point lines: the KEEP form of prow_line (three workgroups per CU); 0: the factors read twice at four (RPDE_S6_KEEP, A/B)
  int zero0 = 0;                  // 1: element 0 of the lines of factor row 0 leaves as 0 -- `pseu[0, 0] = 0` (solve_pres, navier_eq.rs:158-162)
                                  // of the periodic step rides in the store (both parts of wavenumber 0)
  // DERIVE form (see the head of the file): q1 / q2 / r2 may be null; p2 as above, plus
  int derive = 0;
  const double* mu = nullptr;     // [factor rows] lam_r + alpha, indexed like the factor tables (global row)
  const double* aLa = nullptr;    // A.low, chunk-major ASCENDING (the forward pass)
  const double *aLd = nullptr, *aU1d = nullptr, *aU2d = nullptr, *aU2sd = nullptr, *b1d = nullptr;   // chunk-major DESCENDING: A.low, A.up1, A.up2,
                                  // A.up2 shifted (entry k = A.up2_{k-2}), c1 B.up1
  int tdiv = 1;                   // lines per factor row: 2 in the periodic step, where the real and the imaginary part of a wavenumber's
                                  // row are two consecutive real lines (engine.cc build_periodic: real-view transposes around S6)
};
RPDE_HD inline bool prow_line_ok(const ProwLineArgs& a) {
  const bool tabs = a.derive ? (a.mu && a.aLa && a.aLd && a.aU1d && a.aU2d && a.aU2sd && a.b1d) : (a.q1 && a.q2 && a.r2);
  return (a.N == 256 || a.N == 1024 || a.N == 2048 || a.N == 4096) && (a.tdiv == 1 || a.tdiv == 2) && a.in && a.out && a.t0 && a.t1 && a.t2 && a.p2 && tabs &&
         ((((size_t)a.in) | ((size_t)a.out)) & 15) == 0 && (a.ld & 1) == 0 && a.ld > a.N + 1 && a.tabld == a.N;
}

// DERIVE: q2 / r2 of the eight entries ei = 14 + par - 2 i of a thread's descending chunk from pp[j] = p2_{k0 - 2 + j}
template <int T>
RPDE_DEV void prow_derive(const ProwLineArgs& a, double mu, int tid, int par, const double (&pp)[18], double (&qq)[8], double (&rr)[8]) {
  tab_t aL = (tab_t)a.aLd, aU1 = (tab_t)a.aU1d, aU2 = (tab_t)a.aU2d, aU2s = (tab_t)a.aU2sd, b1 = (tab_t)a.b1d;
#pragma unroll
  for (int h = 0; h < 2; ++h) {                             // four elements at a time: the band tables come out of the L2 / L1, a short wait
    double tl[4], t1[4], t2[4], ts[4], tb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ei = 14 + par - 2 * (4 * h + j);
      tl[j] = aL[ei * T + tid]; t1[j] = aU1[ei * T + tid]; t2[j] = aU2[ei * T + tid]; ts[j] = aU2s[ei * T + tid]; tb[j] = b1[ei * T + tid];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { RPDE_PIN(tl[j]); RPDE_PIN(t1[j]); RPDE_PIN(t2[j]); RPDE_PIN(ts[j]); RPDE_PIN(tb[j]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 4 * h + j, ei = 14 + par - 2 * i;
      const double l = (mu * tl[j]) * pp[ei];               // p2_{k-2}
      const double u1 = (tb[j] + mu * t1[j]) - l * (mu * ts[j]);
      qq[i] = -(u1 * pp[ei + 2]);
      rr[i] = -((mu * t2[j]) * pp[ei + 2]);
    }
  }
}

// KEEP: the back-substitution factors q2, r2 of the row stay in registers across the prefix composition (32 doubles; a budget of
// three waves per SIMD) instead of being read a second time -- they are the row's own (no other line shares them), and between the
// two reads the L2 has seen 24 MB of other rows: the second read came from HBM (PMC ratio of the stage 1.25)
template <int N, bool KEEP = false, bool DER = false>
RPDE_DEV void prow_line(Blk& blk, const ProwLineArgs& a) {
  using G = HdctGeom<N>;
  constexpr int T = G::T, W = 6;
  static_assert(!DER || KEEP, "the derived factors are computed once and kept");
  lds_t buf = (lds_t)blk.lds;
  lds_t scr = buf + G::SCR;
  const long off = (long)blk.line * a.ld;
  const long toff = (long)((blk.line + a.line0) / a.tdiv) * a.tabld;
  const int n = N - 1;
  const double mu = DER ? ((tab_t)a.mu)[(blk.line + a.line0) / a.tdiv] : 0.0;

  // ---- the line into the padded buffer: f_k at index k + k / 16 + 2, zeros behind it (the taps reach k + 4)
  RPDE_PHASE(blk, tid) {
    const RowBuf src = row_buf(a.in + off, 8L * N);
    dbl2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = row_ld2(src, 16 * tid, 16 * u * T);      // the pair m = 2 (tid + u T) <= N - 2: inside the row (ld > N + 1)
#pragma unroll
    for (int u = 0; u < 8; ++u) { RPDE_PIN(v[u].x); RPDE_PIN(v[u].y); }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T), q = m + (m >> 4) + 2;
      buf[q] = v[u].x;
      buf[q + 1] = (m + 1 < n) ? v[u].y : 0.0;                // f_{N-1} does not exist; m + 1 stays inside the group of 16
    }
    if (tid == 0) {
#pragma unroll
      for (int k = N; k <= N + 4; ++k) buf[k + (k >> 4) + 2] = 0.0;
    }
  }
  RPDE_SYNC(blk);

  // ---- B2 rows + forward substitution, thread t owns k = 16 t .. 16 t + 15 (ascending: the carry flows t - 1 -> t)
  //   b_k = t0_k f_k + t1_k f_{k+2} + t2_k f_{k+4} (k < N - 1; the last tap only for k < N - 3),  y_k = b_k + q1_k y_{k-2}
  RPDE_TLS(blk, double, y, 16);
  RPDE_TLS(blk, double, cm, 2 * W);
  RPDE_TLS(blk, double, qa, 16);
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * tid;
    const ChunkTab t0 = chunk_tab(a.t0, T), t1 = chunk_tab(a.t1, T), t2 = chunk_tab(a.t2, T);
    double r[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) { const int k = k0 + i; r[i] = buf[k + (k >> 4) + 2]; }
    if constexpr (DER) {
      // q1_k = -(mu aL_k) p2_{k-2}: the ascending chunk t of the row sits in column T - 1 - t of the descending table; the two
      // entries in front of it are the last two of the chunk before (aL_0 = aL_1 = 0: the first thread needs none)
      tab_t p2 = (tab_t)(a.p2 + toff), aL = (tab_t)a.aLa;
      const int col = T - 1 - tid, colp = (tid > 0) ? col + 1 : col;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double pm[8], al[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int i = 8 * h + j; pm[j] = (i < 2) ? p2[(14 + i) * T + colp] : p2[(i - 2) * T + col]; al[j] = aL[i * T + tid]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { RPDE_PIN(pm[j]); RPDE_PIN(al[j]); }
#pragma unroll
        for (int j = 0; j < 8; ++j) RPDE_T(qa)[8 * h + j] = -((mu * al[j]) * pm[j]);
      }
    } else {
      const ChunkTab q1 = chunk_tab(a.q1 + toff, T);
#pragma unroll
      for (int i = 0; i < 16; ++i) RPDE_T(qa)[i] = chunk_ld(q1, tid, i, T);
#pragma unroll
      for (int i = 0; i < 16; ++i) RPDE_PIN(RPDE_T(qa)[i]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                           // the band rows of eight elements at a time (registers)
      double c0[8], c1[8], c2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int j = 8 * h + i; c0[i] = chunk_ld(t0, tid, j, T); c1[i] = chunk_ld(t1, tid, j, T); c2[i] = chunk_ld(t2, tid, j, T); }
#pragma unroll
      for (int i = 0; i < 8; ++i) { RPDE_PIN(c0[i]); RPDE_PIN(c1[i]); RPDE_PIN(c2[i]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int j = 8 * h + i, k = k0 + j;
        double b = c0[i] * r[j] + c1[i] * r[j + 2];
        b += (k < n - 2) ? c2[i] * r[j + 4] : 0.0;
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
  RPDE_SYNC(blk);                                           // everybody has read the line
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

  // ---- back substitution, descending: thread t owns the chunk of thread T - 1 - t (the carry flows t - 1 -> t again)
  //   x_k = p2_k y_k + q2_k x_{k+2} + r2_k x_{k+4}
  RPDE_TLS(blk, double, bb, 16);
  RPDE_TLS(blk, double, kq, KEEP ? 16 : 1);
  RPDE_TLS(blk, double, kr, KEEP ? 16 : 1);
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * (T - 1 - tid);
    const ChunkTab p2 = chunk_tab(a.p2 + toff, T), q2 = chunk_tab(DER ? a.p2 : a.q2 + toff, T), r2 = chunk_tab(DER ? a.p2 : a.r2 + toff, T);   // (DER: q2 / r2 unused)
    double pp[18];                                          // DER: pp[i + 2] = p2_{k0 + i}, pp[0], pp[1] = the two entries below the chunk
    {
#pragma unroll
      for (int i = 0; i < 16; ++i) pp[i + 2] = chunk_ld(p2, tid, i, T);
      if constexpr (DER) {
        const int cp = (tid + 1 < T) ? tid + 1 : tid;       // the chunk below (k0 = 0 needs none: aL_0 = aL_1 = 0)
        pp[0] = chunk_ld(p2, cp, 14, T); pp[1] = chunk_ld(p2, cp, 15, T);
        RPDE_PIN(pp[0]); RPDE_PIN(pp[1]);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) RPDE_PIN(pp[i + 2]);
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int k = k0 + i; RPDE_T(bb)[i] = pp[i + 2] * buf[k + (k >> 4) + 2]; }
    }
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double qq[8], rr[8];
      if constexpr (DER) {
        prow_derive<T>(a, mu, tid, par, pp, qq, rr);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int ei = 14 + par - 2 * i; qq[i] = chunk_ld(q2, tid, ei, T); rr[i] = chunk_ld(r2, tid, ei, T); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { RPDE_PIN(qq[i]); RPDE_PIN(rr[i]); }
      }
      if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { RPDE_T(kq)[8 * par + i] = qq[i]; RPDE_T(kr)[8 * par + i] = rr[i]; }
      }
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
    const ChunkTab q2 = chunk_tab(DER ? a.p2 : a.q2 + toff, T), r2 = chunk_tab(DER ? a.p2 : a.r2 + toff, T);
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      double qq[8], rr[8];
      if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { qq[i] = RPDE_T(kq)[8 * par + i]; rr[i] = RPDE_T(kr)[8 * par + i]; }
      } else {                                              // again: not kept across the prefix
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int ei = 14 + par - 2 * i; qq[i] = chunk_ld(q2, tid, ei, T); rr[i] = chunk_ld(r2, tid, ei, T); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { RPDE_PIN(qq[i]); RPDE_PIN(rr[i]); }
      }
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

  // ---- the solution leaves in pairs, coalesced
  RPDE_PHASE(blk, tid) {
    const RowBuf dst = row_buf(a.out + off, 8L * N);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T);
      const int p = m + (m >> 4) + 2;
      dbl2 v = dbl2{buf[p], buf[p + 1]};
      if (m == 0 && a.zero0 && (blk.line + a.line0) / a.tdiv == 0) v.x = 0.0;
      if (m + 1 < n) row_st2(dst, 16 * tid, 16 * u * T, v);
      else if (m < n) row_st1(dst, 16 * tid, 16 * u * T, v.x);
    }
  }
}

}  // namespace rpde
