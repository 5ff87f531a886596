// Column scans: the y-direction recurrences of the step on arrays whose rows are y-lines' elements
// ("YX" layout: row = y index, contiguous x).  One thread owns one column of one block of rows and
// marches along y; adjacent threads own adjacent columns, so every step of the recurrence is one
// perfectly coalesced row access and the coefficients of a row are wave-uniform (scalar loads).
// No LDS, no barriers, no transposes: the banded Helmholtz solve along y and the Chebyshev
// y-derivatives stream at HBM speed instead of going XY -> line kernel -> YX.
//
// The sequential dependency along y is cut into NB blocks of BR rows.  The Helmholtz solve reads its
// input TWICE and writes its output ONCE (round 2 made three read + write passes over a work array):
//   summary pass   a thread loads its block of rows into registers, runs B2 rows, forward substitution and back
//                  substitution from ZERO inflow states and keeps only the block-end states (no array is written);
//   carry pass     per column and parity: the forward inflow of every block (ascending, tabulated block transfer
//                  factors), then the backward inflow states (descending, tabulated 2 x 2 block transfer matrices;
//                  the response of a block's backward end state to its own forward inflow is tabulated too: `g`);
//   final pass     the same block again, now with its exact inflow states: the rows it stores are final.
// Per element the arithmetic is that of the reference's sequential sweeps:
//   MatVecFdma (B2 rows)      src/solver/matvec.rs:207-228
//   Fdma::fdma fwd / bwd      src/solver/fdma.rs:101-118
//   Chebyshev derivative      funspace gradient (src/field.rs:127-129), d_k = d_{k+2} + 2 (k+1) c_{k+1}
// Same source for the HIP kernels and the host emulation (the body is a function of (column, block)).
#pragma once
#include "platform.h"

namespace rpde {

constexpr int kColMaxFields = 3;
constexpr int kColBR = 32;             // rows per block: a thread keeps BR + 4 input rows in registers
constexpr int kColMaxRanks = 8;
constexpr int kColSumm = 7;            // doubles per column, field and rank in the cross-rank summary

// per-row tables of one Helmholtz-y solver (device pointers, indexed with the GLOBAL row; zero-padded behind the
// system) and the tables of this rank's blocks
struct ColHhTabs {
  const double *t0, *t1, *t2;          // B2 preconditioner rows: b_j = t0 w_j + t1 w_{j+2} + t2 w_{j+4}
  const double *q1;                    // forward substitution y_j = b_j + q1_j y_{j-2}
  const double *m1;                    // [NB][2]: block transfer of the forward chain, per parity
  const double *p2, *q2, *r2;          // back substitution x_j = p2_j y_j + q2_j x_{j+2} + r2_j x_{j+4}
  const double *m2;                    // [NB][2][4]: block transfer matrices of the backward chain
  const double *g;                     // [NB][2][2]: backward end state of a block per unit of its forward inflow
  // optional rank-one term (nullptr: none): out_j += kappa * h_j with kappa = sum over the input rows of w_row in_row,
  // one number per column (the velocity correction: the Chebyshev derivative followed by the Dirichlet projection is
  // local except for ONE entry of the right-hand side that carries a weighted sum of the whole column)
  const double *w, *h;
  // pencil-sharded runs: [nranks][14] = m1[2], m2[2][4], g[2][2] of every rank's rows taken as ONE block
  const double *rk;
};

struct ColHhArgs {
  int n;                 // rows of the banded system (composite size along y)
  int nin;               // valid rows of the input (rows >= nin read as zero)
  int ncols;             // columns (doubles per row that take part)
  int NB;                // number of blocks of kColBR rows on this rank
  long ld;               // pitch of all arrays (doubles)
  int nf;                // fields solved in one launch (grid.z)
  int row0;              // global index of row 0 of the (local) arrays = first row of block 0; even
  int jend;              // rows >= jend are not this rank's (the last rank: n)
  const double* in[kColMaxFields];   // right-hand side after the x part (orthonormal rows)
  double* out[kColMaxFields];        // solution rows n
  int shift[kColMaxFields];          // tap 0 of row j is input row j - shift (rows < 0 read as zero)
  int in_half;                       // > 0: the input columns are parity de-interleaved (column i of the output =
                                     // input column (i & 1) * in_half + (i >> 1)); 0: same columns
  int pair = 0;                      // 1: nf = 2 fields read the SAME input (the velocity correction): the block passes put the
                                     // two workgroups that read the same rows next to each other on one XCD (second read = L2 hit)
  ColHhTabs tab[kColMaxFields];
  double *v1, *s1;       // [nf][NB][2][ld]     block-end values / block inflow of the forward chain
  double *v2, *s2;       // [nf][NB][2][2][ld]  block-end states / block inflow states of the backward chain
  double *dotp, *kap;    // [nf][NB][ld] block parts of the rank-one sums, [nf][ld] the sums (fields with tab.w only)
  int* nanflag;          // raised when the final pass stores a NaN (Integrate::exit); may be null
  // pencil-sharded runs (rows split over the ranks): every rank reduces its blocks to ONE summary per column --
  // forward end values (2), backward end states (2 x 2), its part of the rank-one sum -- the summaries travel to
  // all ranks (one small exchange), and each rank derives the inflow states of its first / last block from them
  int nranks, rank;
  double* summ;          // [nf][kColSumm][ld]            this rank's summary (carry phase 1)
  const double* gath;    // [nranks][nf][kColSumm][ld]    all summaries (carry phase 2)
};

RPDE_HD inline long col_c1(const ColHhArgs& a, int f, int b, int par) { return (((long)f * a.NB + b) * 2 + par) * a.ld; }
RPDE_HD inline long col_c2(const ColHhArgs& a, int f, int b, int par, int c) {
  return ((((long)f * a.NB + b) * 2 + par) * 2 + c) * a.ld;
}
RPDE_HD inline long col_sm(const ColHhArgs& a, int r, int f, int k) { return (((long)r * a.nf + f) * kColSumm + k) * a.ld; }

// loads of the carry passes and of the derivative are issued in batches of kColBatch (they do not depend on
// the recurrence); one load in flight per thread would leave the kernels latency bound
constexpr int kColBatch = 8;
// the carry passes are ONE serial chain per thread over all blocks: their loads go out sixteen blocks at a time (the
// chain waits once per batch for memory: 128 blocks = 8 round trips per direction)
constexpr int kCarryBatch = 16;
constexpr int kCarryBatchBack = 8;     // backward chain: nine values per block (two end states, the forward inflow, M, g)

// one block of one column: FINAL = false keeps the block-end states of a run from zero inflow, FINAL = true
// runs from the exact inflow states and stores the rows.  All BR + 4 row loads are in flight at once.
template <bool FINAL>
RPDE_HD inline void colhh_block(const ColHhArgs& a, int f, int b, int i) {
  constexpr int BR = kColBR;
  const ColHhTabs& t = a.tab[f];
  const int ci = a.in_half ? (i & 1) * a.in_half + (i >> 1) : i;
  const double* __restrict__ w = a.in[f] + ci - (long)a.row0 * a.ld;   // indexed with the global row
  const int j0 = a.row0 + b * BR, jr = j0 - a.shift[f];
  const int j1 = (j0 + BR < a.jend) ? j0 + BR : a.jend;                 // rows [j0, j1) are this block's
  const int rmax = (a.nin < a.jend + 4) ? a.nin : a.jend + 4;            // behind the rank's rows: four halo rows
  double r[BR + 4];
#pragma unroll
  for (int u = 0; u < BR + 4; ++u) r[u] = (jr + u >= 0 && jr + u < rmax) ? w[(long)(jr + u) * a.ld] : 0.0;
  if (t.w) {
    if (!FINAL) {
      // this block's part of the column sum: the input rows [jr, jr + BR) that belong to this rank; the block that holds
      // the end of the system also the tail
      double dot = 0.0;
      const bool tail = b == a.NB - 1 && a.jend >= a.n;
#pragma unroll
      for (int u = 0; u < BR + 4; ++u)
        if ((u < BR && (jr + u < a.jend - a.shift[f] || tail)) || (u >= BR && tail)) dot += t.w[(jr + u > 0) ? jr + u : 0] * r[u];
      a.dotp[((long)f * a.NB + b) * a.ld + i] = dot;
    }
  }
  double ye = 0.0, yo = 0.0;             // previous element of the even / odd forward chain (j0 is even)
  double e1 = 0.0, e2 = 0.0, o1 = 0.0, o2 = 0.0;   // (most recent, the one before) of the even / odd backward chain
  if (FINAL) {
    ye = a.s1[col_c1(a, f, b, 0) + i]; yo = a.s1[col_c1(a, f, b, 1) + i];
    e1 = a.s2[col_c2(a, f, b, 0, 0) + i]; e2 = a.s2[col_c2(a, f, b, 0, 1) + i];
    o1 = a.s2[col_c2(a, f, b, 1, 0) + i]; o2 = a.s2[col_c2(a, f, b, 1, 1) + i];
  }
#pragma unroll
  for (int u = 0; u < BR; ++u) {         // B2 row + forward substitution, ascending, in place
    const int j = j0 + u;
    const double bj = t.t0[j] * r[u] + t.t1[j] * r[u + 2] + t.t2[j] * r[u + 4];
    double& yp = (u & 1) ? yo : ye;
    const double yn = bj + t.q1[j] * yp;
    yp = (j < j1) ? yn : yp;
    r[u] = yn;
  }
  if (!FINAL) {
    a.v1[col_c1(a, f, b, 0) + i] = ye;
    a.v1[col_c1(a, f, b, 1) + i] = yo;
  }
#pragma unroll
  for (int u = BR - 1; u >= 0; --u) {    // back substitution, descending, in place
    const int j = j0 + u;
    double& x1 = (u & 1) ? o1 : e1;
    double& x2 = (u & 1) ? o2 : e2;
    const double xj = t.p2[j] * r[u] + t.q2[j] * x1 + t.r2[j] * x2;
    x2 = (j < j1) ? x1 : x2;
    x1 = (j < j1) ? xj : x1;
    r[u] = xj;
  }
  if (!FINAL) {
    a.v2[col_c2(a, f, b, 0, 0) + i] = e1; a.v2[col_c2(a, f, b, 0, 1) + i] = e2;
    a.v2[col_c2(a, f, b, 1, 0) + i] = o1; a.v2[col_c2(a, f, b, 1, 1) + i] = o2;
  } else {
    double* __restrict__ out = a.out[f] + i - (long)a.row0 * a.ld;
    const double kap = t.w ? a.kap[(long)f * a.ld + i] : 0.0;
    bool bad = false;
#pragma unroll
    for (int u = 0; u < BR; ++u)
      if (j0 + u < j1) {
        const double x = t.w ? r[u] + kap * t.h[j0 + u] : r[u];
        out[(long)(j0 + u) * a.ld] = x;
        bad |= (x != x);
      }
    if (bad && a.nanflag) *a.nanflag = 1;
  }
}

// carry pass; thread = (column, parity).  Forward chain ascending, then the backward chain descending: the state a
// block hands down is its zero-inflow end state + g * (its forward inflow) + M * (its backward inflow state).
// PHASE 0: one rank -- inflow zero, block inflows and the rank-one sum are final.
// PHASE 1 (sharded): the same from zero inflow; what comes out is this rank's summary `summ`.
// PHASE 2 (sharded): the inflow of this rank from everybody's summaries and the rank tables, then the block inflows.
template <int PHASE>
RPDE_HD inline void colhh_carry(const ColHhArgs& a, int f, int i, int par) {
  const ColHhTabs& t = a.tab[f];
  double s = 0.0, s0 = 0.0, s1 = 0.0;
  if (PHASE == 2) {
    // s_r: forward inflow of rank r; (S0, S1)_r: backward inflow state of rank r, from the ranks above
    double sr[kColMaxRanks];
    double run = 0.0;
#pragma unroll
    for (int r = 0; r < kColMaxRanks; ++r) {
      sr[r] = run;
      if (r < a.nranks) run = t.rk[r * 14 + par] * run + a.gath[col_sm(a, r, f, par) + i];
    }
#pragma unroll
    for (int r = kColMaxRanks - 1; r >= 0; --r) {
      if (r == a.rank) s = sr[r];
      if (r < a.nranks && r > a.rank) {
        const double* m = t.rk + r * 14 + 2 + par * 4;
        const double* g = t.rk + r * 14 + 10 + par * 2;
        const double b0 = a.gath[col_sm(a, r, f, 2 + 2 * par) + i], b1 = a.gath[col_sm(a, r, f, 3 + 2 * par) + i];
        const double n0 = m[0] * s0 + m[1] * s1 + (b0 + g[0] * sr[r]);
        const double n1 = m[2] * s0 + m[3] * s1 + (b1 + g[1] * sr[r]);
        s0 = n0; s1 = n1;
      }
    }
    if (t.w && par == 0) {
      double k = 0.0;
      for (int r = 0; r < a.nranks; ++r) k += a.gath[col_sm(a, r, f, 6) + i];
      a.kap[(long)f * a.ld + i] = k;
    }
  } else if (par == 0) {                 // the column sum of the rank-one term (this rank's part)
    double k = 0.0;
    if (t.w) for (int b = 0; b < a.NB; ++b) k += a.dotp[((long)f * a.NB + b) * a.ld + i];
    if (PHASE == 0) { if (t.w) a.kap[(long)f * a.ld + i] = k; }
    else a.summ[col_sm(a, 0, f, 6) + i] = k;
  }
  // The block transfer factors of a batch are loaded TOGETHER with its block-end values, in front of the batch's stores: read
  // one by one inside the chain, every table entry is a memory round trip of its own (the compiler may not move a load
  // from `t.m1` across the store to `a.s1` in front of it: both are plain global pointers).
  for (int bb = 0; bb < a.NB; bb += kCarryBatch) {
    double v[kCarryBatch], mm[kCarryBatch];
#pragma unroll
    for (int u = 0; u < kCarryBatch; ++u) {
      v[u] = (bb + u < a.NB) ? a.v1[col_c1(a, f, bb + u, par) + i] : 0.0;
      mm[u] = (bb + u < a.NB) ? t.m1[(bb + u) * 2 + par] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kCarryBatch; ++u)
      if (bb + u < a.NB) {
        a.s1[col_c1(a, f, bb + u, par) + i] = s;
        s = mm[u] * s + v[u];
      }
  }
  if (PHASE == 1) a.summ[col_sm(a, 0, f, par) + i] = s;
  for (int bt = a.NB - 1; bt >= 0; bt -= kCarryBatchBack) {
    double v0[kCarryBatchBack], v1[kCarryBatchBack], fi[kCarryBatchBack], m[kCarryBatchBack][4], g[kCarryBatchBack][2];
#pragma unroll
    for (int u = 0; u < kCarryBatchBack; ++u) {
      const bool ok = bt - u >= 0;
      const int b = ok ? bt - u : 0;
      v0[u] = ok ? a.v2[col_c2(a, f, b, par, 0) + i] : 0.0;
      v1[u] = ok ? a.v2[col_c2(a, f, b, par, 1) + i] : 0.0;
      fi[u] = ok ? a.s1[col_c1(a, f, b, par) + i] : 0.0;   // written above by this thread
#pragma unroll
      for (int c = 0; c < 4; ++c) m[u][c] = ok ? t.m2[(b * 2 + par) * 4 + c] : 0.0;
#pragma unroll
      for (int c = 0; c < 2; ++c) g[u][c] = ok ? t.g[(b * 2 + par) * 2 + c] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kCarryBatchBack; ++u) {
      const int b = bt - u;
      if (b >= 0) {
        a.s2[col_c2(a, f, b, par, 0) + i] = s0;
        a.s2[col_c2(a, f, b, par, 1) + i] = s1;
        const double n0 = m[u][0] * s0 + m[u][1] * s1 + (v0[u] + g[u][0] * fi[u]);
        const double n1 = m[u][2] * s0 + m[u][3] * s1 + (v1[u] + g[u][1] * fi[u]);
        s0 = n0; s1 = n1;
      }
    }
  }
  if (PHASE == 1) { a.summ[col_sm(a, 0, f, 2 + 2 * par) + i] = s0; a.summ[col_sm(a, 0, f, 3 + 2 * par) + i] = s1; }
}

// ---------------------------------------------------------------------------------------------
// Chebyshev y-derivative of a YX array: d_j = d_{j+2} + 2 (j+1) c_{j+1}, c = S v (composite input: rows
// j and j-2 with the stencil `low`; low == nullptr: the input is orthonormal already), d_0 halved,
// everything times `scale`.  Suffix sums per parity: partial sums per block, carry, final pass.
struct ColDiffArgs {
  int nout;              // output rows (orthonormal size along y)
  int m;                 // input rows (composite size, or nout when low == nullptr)
  int ncols;
  int BR, NB;            // rows per block, blocks on this rank
  long ldi, ldo;
  const double* in;
  double* out;
  const double* low;     // stencil S[k+2, k] (length m) or nullptr
  double scale;
  double *vd, *sd;       // [NB][2][ldo] block sums / block inflow
  int row0;              // global index of row 0 of the (local) arrays = first row of block 0; even
  int jend;              // rows >= jend are not this rank's (the last rank: nout)
  int nranks, rank;      // pencil-sharded runs: the sums of the ranks above are this rank's inflow
  double* summ;          // [2][ldo]            this rank's sums per parity (carry phase 1)
  const double* gath;    // [nranks][2][ldo]    everybody's sums (carry phase 2)
};

RPDE_HD inline double coldiff_c(const ColDiffArgs& a, const double* v, int k) {   // orthonormal coefficient c_k of column v (v indexed with the global row)
  if (k >= a.nout) return 0.0;
  if (!a.low) return v[(long)k * a.ldi];
  double c = (k < a.m) ? v[(long)k * a.ldi] : 0.0;
  if (k >= 2) c += a.low[k - 2] * v[(long)(k - 2) * a.ldi];
  return c;
}

// FINAL = false: block sums only; FINAL = true: add the inflow and write the rows
template <bool FINAL>
RPDE_HD inline void coldiff_pass(const ColDiffArgs& a, int b, int i) {
  const double* __restrict__ v = a.in + i - (long)a.row0 * a.ldi;
  double* __restrict__ out = a.out + i - (long)a.row0 * a.ldo;
  const int j0 = a.row0 + b * a.BR, j1 = (j0 + a.BR < a.jend) ? j0 + a.BR : a.jend;
  double acc[2] = {0.0, 0.0};
  if (FINAL) { acc[0] = a.sd[((long)b * 2 + 0) * a.ldo + i]; acc[1] = a.sd[((long)b * 2 + 1) * a.ldo + i]; }
  for (int jt = j1 - 1; jt >= j0; jt -= kColBatch) {
    double c[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) c[u] = (jt - u >= j0) ? coldiff_c(a, v, jt - u + 1) : 0.0;
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) {
      const int j = jt - u;
      if (j >= j0) {
        acc[j & 1] += 2.0 * (double)(j + 1) * c[u];
        if (FINAL) out[(long)j * a.ldo] = acc[j & 1] * ((j == 0) ? 0.5 * a.scale : a.scale);
      }
    }
  }
  if (!FINAL) { a.vd[((long)b * 2 + 0) * a.ldo + i] = acc[0]; a.vd[((long)b * 2 + 1) * a.ldo + i] = acc[1]; }
}

// PHASE 0: one rank.  PHASE 1 (sharded): this rank's sum per parity -> summ.  PHASE 2 (sharded): inflow = the sums of
// the ranks above, then the block inflows.
template <int PHASE>
RPDE_HD inline void coldiff_carry(const ColDiffArgs& a, int i, int par) {
  double s = 0.0;
  if (PHASE == 2)
    for (int r = a.rank + 1; r < a.nranks; ++r) s += a.gath[((long)r * 2 + par) * a.ldo + i];
  for (int bt = a.NB - 1; bt >= 0; bt -= kCarryBatch) {
    double v[kCarryBatch];
#pragma unroll
    for (int u = 0; u < kCarryBatch; ++u) v[u] = (bt - u >= 0) ? a.vd[((long)(bt - u) * 2 + par) * a.ldo + i] : 0.0;
#pragma unroll
    for (int u = 0; u < kCarryBatch; ++u)
      if (bt - u >= 0) { if (PHASE != 1) a.sd[((long)(bt - u) * 2 + par) * a.ldo + i] = s; s += v[u]; }
  }
  if (PHASE == 1) a.summ[(long)par * a.ldo + i] = s;
}

}  // namespace rpde
