// Column scans: the y-direction recurrences of the step on arrays whose rows are y-lines' elements
// ("YX" layout: row = y index, contiguous x).  One thread owns one column of one block of rows and
// marches along y; adjacent threads own adjacent columns, so every step of the recurrence is one
// perfectly coalesced row access and the coefficients of a row are wave-uniform (scalar loads).
// No LDS, no barriers, no transposes: the banded Helmholtz solve along y and the Chebyshev
// y-derivatives stream at HBM speed instead of going XY -> line kernel -> YX.
//
// The sequential dependency along y is cut into NB blocks of BR rows: a block first runs from a ZERO
// inflow state (pass A), a tiny per-column pass turns the block-end states into block inflow
// states with the tabulated block transfer matrices (carry pass), and the last pass adds the
// tabulated homogeneous response, x_j = x0_j + h1_j s1 + h2_j s2.  Per element the arithmetic is
// that of the reference's sequential sweeps:
//   MatVecFdma (B2 rows)      src/solver/matvec.rs:207-228
//   Fdma::fdma fwd / bwd      src/solver/fdma.rs:101-118
//   Chebyshev derivative      funspace gradient (src/field.rs:127-129), d_k = d_{k+2} + 2 (k+1) c_{k+1}
// Same source for the HIP kernels and the host emulation (the body is a function of (column, block)).
#pragma once
#include "platform.h"

namespace rpde {

constexpr int kColMaxFields = 3;

// per-row tables of one Helmholtz-y solver (device pointers; rows = n)
struct ColHhTabs {
  const double *t0, *t1, *t2;          // B2 preconditioner rows: b_j = t0 w_j + t1 w_{j+2} + t2 w_{j+4}
  const double *q1, *h1a;              // forward substitution y_j = b_j + q1_j y_{j-2}; response to a unit inflow
  const double *m1;                    // [NB][2]: block transfer of the forward chain, per parity
  const double *p2, *q2, *r2;          // back substitution x_j = p2_j y_j + q2_j x_{j+2} + r2_j x_{j+4}
  const double *h1b, *h2b;             // responses to the unit inflow states (1,0) / (0,1)
  const double *m2;                    // [NB][2][4]: block transfer matrices of the backward chain
};

struct ColHhArgs {
  int n;                 // rows of the banded system (composite size along y)
  int nin;               // valid rows of the input (rows >= nin read as zero)
  int ncols;             // columns (doubles per row that take part)
  int BR, NB;            // rows per block (even), number of blocks
  long ld;               // pitch of all arrays (doubles)
  int nf;                // fields solved in one launch (grid.z)
  const double* in[kColMaxFields];   // right-hand side after the x part (orthonormal rows)
  double* z[kColMaxFields];          // work array (zero-inflow solutions), rows n
  double* out[kColMaxFields];        // solution rows n
  ColHhTabs tab[kColMaxFields];
  double *v1, *s1;       // [nf][NB][2][ld]     block-end values / block inflow of the forward chain
  double *v2, *s2;       // [nf][NB][2][2][ld]  block-end states / block inflow states of the backward chain
  int* nanflag;          // raised when the final pass stores a NaN (Integrate::exit); may be null
};

RPDE_HD inline long col_c1(const ColHhArgs& a, int f, int b, int par) { return (((long)f * a.NB + b) * 2 + par) * a.ld; }
RPDE_HD inline long col_c2(const ColHhArgs& a, int f, int b, int par, int c) {
  return ((((long)f * a.NB + b) * 2 + par) * 2 + c) * a.ld;
}

// Rows are processed in batches of kColBatch: the loads of a batch are issued together (they do not
// depend on the recurrence), then the batch is computed and stored.  One load in flight per thread
// would leave the kernels latency bound (measured: 3.0 TB/s with 8192 resident waves x 512 B).
constexpr int kColBatch = 8;

// pass A: B2 rows + forward substitution from a zero inflow, ascending rows of block b
RPDE_HD inline void colhh_fwd(const ColHhArgs& a, int f, int b, int i) {
  const ColHhTabs& t = a.tab[f];
  const double* __restrict__ w = a.in[f] + i;
  double* __restrict__ z = a.z[f] + i;
  const int j0 = b * a.BR, j1 = (j0 + a.BR < a.n) ? j0 + a.BR : a.n;
  auto rd = [&](int j) { return j < a.nin ? w[(long)j * a.ld] : 0.0; };
  double w0 = rd(j0), w1 = rd(j0 + 1), w2 = rd(j0 + 2), w3 = rd(j0 + 3);
  double ze = 0.0, zo = 0.0;   // previous element of the even / odd chain (j0 is even)
  for (int jb = j0; jb < j1; jb += kColBatch) {
    double wn[kColBatch], zz[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) wn[u] = rd(jb + 4 + u);
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) {
      const int j = jb + u;
      const int jc = j < j1 ? j : j1 - 1;               // clamped table index; rows past the block are not stored
      double bj = t.t0[jc] * w0 + t.t1[jc] * w2;
      bj += (j < a.n - 2) ? t.t2[jc] * wn[u] : 0.0;
      double& zp = (u & 1) ? zo : ze;
      const double zj = bj + t.q1[jc] * zp;
      zp = (j < j1) ? zj : zp;
      zz[u] = zj;
      w0 = w1; w1 = w2; w2 = w3; w3 = wn[u];
    }
#pragma unroll
    for (int u = 0; u < kColBatch; ++u)
      if (jb + u < j1) z[(long)(jb + u) * a.ld] = zz[u];
  }
  a.v1[col_c1(a, f, b, 0) + i] = ze;
  a.v1[col_c1(a, f, b, 1) + i] = zo;
}

// carry of the forward chain: inflow of every block, ascending; thread = (column, parity)
RPDE_HD inline void colhh_carry1(const ColHhArgs& a, int f, int i, int par) {
  const ColHhTabs& t = a.tab[f];
  double s = 0.0;
  for (int bb = 0; bb < a.NB; bb += kColBatch) {
    double v[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) v[u] = (bb + u < a.NB) ? a.v1[col_c1(a, f, bb + u, par) + i] : 0.0;
#pragma unroll
    for (int u = 0; u < kColBatch; ++u)
      if (bb + u < a.NB) {
        a.s1[col_c1(a, f, bb + u, par) + i] = s;
        s = t.m1[(bb + u) * 2 + par] * s + v[u];
      }
  }
}

// pass B: finish the forward chain, back substitution from a zero inflow, descending rows of block b
RPDE_HD inline void colhh_mid(const ColHhArgs& a, int f, int b, int i) {
  const ColHhTabs& t = a.tab[f];
  double* __restrict__ z = a.z[f] + i;
  const int j0 = b * a.BR, j1 = (j0 + a.BR < a.n) ? j0 + a.BR : a.n;
  const double se = a.s1[col_c1(a, f, b, 0) + i], so = a.s1[col_c1(a, f, b, 1) + i];
  double e1 = 0.0, e2 = 0.0, o1 = 0.0, o2 = 0.0;   // (most recent, the one before) of the even / odd chain
  for (int jt = j1 - 1; jt >= j0; jt -= kColBatch) {
    double zv[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) zv[u] = (jt - u >= j0) ? z[(long)(jt - u) * a.ld] : 0.0;
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) {
      const int j = jt - u;
      if (j >= j0) {
        const bool odd = j & 1;
        const double y = zv[u] + t.h1a[j] * (odd ? so : se);
        double& x1 = odd ? o1 : e1;
        double& x2 = odd ? o2 : e2;
        const double xj = t.p2[j] * y + t.q2[j] * x1 + t.r2[j] * x2;
        x2 = x1; x1 = xj;
        zv[u] = xj;
      }
    }
#pragma unroll
    for (int u = 0; u < kColBatch; ++u)
      if (jt - u >= j0) z[(long)(jt - u) * a.ld] = zv[u];
  }
  a.v2[col_c2(a, f, b, 0, 0) + i] = e1; a.v2[col_c2(a, f, b, 0, 1) + i] = e2;
  a.v2[col_c2(a, f, b, 1, 0) + i] = o1; a.v2[col_c2(a, f, b, 1, 1) + i] = o2;
}

// carry of the backward chain: inflow state of every block, descending; thread = (column, parity)
RPDE_HD inline void colhh_carry2(const ColHhArgs& a, int f, int i, int par) {
  const ColHhTabs& t = a.tab[f];
  double s0 = 0.0, s1 = 0.0;
  for (int bt = a.NB - 1; bt >= 0; bt -= kColBatch) {
    double v0[kColBatch], v1[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) {
      v0[u] = (bt - u >= 0) ? a.v2[col_c2(a, f, bt - u, par, 0) + i] : 0.0;
      v1[u] = (bt - u >= 0) ? a.v2[col_c2(a, f, bt - u, par, 1) + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) {
      const int b = bt - u;
      if (b >= 0) {
        a.s2[col_c2(a, f, b, par, 0) + i] = s0;
        a.s2[col_c2(a, f, b, par, 1) + i] = s1;
        const double* m = t.m2 + (b * 2 + par) * 4;
        const double n0 = m[0] * s0 + m[1] * s1 + v0[u];
        const double n1 = m[2] * s0 + m[3] * s1 + v1[u];
        s0 = n0; s1 = n1;
      }
    }
  }
}

// pass C: add the homogeneous response, store the solution
RPDE_HD inline void colhh_fin(const ColHhArgs& a, int f, int b, int i) {
  const ColHhTabs& t = a.tab[f];
  const double* __restrict__ z = a.z[f] + i;
  double* __restrict__ out = a.out[f] + i;
  const int j0 = b * a.BR, j1 = (j0 + a.BR < a.n) ? j0 + a.BR : a.n;
  const double e1 = a.s2[col_c2(a, f, b, 0, 0) + i], e2 = a.s2[col_c2(a, f, b, 0, 1) + i];
  const double o1 = a.s2[col_c2(a, f, b, 1, 0) + i], o2 = a.s2[col_c2(a, f, b, 1, 1) + i];
  bool bad = false;
  for (int jb = j0; jb < j1; jb += kColBatch) {
    double zv[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) zv[u] = (jb + u < j1) ? z[(long)(jb + u) * a.ld] : 0.0;
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) {
      const int j = jb + u;
      if (j < j1) {
        const bool odd = u & 1;   // j0 and the batch are even, so the parity of the row is that of u
        const double x = zv[u] + t.h1b[j] * (odd ? o1 : e1) + t.h2b[j] * (odd ? o2 : e2);
        out[(long)j * a.ld] = x;
        bad |= (x != x);
      }
    }
  }
  if (bad && a.nanflag) *a.nanflag = 1;
}

// ---------------------------------------------------------------------------------------------
// Chebyshev y-derivative of a YX array: d_j = d_{j+2} + 2 (j+1) c_{j+1}, c = S v (composite input: rows
// j and j-2 with the stencil `low`; low == nullptr: the input is orthonormal already), d_0 halved,
// everything times `scale`.  Suffix sums per parity: partial sums per block, carry, final pass.
struct ColDiffArgs {
  int nout;              // output rows (orthonormal size along y)
  int m;                 // input rows (composite size, or nout when low == nullptr)
  int ncols;
  int BR, NB;
  long ldi, ldo;
  const double* in;
  double* out;
  const double* low;     // stencil S[k+2, k] (length m) or nullptr
  double scale;
  double *vd, *sd;       // [NB][2][ldo] block sums / block inflow
};

RPDE_HD inline double coldiff_c(const ColDiffArgs& a, const double* v, int k) {   // orthonormal coefficient c_k of column v
  if (k >= a.nout) return 0.0;
  if (!a.low) return v[(long)k * a.ldi];
  double c = (k < a.m) ? v[(long)k * a.ldi] : 0.0;
  if (k >= 2) c += a.low[k - 2] * v[(long)(k - 2) * a.ldi];
  return c;
}

// FINAL = false: block sums only; FINAL = true: add the inflow and write the rows
template <bool FINAL>
RPDE_HD inline void coldiff_pass(const ColDiffArgs& a, int b, int i) {
  const double* __restrict__ v = a.in + i;
  const int j0 = b * a.BR, j1 = (j0 + a.BR < a.nout) ? j0 + a.BR : a.nout;
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
        if (FINAL) a.out[(long)j * a.ldo + i] = acc[j & 1] * ((j == 0) ? 0.5 * a.scale : a.scale);
      }
    }
  }
  if (!FINAL) { a.vd[((long)b * 2 + 0) * a.ldo + i] = acc[0]; a.vd[((long)b * 2 + 1) * a.ldo + i] = acc[1]; }
}

RPDE_HD inline void coldiff_carry(const ColDiffArgs& a, int i, int par) {
  double s = 0.0;
  for (int bt = a.NB - 1; bt >= 0; bt -= kColBatch) {
    double v[kColBatch];
#pragma unroll
    for (int u = 0; u < kColBatch; ++u) v[u] = (bt - u >= 0) ? a.vd[((long)(bt - u) * 2 + par) * a.ldo + i] : 0.0;
#pragma unroll
    for (int u = 0; u < kColBatch; ++u)
      if (bt - u >= 0) { a.sd[((long)(bt - u) * 2 + par) * a.ldo + i] = s; s += v[u]; }
  }
}

}  // namespace rpde
