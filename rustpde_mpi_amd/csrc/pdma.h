// Kernels of the three-term Chebyshev stencil (`cheb_dirichlet_neumann`, the y base of the temperature with the
// "hc" boundary condition: navier.rs:245-248 / 366-369).  Its stencil T_k + a_k T_{k+1} + b_k T_{k+2} couples even and
// odd coefficients, so none of the stride-2 machinery (line-VM scans, column scans, parity blocks) applies; the
// Helmholtz matrix along that axis has seven diagonals and is solved like `PdmaPlus2` (src/solver/pdma_plus2.rs:
// 119-157: a two-term forward and a four-term backward recurrence with a precomputed factorisation).
//
//  * sten3_rows  : composite -> orthonormal along y on a YX array (row = y index): three row taps, coalesced
//  * pdma_cols   : B2 rows (matvec.rs:207-228) + PdmaPlus2 solve along y on a YX array: one thread per column walks
//                  the rows in batches of 16, the next batch in flight while the current one is computed; the row
//                  coefficients are wave-uniform (scalar loads).  4097 columns are 65 waves -- this is the simple
//                  form of the operation (about one CU in four busy), good for a path that is not the benchmark's
//  * sten3_lines / pdma_lines : the same two operations along CONTIGUOUS lines (canonical XY arrays) for the
//                  generic operators (`Space2` to_ortho / from_ortho / forward / backward, `HholtzAdi::solve`) --
//                  setup, initial conditions, snapshots, diagnostics; one thread per line
// One source for both builds: the per-column / per-line bodies are plain functions; the HIP build wraps them in
// kernels, the emulation build in loops.
#pragma once
#include "platform.h"

namespace rpde {

struct PdmaTabs { const double *l2, *ka, *imu, *al, *be, *ga, *de; int n; };   // device tables, each n + 4 long (hostmath.h PdmaTables)

struct Sten3RowsArgs {
  const double* in; long ldi;    // m rows (composite)
  double* out; long ldo;         // m + 2 rows (orthonormal); may not alias `in`
  int m, ncols;
  const double *low1, *low2;     // S[k+1,k], S[k+2,k]
  int row0, nrows;               // rows [row0, row0 + nrows) of the output are produced
};

// Blocked form of pdma_cols (round 5): the two recurrences are linear with column-independent coefficients, so a block of
// kPdmaBR rows solves from ZERO inflow and is corrected afterwards with its exact inflow times tabulated homogeneous
// solutions -- z_j = zp_j + a Phi1_j + b Phi2_j (two inflow values, forward), x_i = xp_i + sum_k c_k Psi_k,i (four, backward).
// The inflows of all blocks come from a serial pass over the blocks' end states with tabulated block transfer matrices
// (2 x 2 forward, 4 x 4 backward; one thread per column, NB steps).  Five launches, every one with (blocks x column tiles)
// workgroups or one thread per column of NB steps: 8320 workgroups at 4097 x 4095 instead of 65 waves.
constexpr int kPdmaBR = 32;
struct PdmaBlkTabs {
  const double *phi1 = nullptr, *phi2 = nullptr;   // n each: forward homogeneous solutions, restarted at every block
  const double* fm = nullptr;                      // NB x 4: forward transfer (z_last, z_last-1) <- (a, b), row-major 2 x 2
  const double *psi1 = nullptr, *psi2 = nullptr, *psi3 = nullptr, *psi4 = nullptr;   // n each: backward homogeneous solutions
  const double* bm = nullptr;                      // NB x 16: backward transfer (x_j0 .. x_j0+3) <- (c1 .. c4), row-major 4 x 4
  int NB = 0;
};
struct PdmaColsArgs {
  const double* in; long ldi;    // n rows (with B2: the first n of the n + 2 orthonormal rows)
  double* out; long ldo;         // n rows
  int n, ncols;
  const double *t0, *t1, *t2;    // B2 rows (taps j, j + 2, j + 4), or null: the input is the right-hand side itself
  PdmaTabs f;
  int* nanflag;                  // raised when a NaN is stored (Integrate::exit on the device), may be null
  PdmaBlkTabs blk{};             // blk.phi1 != null: the blocked form
  double* ws = nullptr;          // workspace of the blocked form: 12 NB rows of ldw doubles (end states 2 + 4, inflows 2 + 4 per block)
  long ldw = 0;
};
RPDE_HD inline size_t pdma_blk_ws_doubles(int n, long ldw) { return (size_t)12 * ((n + kPdmaBR - 1) / kPdmaBR) * (size_t)ldw; }

struct Sten3LinesArgs {
  const double* in; long ldi; double* out; long ldo;
  int nlines, m, es, ncomp;      // element k of component c of a line: p[line * ld + k * es + c]
  const double *low1, *low2;
};

struct PdmaLinesArgs {
  const double* in; long ldi; double* out; long ldo;
  int nlines, n, es, ncomp;
  const double *low1, *low2;     // non-null: rhs_k = c_k + low1_k c_{k+1} + low2_k c_{k+2} (S^T of from_ortho); null: rhs = in
  PdmaTabs f;
  // pencil-sharded "hc" step (lines = the y-lines of an x-pencil): B2 rows in front of the solve (taps j, j + 2, j + 4 of
  // the n input values; matvec.rs:207-228) and the NaN flag of Integrate::exit.  All null by default.
  const double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;
  int* nanflag = nullptr;
};

// ------------------------------------------------------------------------------------------------------------------
RPDE_HD inline void sten3_rows_point(const Sten3RowsArgs& a, int j, int c) {
  double x = 0.0;
  if (j < a.m) x = a.in[(long)j * a.ldi + c];
  if (j >= 1 && j - 1 < a.m) x += a.low1[j - 1] * a.in[(long)(j - 1) * a.ldi + c];
  if (j >= 2 && j - 2 < a.m) x += a.low2[j - 2] * a.in[(long)(j - 2) * a.ldi + c];
  a.out[(long)j * a.ldo + c] = x;
}

constexpr int kPdmaBatch = 16;

// one column: forward (B2 rows + two-term recurrence, ze stored to `out`), then the four-term backward recurrence in place
RPDE_HD inline bool pdma_column(const PdmaColsArgs& a, int c) {
  constexpr int B = kPdmaBatch;
  const int n = a.n;
  const int nin = n;   // the B2 rows never read the last two orthonormal coefficients (their table entries are zero, matvec.rs:215-226): not loaded
  const double* in = a.in + c;
  double* out = a.out + c;
  double z1 = 0.0, z2 = 0.0;
  // forward: rows [j0, j0 + B) need the input rows [j0, j0 + B + 4)
  double cur[B + 4], nxt[B];
#pragma unroll
  for (int q = 0; q < B + 4; ++q) cur[q] = q < nin ? in[(long)q * a.ldi] : 0.0;
  for (int j0 = 0; j0 < n; j0 += B) {
#pragma unroll
    for (int q = 0; q < B; ++q) { const int r = j0 + B + 4 + q; nxt[q] = r < nin ? in[(long)r * a.ldi] : 0.0; }
#pragma unroll
    for (int q = 0; q < B; ++q) {
      const int j = j0 + q;
      if (j < n) {
        double r = cur[q];
        if (a.t0) {
          r = a.t0[j] * cur[q] + a.t1[j] * cur[q + 2];
          r += a.t2[j] * cur[q + 4];
        }
        const double z = (r - a.f.l2[j] * z2 - a.f.ka[j] * z1) * a.f.imu[j];
        out[(long)j * a.ldo] = z;
        z2 = z1; z1 = z;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = cur[B + q];
#pragma unroll
    for (int q = 0; q < B; ++q) cur[4 + q] = nxt[q];
  }
  // backward: x_i = ze_i - al_i x_{i+1} - be_i x_{i+2} - ga_i x_{i+3} - de_i x_{i+4}; the tables are zero past the end
  double x1 = 0.0, x2 = 0.0, x3 = 0.0, x4 = 0.0;
  bool bad = false;
  const int nb = (n + B - 1) / B;
  double zc[B], zn[B];
#pragma unroll
  for (int q = 0; q < B; ++q) { const int r = (nb - 1) * B + q; zc[q] = r < n ? out[(long)r * a.ldo] : 0.0; }
  for (int b = nb - 1; b >= 0; --b) {
#pragma unroll
    for (int q = 0; q < B; ++q) { const int r = (b - 1) * B + q; zn[q] = (b > 0) ? out[(long)r * a.ldo] : 0.0; }
#pragma unroll
    for (int q = B - 1; q >= 0; --q) {
      const int i = b * B + q;
      if (i < n) {
        const double x = zc[q] - a.f.al[i] * x1 - a.f.be[i] * x2 - a.f.ga[i] * x3 - a.f.de[i] * x4;
        out[(long)i * a.ldo] = x;
        bad |= (x != x);
        x4 = x3; x3 = x2; x2 = x1; x1 = x;
      }
    }
#pragma unroll
    for (int q = 0; q < B; ++q) zc[q] = zn[q];
  }
  return bad;
}

// ---- blocked form: block b = rows [b BR, min((b + 1) BR, n)), column c.  Workspace rows (each ldw doubles):
//   S1 = ws + (2 b + k) ldw            forward end state k = 0: z_last, 1: z_last-1       (NB x 2 rows)
//   I1 = S1 + 2 NB ldw                 forward inflow of block b: (z_{j0-1}, z_{j0-2})
//   S2 = I1 + 2 NB ldw                 backward end state k = 0 .. 3: x_{j0 + k}          (NB x 4 rows)
//   I2 = S2 + 4 NB ldw                 backward inflow of block b: x_{j1 + k}
RPDE_HD inline double* pdma_ws_s1(const PdmaColsArgs& a) { return a.ws; }
RPDE_HD inline double* pdma_ws_i1(const PdmaColsArgs& a) { return a.ws + (size_t)2 * a.blk.NB * a.ldw; }
RPDE_HD inline double* pdma_ws_s2(const PdmaColsArgs& a) { return a.ws + (size_t)4 * a.blk.NB * a.ldw; }
RPDE_HD inline double* pdma_ws_i2(const PdmaColsArgs& a) { return a.ws + (size_t)8 * a.blk.NB * a.ldw; }

// forward from zero inflow: B2 rows + two-term recurrence; zp -> out, end state -> S1
RPDE_HD inline void pdma_blk_fwd_local(const PdmaColsArgs& a, int b, int c) {
  constexpr int BR = kPdmaBR;
  const int n = a.n, j0 = b * BR;
  const double* in = a.in + c;
  double* out = a.out + c;
  double cur[BR + 4];
#pragma unroll
  for (int q = 0; q < BR + 4; ++q) cur[q] = (j0 + q < n) ? in[(long)(j0 + q) * a.ldi] : 0.0;
  double z1 = 0.0, z2 = 0.0;
#pragma unroll
  for (int q = 0; q < BR; ++q) {
    const int j = j0 + q;
    if (j < n) {
      double r = cur[q];
      if (a.t0) {
        r = a.t0[j] * cur[q] + a.t1[j] * cur[q + 2];
        r += a.t2[j] * cur[q + 4];
      }
      const double z = (r - a.f.l2[j] * z2 - a.f.ka[j] * z1) * a.f.imu[j];
      out[(long)j * a.ldo] = z;
      z2 = z1; z1 = z;
    }
  }
  double* s1 = pdma_ws_s1(a) + (size_t)2 * b * a.ldw + c;
  s1[0] = z1; s1[a.ldw] = z2;
}
// inflows of every block of one column (forward): a serial pass over the end states
RPDE_HD inline void pdma_blk_fwd_carry(const PdmaColsArgs& a, int c) {
  double a1 = 0.0, a2 = 0.0;
  const double* s1 = pdma_ws_s1(a) + c;
  double* i1 = pdma_ws_i1(a) + c;
  for (int b = 0; b < a.blk.NB; ++b) {
    i1[(size_t)(2 * b) * a.ldw] = a1; i1[(size_t)(2 * b + 1) * a.ldw] = a2;
    const double* m = a.blk.fm + 4 * b;
    const double e1 = s1[(size_t)(2 * b) * a.ldw], e2 = s1[(size_t)(2 * b + 1) * a.ldw];
    const double n1 = e1 + m[0] * a1 + m[1] * a2, n2 = e2 + m[2] * a1 + m[3] * a2;
    a1 = n1; a2 = n2;
  }
}
// exact forward values of the block (in registers), then the backward recurrence from zero inflow; xp -> out, end state -> S2
RPDE_HD inline void pdma_blk_mid(const PdmaColsArgs& a, int b, int c) {
  constexpr int BR = kPdmaBR;
  const int n = a.n, j0 = b * BR;
  double* out = a.out + c;
  const double* i1 = pdma_ws_i1(a) + (size_t)2 * b * a.ldw + c;
  const double a1 = i1[0], a2 = i1[a.ldw];
  double z[BR];
#pragma unroll
  for (int q = 0; q < BR; ++q) {
    const int j = j0 + q;
    z[q] = (j < n) ? out[(long)j * a.ldo] + a1 * a.blk.phi1[j] + a2 * a.blk.phi2[j] : 0.0;
  }
  double x1 = 0.0, x2 = 0.0, x3 = 0.0, x4 = 0.0;
#pragma unroll
  for (int q = BR - 1; q >= 0; --q) {
    const int i = j0 + q;
    if (i < n) {
      const double x = z[q] - a.f.al[i] * x1 - a.f.be[i] * x2 - a.f.ga[i] * x3 - a.f.de[i] * x4;
      out[(long)i * a.ldo] = x;
      x4 = x3; x3 = x2; x2 = x1; x1 = x;
    }
  }
  double* s2 = pdma_ws_s2(a) + (size_t)4 * b * a.ldw + c;
  s2[0] = x1; s2[a.ldw] = x2; s2[2 * a.ldw] = x3; s2[3 * a.ldw] = x4;
}
RPDE_HD inline void pdma_blk_bwd_carry(const PdmaColsArgs& a, int c) {
  double cc[4] = {0.0, 0.0, 0.0, 0.0};
  const double* s2 = pdma_ws_s2(a) + c;
  double* i2 = pdma_ws_i2(a) + c;
  for (int b = a.blk.NB - 1; b >= 0; --b) {
    for (int k = 0; k < 4; ++k) i2[(size_t)(4 * b + k) * a.ldw] = cc[k];
    const double* m = a.blk.bm + 16 * b;
    double nn[4];
    for (int k = 0; k < 4; ++k) {
      double v = s2[(size_t)(4 * b + k) * a.ldw];
      for (int q = 0; q < 4; ++q) v += m[4 * k + q] * cc[q];
      nn[k] = v;
    }
    for (int k = 0; k < 4; ++k) cc[k] = nn[k];
  }
}
RPDE_HD inline bool pdma_blk_final(const PdmaColsArgs& a, int b, int c) {
  constexpr int BR = kPdmaBR;
  const int n = a.n, j0 = b * BR;
  double* out = a.out + c;
  const double* i2 = pdma_ws_i2(a) + (size_t)4 * b * a.ldw + c;
  const double c1 = i2[0], c2 = i2[a.ldw], c3 = i2[2 * a.ldw], c4 = i2[3 * a.ldw];
  bool bad = false;
  double x[BR];
#pragma unroll
  for (int q = 0; q < BR; ++q) { const int i = j0 + q; x[q] = (i < n) ? out[(long)i * a.ldo] : 0.0; }
#pragma unroll
  for (int q = 0; q < BR; ++q) {
    const int i = j0 + q;
    if (i < n) {
      const double v = x[q] + c1 * a.blk.psi1[i] + c2 * a.blk.psi2[i] + c3 * a.blk.psi3[i] + c4 * a.blk.psi4[i];
      out[(long)i * a.ldo] = v;
      bad |= (v != v);
    }
  }
  return bad;
}

RPDE_HD inline void sten3_line(const Sten3LinesArgs& a, int line, int comp) {
  const double* in = a.in + (long)line * a.ldi + comp;
  double* out = a.out + (long)line * a.ldo + comp;
  const int m = a.m, es = a.es;
  double p1 = 0.0, p2 = 0.0;   // in[k-1], in[k-2]
  for (int k = 0; k < m + 2; ++k) {
    const double v = k < m ? in[(long)k * es] : 0.0;
    double x = v;
    if (k >= 1 && k - 1 < m) x += a.low1[k - 1] * p1;
    if (k >= 2) x += a.low2[k - 2] * p2;
    out[(long)k * es] = x;
    p2 = p1; p1 = v;
  }
}

RPDE_HD inline void pdma_line(const PdmaLinesArgs& a, int line, int comp) {
  const double* in = a.in + (long)line * a.ldi + comp;
  double* out = a.out + (long)line * a.ldo + comp;
  const int n = a.n, es = a.es;
  double z1 = 0.0, z2 = 0.0;
  for (int j = 0; j < n; ++j) {
    double r = in[(long)j * es];
    if (a.low1) r += a.low1[j] * in[(long)(j + 1) * es] + a.low2[j] * in[(long)(j + 2) * es];   // S^T c: the input has n + 2 entries
    if (a.t0) {                                                                              // B2 row: the input has n entries
      r *= a.t0[j];
      if (j + 2 < n) r += a.t1[j] * in[(long)(j + 2) * es];
      if (j + 4 < n) r += a.t2[j] * in[(long)(j + 4) * es];
    }
    const double z = (r - a.f.l2[j] * z2 - a.f.ka[j] * z1) * a.f.imu[j];
    out[(long)j * es] = z;
    z2 = z1; z1 = z;
  }
  double x1 = 0.0, x2 = 0.0, x3 = 0.0, x4 = 0.0;
  bool bad = false;
  for (int i = n - 1; i >= 0; --i) {
    const double x = out[(long)i * es] - a.f.al[i] * x1 - a.f.be[i] * x2 - a.f.ga[i] * x3 - a.f.de[i] * x4;
    out[(long)i * es] = x;
    bad |= (x != x);
    x4 = x3; x3 = x2; x2 = x1; x1 = x;
  }
  if (bad && a.nanflag) *a.nanflag = 1;
}

}  // namespace rpde
