// Column scans in ONE pass over the data (round 4; one rank -- pencil-sharded runs keep the three kernels of colscan.h,
// their cross-rank summary is an exchange between kernels anyway).
//
// colscan.h reads the input twice: a summary pass from zero inflow, a serial carry kernel, a final pass that loads the
// same rows again.  Here a workgroup of W waves owns 64 columns x W blocks of kColBR rows (a "super-block") and every
// thread keeps its block's rows in registers from the load to the store:
//   1. zero-inflow solve of the block in registers (B2 rows, forward and back substitution: the rows now hold x0);
//      the block-end states go to LDS;
//   2. waves 0 / 1 (one per parity) chain the W blocks' states through LDS into the super-block's aggregate and publish
//      it (7 doubles per column); an arrival counter per (field, column tile) is raised;
//   3. when all NSB super-blocks of the column tile have arrived, the same two waves compose the aggregates with the
//      tabulated super-block transfers into this super-block's inflow states, then chain the W blocks once more with
//      the exact inflow: every block's inflow states land in LDS;
//   4. every thread corrects its rows: x_j = x0_j + s F_j + S0 H0_j + S1 H1_j (+ kappa h_j), F / H0 / H1 = the tabulated
//      responses of row j to its block's forward inflow and backward inflow states (the solve is linear), and stores.
// There is no serial chain ACROSS workgroups: a super-block's aggregate depends on its own rows only, so the NSB
// workgroups of a column tile meet once.  Workgroups take their (field, tile, super-block) from a ticket counter in
// arrival order, so the partners a workgroup waits for have lower tickets or are the next to start: no deadlock as long
// as NSB * nf workgroups fit on the chip at once (they do by a wide margin: NSB <= 16).  The wait is bounded all the same;
// a wait that runs out raises `err` instead of hanging the GPU.
// Arithmetic per element as colscan.h (same references: matvec.rs:207-228, fdma.rs:101-118); results agree with the
// three-kernel form to round-off (the correction form replaces a second run of the recurrences).
#pragma once
#include "colscan.h"

namespace rpde {

constexpr int kCol1Agg = 7;            // doubles per column in a block's / super-block's state: ye yo | e1 e2 | o1 o2 | dot
constexpr int kCol1Inf = 6;            // inflow states of a block: s_even s_odd | S_even[2] | S_odd[2]
constexpr int kCol1Tile = 64;          // columns per workgroup (one wave wide)
constexpr int kCol1MaxW = 16;
constexpr int kCol1MaxNSB = 32;
constexpr int kCol1TabPerBlock = 14;   // m1[2], m2[2][4], g[2][2]

struct ColHh1Tabs {
  const double *F, *H0, *H1;           // per row (padded like the tables of ColHhTabs)
  const double *m1w, *m2w, *gw;        // [NSB][2], [NSB][2][4], [NSB][2][2]: transfers of the super-blocks
};
struct ColHh1Args {
  ColHhArgs a;                         // the operation (one rank: nranks <= 1, row0 = 0)
  ColHh1Tabs x[kColMaxFields];
  int W, NSB, tiles;                   // blocks per workgroup, super-blocks per column, column tiles of 64
  double* agg;                         // [nf][tiles][NSB][7][64] aggregates of the super-blocks
  int* sync;                           // [0]: ticket counter, [1 + f * tiles + tile]: arrivals; zero before the launch
  int* err;                            // raised when a wait ran out
};
RPDE_HD inline long col1_agg(const ColHh1Args& A, int f, int tile, int q) {
  return ((((long)f * A.tiles + tile) * A.NSB) + q) * (kCol1Agg * kCol1Tile);
}

struct ColLoc { double v[kCol1Agg]; };

// zero-inflow solve of block b of column i (colhh_block<false> without the stores): on return r[0 .. BR) hold x0
RPDE_HD inline void colhh1_local(const ColHhArgs& a, int f, int b, int i, double (&r)[kColBR + 4], ColLoc& L) {
  constexpr int BR = kColBR;
  const ColHhTabs& t = a.tab[f];
  const int ci = a.in_half ? (i & 1) * a.in_half + (i >> 1) : i;
  const double* __restrict__ w = a.in[f] + ci;
  const int j0 = b * BR, jr = j0 - a.shift[f];
  const int j1 = (j0 + BR < a.n) ? j0 + BR : a.n;
  const int rmax = (a.nin < a.n + 4) ? a.nin : a.n + 4;
#pragma unroll
  for (int u = 0; u < BR + 4; ++u) r[u] = (jr + u >= 0 && jr + u < rmax) ? w[(long)(jr + u) * a.ld] : 0.0;
  double dot = 0.0;
  if (t.w) {
    const bool tail = b == a.NB - 1;
#pragma unroll
    for (int u = 0; u < BR + 4; ++u)
      if ((u < BR && (jr + u < a.n - a.shift[f] || tail)) || (u >= BR && tail)) dot += t.w[(jr + u > 0) ? jr + u : 0] * r[u];
  }
  double ye = 0.0, yo = 0.0, e1 = 0.0, e2 = 0.0, o1 = 0.0, o2 = 0.0;
#pragma unroll
  for (int u = 0; u < BR; ++u) {
    const int j = j0 + u;
    const double bj = t.t0[j] * r[u] + t.t1[j] * r[u + 2] + t.t2[j] * r[u + 4];
    double& yp = (u & 1) ? yo : ye;
    const double yn = bj + t.q1[j] * yp;
    yp = (j < j1) ? yn : yp;
    r[u] = yn;
  }
#pragma unroll
  for (int u = BR - 1; u >= 0; --u) {
    const int j = j0 + u;
    double& x1 = (u & 1) ? o1 : e1;
    double& x2 = (u & 1) ? o2 : e2;
    const double xj = t.p2[j] * r[u] + t.q2[j] * x1 + t.r2[j] * x2;
    x2 = (j < j1) ? x1 : x2;
    x1 = (j < j1) ? xj : x1;
    r[u] = xj;
  }
  L.v[0] = ye; L.v[1] = yo; L.v[2] = e1; L.v[3] = e2; L.v[4] = o1; L.v[5] = o2; L.v[6] = dot;
}

// rows of block b from x0 and the block's inflow states
RPDE_HD inline void colhh1_final(const ColHhArgs& a, const ColHh1Tabs& x, int f, int b, int i, const double (&r)[kColBR + 4],
                                 const double (&inf)[kCol1Inf], double kap) {
  constexpr int BR = kColBR;
  const ColHhTabs& t = a.tab[f];
  const int j0 = b * BR, j1 = (j0 + BR < a.n) ? j0 + BR : a.n;
  double* __restrict__ out = a.out[f] + i;
  bool bad = false;
#pragma unroll
  for (int u = 0; u < BR; ++u) {
    const int j = j0 + u, p = u & 1;
    if (j < j1) {
      double v = r[u] + inf[p] * x.F[j] + inf[2 + 2 * p] * x.H0[j] + inf[3 + 2 * p] * x.H1[j];
      if (t.w) v += kap * t.h[j];
      out[(long)j * a.ld] = v;
      bad |= (v != v);
    }
  }
  if (bad && a.nanflag) *a.nanflag = 1;
}

// The chains over the W blocks of a super-block, one parity, one column.  `loc` = [W][7][64] zero-inflow states of the
// blocks, `tb` = [W][14] their transfers (identity for blocks past the end), `inf` = [W][6][64] inflow states (out).
// s_in / S0 / S1: the inflow of the super-block; returns the state it hands on (forward end value, backward end state).
RPDE_HD inline void colhh1_chain(const double* loc, const double* tb, double* inf, int W, int par, int lane, double s_in, double S0,
                                 double S1, double& s_out, double& T0, double& T1) {
  double s = s_in;
  for (int w = 0; w < W; ++w) {
    inf[(w * kCol1Inf + par) * kCol1Tile + lane] = s;
    s = tb[w * kCol1TabPerBlock + par] * s + loc[(w * kCol1Agg + par) * kCol1Tile + lane];
  }
  s_out = s;
  double t0 = S0, t1 = S1;
  for (int w = W - 1; w >= 0; --w) {
    inf[(w * kCol1Inf + 2 + 2 * par) * kCol1Tile + lane] = t0;
    inf[(w * kCol1Inf + 3 + 2 * par) * kCol1Tile + lane] = t1;
    const double* m = tb + w * kCol1TabPerBlock + 2 + par * 4;
    const double* g = tb + w * kCol1TabPerBlock + 10 + par * 2;
    const double fi = inf[(w * kCol1Inf + par) * kCol1Tile + lane];
    const double v0 = loc[(w * kCol1Agg + 2 + 2 * par) * kCol1Tile + lane], v1 = loc[(w * kCol1Agg + 3 + 2 * par) * kCol1Tile + lane];
    const double n0 = m[0] * t0 + m[1] * t1 + (v0 + g[0] * fi);
    const double n1 = m[2] * t0 + m[3] * t1 + (v1 + g[1] * fi);
    t0 = n0; t1 = n1;
  }
  T0 = t0; T1 = t1;
}

// inflow of super-block q0 of a column from the aggregates `agg` = [NSB][7][64] of all super-blocks, one parity;
// `sc` = [NSB][64] scratch (the forward inflow of every super-block)
RPDE_HD inline void colhh1_compose(const double* agg, const ColHh1Tabs& x, double* sc, int NSB, int q0, int par, int lane, double& s_in,
                                   double& S0, double& S1) {
  double s = 0.0;
  s_in = 0.0;
  for (int q = 0; q < NSB; ++q) {
    sc[q * kCol1Tile + lane] = s;
    if (q == q0) s_in = s;
    s = x.m1w[q * 2 + par] * s + agg[(q * kCol1Agg + par) * kCol1Tile + lane];
  }
  double t0 = 0.0, t1 = 0.0;
  for (int q = NSB - 1; q > q0; --q) {
    const double* m = x.m2w + (q * 2 + par) * 4;
    const double* g = x.gw + (q * 2 + par) * 2;
    const double fi = sc[q * kCol1Tile + lane];
    const double v0 = agg[(q * kCol1Agg + 2 + 2 * par) * kCol1Tile + lane], v1 = agg[(q * kCol1Agg + 3 + 2 * par) * kCol1Tile + lane];
    const double n0 = m[0] * t0 + m[1] * t1 + (v0 + g[0] * fi);
    const double n1 = m[2] * t0 + m[3] * t1 + (v1 + g[1] * fi);
    t0 = n0; t1 = n1;
  }
  S0 = t0; S1 = t1;
}

// transfers of block b of field f into tb[14] (identity behind the last block)
RPDE_HD inline void colhh1_block_tab(const ColHhTabs& t, int b, int NB, int k, double& v) {
  if (b >= NB) { v = (k < 2 || k == 2 || k == 5 || k == 6 || k == 9) ? 1.0 : 0.0; return; }   // m1 = 1, m2 = I, g = 0
  v = (k < 2) ? t.m1[b * 2 + k] : (k < 10) ? t.m2[b * 8 + (k - 2)] : t.g[b * 4 + (k - 10)];
}

}  // namespace rpde
