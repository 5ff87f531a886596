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
//   3. the workgroup that arrives LAST at a column tile turns the aggregates of the tile's NSB super-blocks into their
//      inflow states (one forward and one backward sweep with the tabulated super-block transfers, colhh1_sweep),
//      publishes them in place of the aggregates and raises the tile's ready flag; the others wait for that flag and
//      read their own six rows -- the exchange is O(NSB) per tile (every workgroup composing for itself from all
//      aggregates was O(NSB^2) traffic and measured 0.42 ms with 16 super-blocks against 0.31 ms with 8 at 4097^2).
//      Waves 0 / 1 then chain the W blocks once more with the exact inflow: every block's inflow states land in LDS;
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

#ifdef RPDE_EMU
using ktab_t = const double*;
#else
// tables in the CONSTANT address space: a wave-uniform index is a scalar load whatever the kernel has stored before (the
// ticket atomic at the top of the kernel would otherwise turn every coefficient into a vector load)
using ktab_t = const __attribute__((address_space(4))) double*;
#endif

constexpr int kCol1Agg = 7;            // doubles per column in a block's / super-block's state: ye yo | e1 e2 | o1 o2 | dot
constexpr int kCol1Inf = 6;            // inflow states of a block: s_even s_odd | S_even[2] | S_odd[2]
constexpr int kCol1Stg = 6;            // doubles per column of a staged aggregate (the first six of kCol1Agg)
constexpr int kCol1Tile = 64;          // columns per workgroup (one wave wide)
constexpr int kCol1MaxW = 16;
constexpr int kCol1MaxNSB = 32;

struct ColHh1Tabs {
  const double *F, *H0, *H1;           // per row (padded like the tables of ColHhTabs)
  const double *m1w, *m2w, *gw;        // [NSB][2], [NSB][2][4], [NSB][2][2]: transfers of the super-blocks
};
struct ColHh1Args {
  ColHhArgs a;                         // the operation (one rank: nranks <= 1, row0 = 0)
  ColHh1Tabs x[kColMaxFields];
  int W, NSB, tiles;                   // blocks per workgroup, super-blocks per column, column tiles of 64
  double* agg;                         // [nf][tiles][NSB][7][64] aggregates of the super-blocks, then (in place) their inflow states;
                                       // slot 6 of super-block 0 becomes the rank-one sum of the column
  int* sync;                           // [0]: ticket counter, [1 + f * tiles + tile]: arrivals; zero before the launch
  int* ready;                          // [f * tiles + tile]: the inflow states of the tile are published; zero before the launch
  int* err;                            // raised when a wait ran out
  long long* trace = nullptr;          // diagnostics (Navier2DEngine::trace_launch): per workgroup, clock values of thread 0 at the phase boundaries
};
RPDE_HD inline long col1_agg(const ColHh1Args& A, int f, int tile, int q) {
  return ((((long)f * A.tiles + tile) * A.NSB) + q) * (kCol1Agg * kCol1Tile);
}

struct ColLoc { double v[kCol1Agg]; };

// zero-inflow solve of block b of column i (colhh_block<false> without the stores): on return r[0 .. BR) hold x0
RPDE_HD inline void colhh1_local(const ColHhArgs& a, int f, int b, int i, double (&r)[kColBR + 4], ColLoc& L) {
  constexpr int BR = kColBR;
  const ColHhTabs& t = a.tab[f];
  ktab_t tw = (ktab_t)t.w, t0 = (ktab_t)t.t0, t1 = (ktab_t)t.t1, t2 = (ktab_t)t.t2, q1 = (ktab_t)t.q1, p2 = (ktab_t)t.p2, q2 = (ktab_t)t.q2,
         r2 = (ktab_t)t.r2;
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
      if ((u < BR && (jr + u < a.n - a.shift[f] || tail)) || (u >= BR && tail)) dot += tw[(jr + u > 0) ? jr + u : 0] * r[u];
  }
  double ye = 0.0, yo = 0.0, e1 = 0.0, e2 = 0.0, o1 = 0.0, o2 = 0.0;
#pragma unroll
  for (int u = 0; u < BR; ++u) {
    const int j = j0 + u;
    const double bj = t0[j] * r[u] + t1[j] * r[u + 2] + t2[j] * r[u + 4];
    double& yp = (u & 1) ? yo : ye;
    const double yn = bj + q1[j] * yp;
    yp = (j < j1) ? yn : yp;
    r[u] = yn;
  }
#pragma unroll
  for (int u = BR - 1; u >= 0; --u) {
    const int j = j0 + u;
    double& x1 = (u & 1) ? o1 : e1;
    double& x2 = (u & 1) ? o2 : e2;
    const double xj = p2[j] * r[u] + q2[j] * x1 + r2[j] * x2;
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
  ktab_t F = (ktab_t)x.F, H0 = (ktab_t)x.H0, H1 = (ktab_t)x.H1, h = (ktab_t)t.h;
  bool bad = false;
#pragma unroll
  for (int u = 0; u < BR; ++u) {
    const int j = j0 + u, p = u & 1;
    if (j < j1) {
      double v = r[u] + inf[p] * F[j] + inf[2 + 2 * p] * H0[j] + inf[3 + 2 * p] * H1[j];
      if (t.w) v += kap * h[j];
      out[(long)j * a.ld] = v;
      bad |= (v != v);
    }
  }
  if (bad && a.nanflag) *a.nanflag = 1;
}

// transfers of block b (identity behind the last block): m1, the 2 x 2 matrix m2, g of one parity
RPDE_HD inline void colhh1_block_tab(const ColHhTabs& t, int b, int NB, int par, double& m1, double (&m2)[4], double (&g)[2]) {
  if (b >= NB) { m1 = 1.0; m2[0] = 1.0; m2[1] = 0.0; m2[2] = 0.0; m2[3] = 1.0; g[0] = 0.0; g[1] = 0.0; return; }
  m1 = ((ktab_t)t.m1)[b * 2 + par];
#pragma unroll
  for (int k = 0; k < 4; ++k) m2[k] = ((ktab_t)t.m2)[b * 8 + par * 4 + k];
  g[0] = ((ktab_t)t.g)[b * 4 + par * 2]; g[1] = ((ktab_t)t.g)[b * 4 + par * 2 + 1];
}

// The chains over the W blocks b0 .. b0 + W - 1 of a super-block, one parity, one column, IN PLACE in `loc` = [W][7][64]
// (slot par: forward end value of the block from zero inflow, slots 2 + 2 par, 3 + 2 par: its backward end state).
//   FIRST = true   from zero inflow of the super-block: slot par becomes the block's forward inflow in that run (f1);
//                  returns the super-block's aggregate (forward end value, backward end state);
//   FIRST = false  with the super-block's exact inflow (s_in | S0, S1): slot par becomes the exact forward inflow
//                  f1 + P s_in (P = product of the transfer factors in front of the block: the chain is linear), slots
//                  2 + 2 par, 3 + 2 par the exact backward inflow states.
template <bool FIRST>
RPDE_HD inline void colhh1_chain(double* loc, const ColHhTabs& t, int b0, int NB, int W, int par, int lane, double s_in, double S0,
                                 double S1, double& s_out, double& T0, double& T1) {
  double s = FIRST ? 0.0 : 1.0;            // FIRST: running value, else: running product P
#pragma unroll 4
  for (int w = 0; w < W; ++w) {
    double* L = loc + (long)w * kCol1Agg * kCol1Tile + lane;
    const double m1 = (b0 + w < NB) ? ((ktab_t)t.m1)[(b0 + w) * 2 + par] : 1.0;
    if (FIRST) {
      const double a = L[par * kCol1Tile];
      L[par * kCol1Tile] = s;
      s = m1 * s + a;
    } else {
      L[par * kCol1Tile] += s * s_in;
      s *= m1;
    }
  }
  s_out = s;
  double t0 = S0, t1 = S1;
#pragma unroll 4
  for (int w = W - 1; w >= 0; --w) {
    double* L = loc + (long)w * kCol1Agg * kCol1Tile + lane;
    double m1, m[4], g[2];
    colhh1_block_tab(t, b0 + w, NB, par, m1, m, g);
    const double fi = L[par * kCol1Tile];
    const double v0 = L[(2 + 2 * par) * kCol1Tile], v1 = L[(3 + 2 * par) * kCol1Tile];
    if (!FIRST) { L[(2 + 2 * par) * kCol1Tile] = t0; L[(3 + 2 * par) * kCol1Tile] = t1; }
    const double n0 = m[0] * t0 + m[1] * t1 + (v0 + g[0] * fi);
    const double n1 = m[2] * t0 + m[3] * t1 + (v1 + g[1] * fi);
    t0 = n0; t1 = n1;
  }
  T0 = t0; T1 = t1;
}

// inflow states of ALL super-blocks of a column from their aggregates, one parity, IN PLACE in `stg` = [NSB][6][64]
// (slot par: forward end value -> forward inflow; slots 2 + 2 par, 3 + 2 par: backward end state -> backward inflow states)
RPDE_HD inline void colhh1_sweep(double* stg, const ColHh1Tabs& x, int NSB, int par, int lane) {
  double s = 0.0;
#pragma unroll 4
  for (int q = 0; q < NSB; ++q) {
    double* G = stg + (long)q * kCol1Stg * kCol1Tile + lane;
    const double a = G[par * kCol1Tile];
    G[par * kCol1Tile] = s;
    s = ((ktab_t)x.m1w)[q * 2 + par] * s + a;
  }
  double t0 = 0.0, t1 = 0.0;
#pragma unroll 4
  for (int q = NSB - 1; q >= 0; --q) {
    double* G = stg + (long)q * kCol1Stg * kCol1Tile + lane;
    ktab_t m = (ktab_t)x.m2w + (q * 2 + par) * 4;
    ktab_t g = (ktab_t)x.gw + (q * 2 + par) * 2;
    const double fi = G[par * kCol1Tile];
    const double v0 = G[(2 + 2 * par) * kCol1Tile], v1 = G[(3 + 2 * par) * kCol1Tile];
    G[(2 + 2 * par) * kCol1Tile] = t0; G[(3 + 2 * par) * kCol1Tile] = t1;
    const double n0 = m[0] * t0 + m[1] * t1 + (v0 + g[0] * fi);
    const double n1 = m[2] * t0 + m[3] * t1 + (v1 + g[1] * fi);
    t0 = n0; t1 = n1;
  }
}

// ints of the synchronisation area for `tiles` column tiles and up to kColMaxFields fields:
// [0] ticket counter | arrivals per (field, tile) | ready flags per (field, tile) | [col1_err_index] error flag
RPDE_HD inline int col1_err_index(int tiles) { return 1 + 2 * kColMaxFields * tiles; }
RPDE_HD inline size_t col1_sync_ints(int tiles) { return (size_t)col1_err_index(tiles) + 1; }

// dynamic LDS of the kernel (doubles): the blocks' states, the staged aggregates of the tile, the rank-one sums
RPDE_HD inline size_t col1_lds_doubles(int W, int NSB) {
  return (size_t)W * kCol1Agg * kCol1Tile + (size_t)NSB * kCol1Stg * kCol1Tile + kCol1Tile;
}

}  // namespace rpde
