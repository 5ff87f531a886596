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
  // Synchronisation area of THIS launch site, never reset: counters only grow, a launch is an epoch.  Workgroup number t
  // (over all launches of the site) belongs to epoch t / gridDim.x + 1 and is ticket t mod gridDim.x of its launch.
  unsigned long long* sync;            // [0]: ticket counter; [1 + f * tiles + tile]: arrivals (NSB per epoch);
                                       // [1 + kColMaxFields * tiles + f * tiles + tile]: epoch whose inflow states are published
  int* err;                            // raised when a wait ran out
  long long* trace = nullptr;          // diagnostics (Navier2DEngine::trace_launch): per workgroup, clock values of thread 0 at the phase boundaries
};
RPDE_HD inline long col1_agg(const ColHh1Args& A, int f, int tile, int q) {
  return ((((long)f * A.tiles + tile) * A.NSB) + q) * (kCol1Agg * kCol1Tile);
}

struct ColLoc { double v[kCol1Agg]; };

constexpr int kCol1TabPerBlock = 14;   // m1[2], m2[2][4], g[2][2] of a block / a super-block

// Where a thread takes the coefficients of its block's rows from.  Col1TabDirect: the tables themselves (host emulation).
// Col1TabLanes (device): every coefficient is wave-uniform, but a scalar load per row and table is a chain of cold misses
// of the scalar cache (measured: 18 us of a workgroup's 52 us went into the zero-inflow solve, 13 us into the first chain)
// -- and all waves of a workgroup wait at the same time.  Instead the lanes of a wave fetch the coefficients of the block's
// 32 rows with ONE vector load per pair of tables, issued together with the row loads; row u's coefficient is read back
// with v_readlane into a scalar register pair.
struct Col1TabDirect {
  const double *t0_, *t1_, *t2_, *q1_, *p2_, *q2_, *r2_, *w_, *F_, *H0_, *H1_, *h_;
  int j0, jr;
  Col1TabDirect(const ColHhTabs& t, const ColHh1Tabs& x, int j0_, int jr_)
      : t0_(t.t0), t1_(t.t1), t2_(t.t2), q1_(t.q1), p2_(t.p2), q2_(t.q2), r2_(t.r2), w_(t.w), F_(x.F), H0_(x.H0), H1_(x.H1), h_(t.h), j0(j0_), jr(jr_) {}
  double t0(int u) const { return t0_[j0 + u]; }
  double t1(int u) const { return t1_[j0 + u]; }
  double t2(int u) const { return t2_[j0 + u]; }
  double q1(int u) const { return q1_[j0 + u]; }
  double p2(int u) const { return p2_[j0 + u]; }
  double q2(int u) const { return q2_[j0 + u]; }
  double r2(int u) const { return r2_[j0 + u]; }
  double w(int u) const { return w_[(jr + u > 0) ? jr + u : 0]; }
  double F(int u) const { return F_[j0 + u]; }
  double H0(int u) const { return H0_[j0 + u]; }
  double H1(int u) const { return H1_[j0 + u]; }
  double h(int u) const { return h_[j0 + u]; }
};
#ifndef RPDE_EMU
__device__ __forceinline__ double col1_lane(double v, int l) { return rpde_lane_f64(v, l); }   // the value lane l holds, wave-uniform
struct Col1TabLanes {
  double a, b, c, d, e, g, k;          // lanes 0 .. 31 | 32 .. 63:  t0 | t1,  t2 | q1,  p2 | q2,  r2 | -,  w (36 lanes),  F | H0,  H1 | h
  __device__ __forceinline__ Col1TabLanes(const ColHhTabs& t, const ColHh1Tabs& x, int j0, int jr, int lane) {
    using gt = const __attribute__((address_space(1))) double*;
    const int l = lane & 31, j = j0 + l;
    const bool up = lane >= 32;
    // the table pointers as VALUES in scalar registers first: a select between two fields of the argument block would
    // otherwise become a per-lane load of the POINTER (one more memory round trip in front of every table load)
    auto sp = [](const double* p) { asm volatile("" : "+s"(p)); return p; };
    const double *pt0 = sp(t.t0), *pt1 = sp(t.t1), *pt2 = sp(t.t2), *pq1 = sp(t.q1), *pp2 = sp(t.p2), *pq2 = sp(t.q2), *pr2 = sp(t.r2),
                 *pw = sp(t.w), *ph = sp(t.h), *pF = sp(x.F), *pH0 = sp(x.H0), *pH1 = sp(x.H1);
    a = ((gt)(up ? pt1 : pt0))[j];
    b = ((gt)(up ? pq1 : pt2))[j];
    c = ((gt)(up ? pq2 : pp2))[j];
    d = ((gt)pr2)[j];
    g = ((gt)(up ? pH0 : pF))[j];
    k = ((gt)((up && pw) ? ph : pH1))[j];
    e = (pw && lane < kColBR + 4) ? ((gt)pw)[(jr + lane > 0) ? jr + lane : 0] : 0.0;
  }
  __device__ __forceinline__ double t0(int u) const { return col1_lane(a, u); }
  __device__ __forceinline__ double t1(int u) const { return col1_lane(a, 32 + u); }
  __device__ __forceinline__ double t2(int u) const { return col1_lane(b, u); }
  __device__ __forceinline__ double q1(int u) const { return col1_lane(b, 32 + u); }
  __device__ __forceinline__ double p2(int u) const { return col1_lane(c, u); }
  __device__ __forceinline__ double q2(int u) const { return col1_lane(c, 32 + u); }
  __device__ __forceinline__ double r2(int u) const { return col1_lane(d, u); }
  __device__ __forceinline__ double w(int u) const { return col1_lane(e, u); }
  __device__ __forceinline__ double F(int u) const { return col1_lane(g, u); }
  __device__ __forceinline__ double H0(int u) const { return col1_lane(g, 32 + u); }
  __device__ __forceinline__ double H1(int u) const { return col1_lane(k, u); }
  __device__ __forceinline__ double h(int u) const { return col1_lane(k, 32 + u); }
};
#endif

// zero-inflow solve of block b of column i (colhh_block<false> without the stores): on return r[0 .. BR) hold x0
template <class Tab>
RPDE_HD inline void colhh1_local(const ColHhArgs& a, int f, int b, int i, double (&r)[kColBR + 4], ColLoc& L, const Tab& tb) {
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
      if ((u < BR && (jr + u < a.n - a.shift[f] || tail)) || (u >= BR && tail)) dot += tb.w(u) * r[u];
  }
  double ye = 0.0, yo = 0.0, e1 = 0.0, e2 = 0.0, o1 = 0.0, o2 = 0.0;
#pragma unroll
  for (int u = 0; u < BR; ++u) {
    const int j = j0 + u;
    const double bj = tb.t0(u) * r[u] + tb.t1(u) * r[u + 2] + tb.t2(u) * r[u + 4];
    double& yp = (u & 1) ? yo : ye;
    const double yn = bj + tb.q1(u) * yp;
    yp = (j < j1) ? yn : yp;
    r[u] = yn;
  }
#pragma unroll
  for (int u = BR - 1; u >= 0; --u) {
    const int j = j0 + u;
    double& x1 = (u & 1) ? o1 : e1;
    double& x2 = (u & 1) ? o2 : e2;
    const double xj = tb.p2(u) * r[u] + tb.q2(u) * x1 + tb.r2(u) * x2;
    x2 = (j < j1) ? x1 : x2;
    x1 = (j < j1) ? xj : x1;
    r[u] = xj;
  }
  L.v[0] = ye; L.v[1] = yo; L.v[2] = e1; L.v[3] = e2; L.v[4] = o1; L.v[5] = o2; L.v[6] = dot;
}

// rows of block b from x0 and the block's inflow states (`store`: the column exists)
template <class Tab>
RPDE_HD inline void colhh1_final(const ColHhArgs& a, int f, int b, int i, const double (&r)[kColBR + 4], const double (&inf)[kCol1Inf],
                                 double kap, const Tab& tb, bool store) {
  constexpr int BR = kColBR;
  const ColHhTabs& t = a.tab[f];
  const int j0 = b * BR, j1 = (j0 + BR < a.n) ? j0 + BR : a.n;
  double* __restrict__ out = a.out[f] + i;
  bool bad = false;
#pragma unroll
  for (int u = 0; u < BR; ++u) {
    const int j = j0 + u, p = u & 1;
    if (j < j1) {
      double v = r[u] + inf[p] * tb.F(u) + inf[2 + 2 * p] * tb.H0(u) + inf[3 + 2 * p] * tb.H1(u);
      if (t.w) v += kap * tb.h(u);
      if (store) out[(long)j * a.ld] = v;
      bad |= store && (v != v);
    }
  }
  if (bad && a.nanflag) *a.nanflag = 1;
}

// entry k of tb[14] = m1[2], m2[2][4], g[2][2] of block b (nullptr behind the last block: the identity, colhh1_ident_tab)
RPDE_HD inline const double* colhh1_block_tab(const ColHhTabs& t, int b, int NB, int k) {
  if (b >= NB) return nullptr;
  return (k < 2) ? t.m1 + (b * 2 + k) : (k < 10) ? t.m2 + (b * 8 + (k - 2)) : t.g + (b * 4 + (k - 10));
}
RPDE_HD inline double colhh1_ident_tab(int k) { return (k < 2 || k == 2 || k == 5 || k == 6 || k == 9) ? 1.0 : 0.0; }   // m1 = 1, m2 = I, g = 0
RPDE_HD inline const double* colhh1_super_tab(const ColHh1Tabs& x, int q, int k) {
  return (k < 2) ? x.m1w + (q * 2 + k) : (k < 10) ? x.m2w + (q * 8 + (k - 2)) : x.gw + (q * 4 + (k - 10));
}

// The chains over the W blocks b0 .. b0 + W - 1 of a super-block, one parity, one column, IN PLACE in `loc` = [W][7][64]
// (slot par: forward end value of the block from zero inflow, slots 2 + 2 par, 3 + 2 par: its backward end state).
//   FIRST = true   from zero inflow of the super-block: slot par becomes the block's forward inflow in that run (f1);
//                  returns the super-block's aggregate (forward end value, backward end state);
//   FIRST = false  with the super-block's exact inflow (s_in | S0, S1): slot par becomes the exact forward inflow
//                  f1 + P s_in (P = product of the transfer factors in front of the block: the chain is linear), slots
//                  2 + 2 par, 3 + 2 par the exact backward inflow states.
// `tb` = [W][14] the blocks' transfers (colhh1_block_tab; LDS on the device).
template <bool FIRST>
RPDE_HD inline void colhh1_chain(double* loc, const double* tb, int W, int par, int lane, double s_in, double S0, double S1, double& s_out,
                                 double& T0, double& T1) {
  double s = FIRST ? 0.0 : 1.0;            // FIRST: running value, else: running product P
#pragma unroll 4
  for (int w = 0; w < W; ++w) {
    double* L = loc + (long)w * kCol1Agg * kCol1Tile + lane;
    const double m1 = tb[w * kCol1TabPerBlock + par];
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
    const double* m = tb + w * kCol1TabPerBlock + 2 + par * 4;
    const double* g = tb + w * kCol1TabPerBlock + 10 + par * 2;
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
// `tw` = [NSB][14] the super-blocks' transfers (colhh1_super_tab; LDS on the device).
RPDE_HD inline void colhh1_sweep(double* stg, const double* tw, int NSB, int par, int lane) {
  double s = 0.0;
#pragma unroll 4
  for (int q = 0; q < NSB; ++q) {
    double* G = stg + (long)q * kCol1Stg * kCol1Tile + lane;
    const double a = G[par * kCol1Tile];
    G[par * kCol1Tile] = s;
    s = tw[q * kCol1TabPerBlock + par] * s + a;
  }
  double t0 = 0.0, t1 = 0.0;
#pragma unroll 4
  for (int q = NSB - 1; q >= 0; --q) {
    double* G = stg + (long)q * kCol1Stg * kCol1Tile + lane;
    const double* m = tw + q * kCol1TabPerBlock + 2 + par * 4;
    const double* g = tw + q * kCol1TabPerBlock + 10 + par * 2;
    const double fi = G[par * kCol1Tile];
    const double v0 = G[(2 + 2 * par) * kCol1Tile], v1 = G[(3 + 2 * par) * kCol1Tile];
    G[(2 + 2 * par) * kCol1Tile] = t0; G[(3 + 2 * par) * kCol1Tile] = t1;
    const double n0 = m[0] * t0 + m[1] * t1 + (v0 + g[0] * fi);
    const double n1 = m[2] * t0 + m[3] * t1 + (v1 + g[1] * fi);
    t0 = n0; t1 = n1;
  }
}

// 64-bit words of the synchronisation area of a launch site with `tiles` column tiles and up to kColMaxFields fields
RPDE_HD inline size_t col1_sync_words(int tiles) { return (size_t)1 + 2 * (size_t)kColMaxFields * tiles; }

// dynamic LDS of the kernel (doubles): the blocks' states, the staged aggregates of the tile, the rank-one sums
RPDE_HD inline size_t col1_lds_doubles(int W, int NSB) {
  return (size_t)W * kCol1Agg * kCol1Tile + (size_t)NSB * kCol1Stg * kCol1Tile + kCol1Tile + (size_t)(W + NSB) * kCol1TabPerBlock;
}

// ---------------------------------------------------------------------------------------------
// Chebyshev y-derivative of a YX array (coldiff_pass / coldiff_carry of colscan.h) in ONE pass, one rank: a workgroup of
// kDiff1W waves owns 64 columns x kDiff1W blocks of kDiff1BR rows; a thread keeps the suffix sums of its rows in registers,
// the block sums go through LDS, the super-block sums through global memory.  Suffix sums only need the super-blocks ABOVE:
// tickets hand the super-blocks of a tile out from the top, so whatever a workgroup waits for was started before it.
constexpr int kDiff1BR = 16;           // rows per thread
constexpr int kDiff1W = 16;            // waves (= blocks) per workgroup: 256 rows
constexpr int kDiff1Rows = kDiff1BR * kDiff1W;
struct ColDiff1Args {
  ColDiffArgs a;                       // one rank: row0 = 0, jend = nout
  int NSB, tiles;                      // super-blocks per column, column tiles of 64
  double* tot;                         // [tiles][NSB][2][64] sums of the super-blocks, per parity
  unsigned long long* sync;            // this launch site's area, never reset (see ColHh1Args): [0] ticket counter,
                                       // [1 + tile * NSB + q]: epoch whose sums of super-block q are published
  int* err;                            // raised when a wait ran out
};
RPDE_HD inline long coldiff1_tot(const ColDiff1Args& A, int tile, int q) { return ((long)tile * A.NSB + q) * (2 * kCol1Tile); }

// rows [j0, j0 + BR) of column i (j0 even): d[u] = the sum over the block's rows j' >= j0 + u of the parity of u of
// 2 (j' + 1) c_{j'+1}, tot = the block's sums.  `lowc(u)` = the stencil coefficient low[j0 + u - 1] (k - 2 for k = j0 + u + 1).
template <class LowC>
RPDE_HD inline void coldiff1_local(const ColDiffArgs& a, int j0, int i, double (&d)[kDiff1BR], double (&tot)[2], const LowC& lowc) {
  constexpr int BR = kDiff1BR;
  const double* __restrict__ v = a.in + i;
  double x[BR + 2];                    // x[u] = v_{j0 - 1 + u}
#pragma unroll
  for (int u = 0; u < BR + 2; ++u) { const int k = j0 - 1 + u; x[u] = (k >= 0 && k < a.m) ? v[(long)k * a.ldi] : 0.0; }
  double acc[2] = {0.0, 0.0};
#pragma unroll
  for (int u = BR - 1; u >= 0; --u) {
    const int j = j0 + u, k = j + 1;
    double c = 0.0;
    if (k < a.nout) c = a.low ? x[u + 2] + ((k >= 2) ? lowc(u) * x[u] : 0.0) : x[u + 2];
    if (j < a.nout) acc[u & 1] += 2.0 * (double)(j + 1) * c;
    d[u] = acc[u & 1];
  }
  tot[0] = acc[0]; tot[1] = acc[1];
}
RPDE_HD inline void coldiff1_store(const ColDiffArgs& a, int j0, int i, const double (&d)[kDiff1BR], const double (&in)[2]) {
  double* __restrict__ out = a.out + i;
#pragma unroll
  for (int u = 0; u < kDiff1BR; ++u) {
    const int j = j0 + u;
    if (j < a.nout) out[(long)j * a.ldo] = (d[u] + in[u & 1]) * ((j == 0) ? 0.5 * a.scale : a.scale);
  }
}

}  // namespace rpde
