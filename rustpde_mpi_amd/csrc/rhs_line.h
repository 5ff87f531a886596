// S3 of the confined step as one whole-line kernel: per x-line (one y row j of a YX array)
//
//   forward transform of the convection term (physical x -> orthonormal coefficients, 2/3 rule)      hdct_line.h
//   right-hand side of the Helmholtz problem,                    src/navier_stokes/navier_eq.rs:176-224
//        velx:  rhs = S_x S_y u - dt d/dx p - dt conv
//        vely:  rhs = S_x S_y v - dt d/dy p + dt (S_x S_y T + T_bc) - dt conv
//        temp:  rhs = S_x S_y T + dt ka lap(T_bc) - dt conv
//      (S = composite -> orthonormal stencil, funspace `to_ortho`, src/field.rs:113-115; d/dx p, d/dy p are arrays the
//       step keeps; the y stencil reaches two rows back: rows j and j - 2 of the state)
//   x part of HholtzAdi::solve_par, src/solver/hholtz_adi.rs:149-169:  B2 rows (MatVecFdma, src/solver/matvec.rs:207-228),
//   forward and backward substitution of the swept 4-diagonal system (Fdma::fdma, src/solver/fdma.rs:101-118)
//
// The line program of the same stage (engine.cc `rhs`) runs 512 threads per line with two LDS slots -- two workgroups
// per CU.  Here: 256 threads, ONE line buffer (35 KB, four workgroups per CU).  The element-wise part of the assembly
// happens in the transform's own output layout (thread t owns the pairs k = 2 (t + 256 u), coalesced 16-byte loads of the
// state rows); the banded part needs neighbours along the line and sequential sweeps, so the right-hand side goes through
// the buffer once into CHUNK ownership (thread t owns k = 16 t .. 16 t + 15, padded index k + k / 16: stride 17, no bank
// conflicts) where the sweeps run as chunked scans: each chunk reduced to an affine map of its inflow, the maps
// prefix-composed across the threads (DPP in a wave, wave totals through LDS), every chunk re-run with its exact inflow.
// Per element the arithmetic is that of the reference's sequential sweeps.
#pragma once
#include "hdct_line.h"

namespace rpde {

struct RhsLineArgs {
  int which = 0;                        // 0 velx, 1 vely, 2 temp
  const double* conv = nullptr;         // convection term, physical x (N + 1 values per line)
  const double* st = nullptr;           // state of this field: composite coefficients, rows j and j - 2 are read
  const double* st2 = nullptr;          // vely: the temperature state (buoyancy)
  const double* grad = nullptr;         // velx: d/dx p, vely: d/dy p (orthonormal rows); temp: unused
  const double* tbc = nullptr;          // vely: T_bc rows; temp: lap(T_bc) rows
  int tbc_cols = -1;                    // >= 0: only the first tbc_cols coefficients of a tbc row can be non-zero (a lift that does not
                                        // depend on x has one x-coefficient per row, its Laplacian may vanish altogether): the rest of
                                        // the row is not read (Navier2DEngine::analyse_lift); -1: whole rows
  double* out = nullptr;                // composite coefficients after the x solve (N - 1 per line)
  long ld = 0;                          // all arrays share the pitch
  int nlines = 0, line0 = 0;            // local lines, global index of the first one
  int N = 0, cut = 0;
  double dt = 0.0, ka = 0.0;
  const double* lowy = nullptr;         // y stencil S[j, j - 2] of the state's base (indexed with j - 2)
  const double* lowy2 = nullptr;        // the same for st2
  int stx = 2;                          // x stencil of the field: 2 = Dirichlet (-1), 1 = table lowx
  const double* lowx = nullptr;         // Neumann x stencil table (temp; the buoyancy term of vely)
  const double* tw = nullptr; const double* tw2 = nullptr;
  const double *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;   // B2 rows, chunk-major ascending  [i * T + t] = tab[16 t + i]
  const double* q1 = nullptr;                                  // forward substitution, chunk-major ascending
  const double *p2 = nullptr, *q2 = nullptr, *r2 = nullptr;   // back substitution, chunk-major DESCENDING [i * T + t] = tab[16 (T-1-t) + i]
};
RPDE_HD inline bool rhs_line_ok(const RhsLineArgs& a) {
  return (a.N == 256 || a.N == 1024 || a.N == 4096) && (((size_t)a.conv) & 15) == 0 && (((size_t)a.st) & 15) == 0 && (a.ld & 1) == 0 &&
         a.ld > a.N + 1 && a.which >= 0 && a.which <= 2 && a.stx == (a.which == 2 ? 1 : 2) && (a.which == 0 || a.lowx != nullptr);
}

// out[i * T + t] = tab[tau(t) * 16 + i], tau(t) = t (dir > 0) or T - 1 - t; entries past the end of tab are `pad`
inline std::vector<double> chunk_major16(const std::vector<double>& tab, int T, int dir, double pad = 0.0) {
  std::vector<double> out((size_t)T * 16, pad);
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < 16; ++i) {
      const size_t k = (size_t)(dir > 0 ? t : T - 1 - t) * 16 + i;
      if (k < tab.size()) out[(size_t)i * T + t] = tab[k];
    }
  return out;
}

// exclusive prefix composition of one affine map per thread (and parity): on return (v1, v2) of `cm` hold the inflow
// state of the thread's chunk.  cm: [par][6] = m11 m12 m21 m22 v1 v2; scr: 2 * NW * 6 doubles of LDS.
template <int ORDER, int T>
RPDE_DEV void chunk_prefix(Blk& blk, lds_t scr
#ifdef RPDE_EMU
                           , std::vector<double>& cm_st
#else
                           , double* cm
#endif
) {
  constexpr int W = 6;
#ifdef RPDE_EMU
  (void)blk; (void)scr;
  Affine<ORDER> run[2] = {affine_identity<ORDER>(), affine_identity<ORDER>()};
  for (int t = 0; t < T; ++t)
    for (int par = 0; par < 2; ++par) {
      double* m = &cm_st[(size_t)t * 2 * W + par * W];
      const Affine<ORDER> mine{m[0], m[1], m[2], m[3], m[4], m[5]};
      m[4] = run[par].v1; m[5] = run[par].v2;
      run[par] = affine_compose<ORDER>(mine, run[par]);
    }
#else
  (void)blk;
  constexpr int NW = (T + 63) / 64;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Affine<ORDER> inc[2], exc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    double* m = cm + par * W;
    inc[par] = Affine<ORDER>{m[0], m[1], m[2], m[3], m[4], m[5]};
  }
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    inc[par] = affine_wave_scan<ORDER>(inc[par]);
    exc[par] = affine_dpp<ORDER, 0x138, 0xF>(inc[par]);   // wave_shr:1, lane 0 gets the identity
  }
  if constexpr (NW > 1) {
    if (lane == 63) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        lds_t wt = scr + (par * NW + wave) * W;
        wt[0] = inc[par].m11; wt[1] = inc[par].m12; wt[2] = inc[par].m21; wt[3] = inc[par].m22;
        wt[4] = inc[par].v1; wt[5] = inc[par].v2;
      }
    }
    __syncthreads();
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      Affine<ORDER> pre = affine_identity<ORDER>();
      for (int x = 0; x < wave; ++x) {                     // at most NW - 1 = 3 compositions
        clds_t p = scr + (par * NW + x) * W;
        pre = affine_compose<ORDER>(Affine<ORDER>{p[0], p[1], p[2], p[3], p[4], p[5]}, pre);
      }
      exc[par] = affine_compose<ORDER>(exc[par], pre);
    }
  }
#pragma unroll
  for (int par = 0; par < 2; ++par) { cm[par * W + 4] = exc[par].v1; cm[par * W + 5] = exc[par].v2; }
#endif
}

template <int N, int WHICH>
RPDE_DEV void rhs_line(Blk& blk, const RhsLineArgs& a) {
  using G = HdctGeom<N>;
  constexpr int T = G::T, NW = G::NW, W = 6;
  lds_t buf = (lds_t)blk.lds;
  lds_t scr = buf + G::SCR;
  const int line = blk.line, gline = line + a.line0;
  const long off = (long)line * a.ld;
  const bool has2 = gline >= 2;
  const double dt = a.dt;

  // ---- forward transform of the convection term; its results leave as -dt c_k (pairs m = 2 (t + u T)) straight into the line
  // buffer at the padded index the assembly below uses -- the same thread owns the same pairs there, and when the transform emits
  // (behind its last barrier) nobody reads the exchange planes any more.  (Rounds 3 - 4 kept them in 17 registers per thread
  // across the assembly: 184 registers at a budget of 168, i.e. 18 - 20 spilled ones whose scratch traffic was a quarter of the
  // stage's HBM bytes -- PMC ratio 1.30, profiles/r05_pmc_traffic.txt.)
  DctLineArgs f{a.conv, a.ld, N + 1, nullptr, 0, a.nlines, N, 0, a.tw, a.tw2, 1.0};
  f.fwd = 1; f.cut = a.cut;
  hdct_core<N>(blk, f, false, HdctNoFetch{}, [&](int tid, int u, int m, double e0, double e1) {
    (void)m;                                                // m = 2 (tid + u T); 2 u T is a multiple of 16: one runtime term per thread
    const int q = (u == 8) ? N + (N >> 4) + 2 : 2 * tid + ((2 * tid) >> 4) + 2 + 17 * (u * T / 8);
    buf[q] = -dt * e0;
    if (u != 8) buf[q + 1] = -dt * e1;                      // (u = 8: m = N, the single coefficient of thread 0)
  });

  // ---- element-wise assembly in the same layout, then into the buffer at the padded index k + k / 16 (+ 2; the
  // taps of the B2 rows reach k + 4 <= N + 3: zeros there).  No barrier in front: the transform's last phase reads the
  // scratch area only.  State rows are zero behind their N - 1 coefficients (nothing ever writes there), so the loads
  // need no predicates; the pair in front of the line (k = -2, -1) is the only one that may not be read.
#ifndef RPDE_EMU
  asm volatile("" ::: "memory");                            // the loads below stay below: they would cost the transform its registers
  __builtin_amdgcn_sched_barrier(0);
#endif
  constexpr bool stab = (WHICH == 2);                   // the temperature is Neumann in x (table stencil), the velocities Dirichlet
  const double cy = has2 ? ((tab_t)a.lowy)[gline - 2] : 0.0;
  const double cy2 = (has2 && (WHICH == 1)) ? ((tab_t)a.lowy2)[gline - 2] : 0.0;
  const long off2 = has2 ? off - 2 * a.ld : off;           // row j - 2 (row j with a zero factor when it does not exist)
  const long rowb = (long)(N + 2) * 8;                      // bytes of a row that may be touched (pairs up to N, N + 1)
  const RowBuf s0 = row_buf(a.st + off, rowb), s2 = row_buf(a.st + off2, rowb);
  const RowBuf t0r = row_buf(((WHICH == 1) ? a.st2 : a.st) + off, rowb), t2r = row_buf(((WHICH == 1) ? a.st2 : a.st) + off2, rowb);
  // the tbc row when only its first tcols <= 2 T coefficients can be non-zero: they lie in block u = 0 of the pairs (m = 2 (t + u T)),
  // so the other blocks are not loaded at all and the descriptor of the row ends behind them (block 0 has no uniform offset: the
  // range check of the per-thread offset is exact); a longer non-zero part reads whole rows
  const bool tpart = a.tbc_cols >= 0 && a.tbc_cols <= 2 * T;
  const int tcols = tpart ? a.tbc_cols : N + 2;
  const long tbcb = tpart ? 8L * ((tcols + 1) & ~1) : rowb;
  const int tub = tpart ? (tcols > 0 ? 1 : 0) : 8;
  const RowBuf g2 = row_buf(((WHICH == 2) ? a.tbc : a.grad) + off, (WHICH == 2) ? tbcb : rowb), b2 = row_buf(((WHICH == 1) ? a.tbc : a.conv) + off, (WHICH == 1) ? tbcb : rowb);
  const RowBuf lx2 = row_buf((stab || (WHICH == 1)) ? a.lowx : a.tw2, rowb);
  const double gfac = (WHICH == 2) ? dt * a.ka : -dt;      // factor of the row g2: d/dx p, d/dy p (- dt) or lap(T_bc) (dt ka)
  RPDE_PHASE(blk, tid) {
    const int v = 16 * tid, vm = v - 16;                    // byte offsets of the thread's pair and of the pair in front of it
    constexpr int NB = (WHICH == 1) ? 2 : 4;                // pairs per batch (vely has eleven rows in flight per pair)
#pragma unroll
    for (int h = 0; h < 8 / NB; ++h) {
      // all loads of a batch of pairs go out together and are pinned there (left alone, the compiler sinks every load to its
      // first use and waits for it: one memory round trip per load)
      dbl2 a0[NB], am[NB], c0[NB], cm[NB], g[NB], lt[NB], b0[NB], bm[NB], d0[NB], dm[NB], tb[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int u = NB * h + i, so = u * T * 16;           // uniform part of the offset: u T pairs
        // the pair in front: one pair less in the uniform part -- except in the first block, where it is the per-thread
        // part that steps back and thread 0 reads zero (in front of the line)
        const int wm = (u == 0) ? vm : v, som = (u == 0) ? 0 : so - 16;
        a0[i] = row_ld2(s0, v, so); am[i] = row_ld2(s0, wm, som); c0[i] = row_ld2(s2, v, so); cm[i] = row_ld2(s2, wm, som);
        g[i] = ((WHICH != 2) || u < tub) ? row_ld2(g2, v, so) : dbl2{0.0, 0.0};
        lt[i] = (stab || (WHICH == 1)) ? row_ld2(lx2, wm, som) : dbl2{0.0, 0.0};
        if (WHICH == 1) {
          b0[i] = row_ld2(t0r, v, so); bm[i] = row_ld2(t0r, wm, som); d0[i] = row_ld2(t2r, v, so); dm[i] = row_ld2(t2r, wm, som);
          tb[i] = (u < tub) ? row_ld2(b2, v, so) : dbl2{0.0, 0.0};
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        RPDE_PIN(a0[i].x); RPDE_PIN(a0[i].y); RPDE_PIN(am[i].x); RPDE_PIN(am[i].y); RPDE_PIN(c0[i].x); RPDE_PIN(c0[i].y);
        RPDE_PIN(cm[i].x); RPDE_PIN(cm[i].y); RPDE_PIN(g[i].x); RPDE_PIN(g[i].y);
        if (stab || (WHICH == 1)) { RPDE_PIN(lt[i].x); RPDE_PIN(lt[i].y); }
        if (WHICH == 1) {
          RPDE_PIN(b0[i].x); RPDE_PIN(b0[i].y); RPDE_PIN(bm[i].x); RPDE_PIN(bm[i].y); RPDE_PIN(d0[i].x); RPDE_PIN(d0[i].y);
          RPDE_PIN(dm[i].x); RPDE_PIN(dm[i].y); RPDE_PIN(tb[i].x); RPDE_PIN(tb[i].y);
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int u = NB * h + i;
        const dbl2 lx = stab ? lt[i] : dbl2{-1.0, -1.0};
        // S_x S_y state: rows j, j - 2 at k and k - 2
        const double sx = a0[i].x + cy * c0[i].x, sy = a0[i].y + cy * c0[i].y;
        const double smx = am[i].x + cy * cm[i].x, smy = am[i].y + cy * cm[i].y;
        const int q = 2 * tid + ((2 * tid) >> 4) + 2 + 17 * (u * T / 8);   // m = 2 (tid + u T) at m + m / 16 + 2
        double rx = buf[q] + (sx + lx.x * smx) + gfac * g[i].x;
        double ry = buf[q + 1] + (sy + lx.y * smy) + gfac * g[i].y;
        if (WHICH == 1) {                                   // buoyancy: dt (S_xN S_y T + T_bc)
          rx += dt * ((b0[i].x + cy2 * d0[i].x) + lt[i].x * (bm[i].x + cy2 * dm[i].x) + tb[i].x);
          ry += dt * ((b0[i].y + cy2 * d0[i].y) + lt[i].y * (bm[i].y + cy2 * dm[i].y) + tb[i].y);
        }
        buf[q] = rx;
        buf[q + 1] = ry;                                    // m + 1 stays inside the group of 16
      }
#ifndef RPDE_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (tid == 0) {                                         // k = N: no own coefficient, the stencil tap of k - 2 only; zeros behind
      const int sn = (N - 2) * 8;
      const double lxn = stab ? row_ld1(lx2, 0, sn) : -1.0;
      double rn = buf[N + (N >> 4) + 2] + lxn * (row_ld1(s0, 0, sn) + cy * row_ld1(s2, 0, sn)) + gfac * (((WHICH != 2) || tcols > N) ? row_ld1(g2, 0, N * 8) : 0.0);
      if ((WHICH == 1))
        rn += dt * (row_ld1(lx2, 0, sn) * (row_ld1(t0r, 0, sn) + cy2 * row_ld1(t2r, 0, sn)) + ((tcols > N) ? row_ld1(b2, 0, N * 8) : 0.0));
      buf[N + (N >> 4) + 2] = rn;
#pragma unroll
      for (int k = N + 1; k <= N + 4; ++k) buf[k + (k >> 4) + 2] = 0.0;
    }
  }
  RPDE_SYNC(blk);

  // ---- B2 rows + forward substitution, thread t owns k = 16 t .. 16 t + 15 (ascending: the carry flows t - 1 -> t)
  //   b_k = t0_k r_k + t1_k r_{k+2} + t2_k r_{k+4} (k < N - 1; the last tap only for k < N - 3),  y_k = b_k + q1_k y_{k-2}
  const int n = N - 1;
  RPDE_TLS(blk, double, y, 16);
  RPDE_TLS(blk, double, cm, 2 * W);
  RPDE_TLS(blk, double, qa, 16);
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * tid;
    const ChunkTab t0 = chunk_tab(a.t0, T), t1 = chunk_tab(a.t1, T), t2 = chunk_tab(a.t2, T), q1 = chunk_tab(a.q1, T);
    double r[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) { const int k = k0 + i; r[i] = buf[k + (k >> 4) + 2]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) RPDE_T(qa)[i] = chunk_ld(q1, tid, i, T);
#pragma unroll
    for (int i = 0; i < 16; ++i) RPDE_PIN(RPDE_T(qa)[i]);
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

  // ---- back substitution, descending: thread t owns the chunk of thread T - 1 - t (the carry flows t - 1 -> t again)
  //   x_k = p2_k y_k + q2_k x_{k+2} + r2_k x_{k+4}
  RPDE_TLS(blk, double, bb, 16);
  RPDE_PHASE(blk, tid) {
    const int k0 = 16 * (T - 1 - tid);
    const ChunkTab p2 = chunk_tab(a.p2, T), q2 = chunk_tab(a.q2, T), r2 = chunk_tab(a.r2, T);
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
    const ChunkTab q2 = chunk_tab(a.q2, T), r2 = chunk_tab(a.r2, T);
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
  // ---- the solution leaves in pairs, coalesced
  RPDE_PHASE(blk, tid) {
    gmem2_t dst = (gmem2_t)(a.out + off);
    gmem_t dst1 = (gmem_t)(a.out + off);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (tid + u * T);
      const int p = m + (m >> 4) + 2;
      const dbl2 v = dbl2{buf[p], buf[p + 1]};
      if (m + 1 < n) dst[m >> 1] = v;
      else if (m < n) dst1[m] = v.x;
    }
  }
}

}  // namespace rpde
