#include "ops.h"

#include <algorithm>
#include <cmath>

namespace rpde {

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Transform plan of an axis.  Power-of-two lengths (Chebyshev n = 2^k + 1, Fourier nx = 2^k) keep the packed plans the whole-line
// kernels are built on; EVERY other length runs Bluestein's algorithm through the same power-of-two FFT (line_vm.h) -- the
// reference accepts any n (funspace over rustdct / rustfft / realfft; benches/benchmark_navier.rs:6-7 runs 128, 264, 512).
// Limits: the padded work area of M complex numbers must fit the LDS slots of one workgroup -- two slots for a Chebyshev axis
// (M <= 8192: n <= 4097, like the power-of-two plans), one for a Fourier axis (M <= 8192: nx <= 5461).
AxisTables::AxisTables(const Base& b) : base(b) {
  static const bool force_direct = std::getenv("RPDE_DCT_DIRECT") && std::atoi(std::getenv("RPDE_DCT_DIRECT")) != 0;
  if (b.is_cheb()) {
    const int N = b.n - 1;
    RPDE_REQUIRE(N >= 1, "a Chebyshev axis needs at least two points");
    slot_len = slot_len_for(b.n);
    if (is_pow2(N) && N >= 2 && N <= 4096) {
      fft_n = N;
      tw.upload(fft_twiddles(N));
      tw2.upload(dct_split_twiddles(N));
    } else if (force_direct && b.n <= 500) {
      tw2.upload(dct_direct_costab(N));
    } else {
      blu_m = bluestein_len(2 * N + 1);
      RPDE_REQUIRE(blu_m <= 8192, "Chebyshev axis of " + std::to_string(b.n) + " points: at most 4097 "
                   "(the work area of one line must fit the LDS of a workgroup)");
      tw.upload(fft_twiddles(blu_m));
      blu.upload(bluestein_dct_tables(N, blu_m));
      slot_len = std::max(slot_len, (fft_work_doubles(blu_m) / 2 + 1) & ~1);
    }
    fwd_post.upload(cheb_fwd_post(b.n));
    bwd_pre.upload(cheb_bwd_pre(b.n));
    Mv3Tables pv = pinv_tables(b);
    pv0.upload(pv.t0); pv1.upload(pv.t1); pv2.upload(pv.t2);
    if (b.kind == kChebDirichletNeumann) {
      low.upload(stencil_low(b));
      low1.upload(stencil_low1(b));
      fo_pdma.upload(pdma_factor(from_ortho7(b)));
      ortho = std::make_unique<AxisTables>(make_base(kChebyshev, b.n));
    } else if (b.is_composite()) {
      low.upload(stencil_low(b));
      FromOrthoTables f = from_ortho_tables(b);
      fo_t0.upload(f.t0); fo_t1.upload(f.t1); fo_t2.upload(f.t2);
      const LineClass lc = line_class_for(slot_len);
      fo_pup.upload(chunk_major(f.p_up, lc, +1)); fo_qup.upload(chunk_major(f.q_up, lc, +1));
      fo_qdn.upload(chunk_major(f.q_dn, lc, -1));
    }
  } else {
    RPDE_REQUIRE(b.n >= 2, "fourier_r2c needs nx >= 2");
    slot_len = slot_len_for(b.n + 2);
    if (is_pow2(b.n) && b.n >= 4) {
      RPDE_REQUIRE(b.n <= 16384, "fourier_r2c: nx = 2^k up to 16384");
      fft_n = b.n / 2;
      tw.upload(fft_twiddles(fft_n));
      tw2.upload(rfft_split_twiddles(b.n));
    } else {
      blu_m = bluestein_len(b.n + b.n / 2);
      RPDE_REQUIRE(blu_m <= 8192, "fourier_r2c of " + std::to_string(b.n) + " points: lengths other than nx = 2^k are supported "
                   "up to nx = 5461 (the Bluestein work area of one line must fit the LDS of a workgroup)");
      tw.upload(fft_twiddles(blu_m));
      blu.upload(bluestein_rfft_tables(b.n, blu_m));
      slot_len = std::max(slot_len, fft_work_doubles(blu_m));
    }
  }
}

void PdmaDev::upload(const PdmaTables& t) {
  n = t.n;
  host = t;
  l2.upload(t.l2); ka.upload(t.ka); imu.upload(t.imu); al.upload(t.al); be.upload(t.be); ga.upload(t.ga); de.upload(t.de);
}
void PdmaDev::upload_blocks() {
  const PdmaBlockTables b = pdma_block_tables(host, kPdmaBR);
  NB = b.NB;
  phi1.upload(b.phi1); phi2.upload(b.phi2); fm.upload(b.fm);
  psi1.upload(b.psi1); psi2.upload(b.psi2); psi3.upload(b.psi3); psi4.upload(b.psi4); bm.upload(b.bm);
}

FdmaDev upload_fdma(const FdmaTables& t, int slot_len) {
  FdmaDev d;
  d.n = (int)t.p2.size();
  const LineClass lc = line_class_for(slot_len);
  d.q1.upload(chunk_major(t.q1, lc, +1));
  d.p2.upload(chunk_major(t.p2, lc, -1, 1.0));
  d.q2.upload(chunk_major(t.q2, lc, -1));
  d.r2.upload(chunk_major(t.r2, lc, -1));
  return d;
}

// ------------------------------------------------------------------------------------------
ProgramBuilder::ProgramBuilder(int nslots, int slot_len, int nlines, int ncomp) {
  pg.nslots = nslots;
  pg.slot_len = slot_len;
  pg.nlines = nlines;
  pg.ncomp = ncomp;
  pg.tw = pg.tw2 = 0;
  pg.blu_m = 0;
}
void ProgramBuilder::set_fft(const AxisTables& ax) {
  ax_ = &ax;
  pg.fft_n = ax.fft_n;
  pg.blu_m = ax.blu_m;
  pg.tw = ax.tw.p ? tab(ax.tw.p) : 0;
  pg.tw2 = tab(ax.blu_m > 0 ? ax.blu.p : ax.tw2.p);
}
int ProgramBuilder::arr(double* p, long ld, int es, long coff) {
  for (int i = 0; i < narr_; ++i)
    if (pg.arr[i].p == p && pg.arr[i].ld == ld && pg.arr[i].es == es && pg.arr[i].coff == coff) return i;
  RPDE_REQUIRE(narr_ < kMaxArr, "too many arrays in a line program");
  pg.arr[narr_] = ArrayRef{p, ld, coff, es, 0};
  return narr_++;
}
int ProgramBuilder::tab(const double* t) {
  RPDE_REQUIRE(t != nullptr, "null table");
  for (int i = 0; i < ntab_; ++i)
    if (pg.tabs[i] == t) return i;
  RPDE_REQUIRE(ntab_ < kMaxTab, "too many tables in a line program");
  pg.tabs[ntab_] = t;
  return ntab_++;
}
Op& ProgramBuilder::push(int code) {
  RPDE_REQUIRE(pg.nops < kMaxOps, "line program too long");
  Op& o = pg.ops[pg.nops++];
  o = Op{};
  o.code = code;
  o.tab = -1;
  o.s0 = 1.0;
  return o;
}
void ProgramBuilder::load(int d, int a, int n, double s0, bool acc, int half) {
  RPDE_REQUIRE(n <= pg.slot_len, "line longer than the slot");
  Op& o = push(OP_LOAD); o.d = d; o.arr = a; o.n = n; o.s0 = s0; o.acc = acc; o.i0 = half > 0; o.i1 = half;
}
void ProgramBuilder::load_cik(int d, int a, int n, double s0, bool acc) {
  RPDE_REQUIRE(n <= pg.slot_len && n % 2 == 0, "load_cik: interleaved complex line expected");
  Op& o = push(OP_LOAD); o.d = d; o.arr = a; o.n = n; o.s0 = s0; o.acc = acc; o.i0 = 2; o.i1 = 0;
}
void ProgramBuilder::loadmul(int d, int a, int n, double s0) {
  Op& o = push(OP_LOAD); o.d = d; o.arr = a; o.n = n; o.s0 = s0; o.acc = 2;
}
void ProgramBuilder::loadx(int d, int a, int n, int rows, const double* lowtab, double s0, bool acc, int half) {
  Op& o = push(OP_LOADX); o.d = d; o.arr = a; o.n = n; o.i1 = rows; o.tab = tab(lowtab); o.s0 = s0; o.acc = acc; o.i0 = half;
}
void ProgramBuilder::pair_last_loads() {
  RPDE_REQUIRE(pg.nops >= 2, "pair_last_loads: two load ops expected");
  Op& o1 = pg.ops[pg.nops - 2];
  const Op& o2 = pg.ops[pg.nops - 1];
  RPDE_REQUIRE(o2.code == OP_LOAD && o2.i0 == 0 && pg.arr[o2.arr].es == 1, "pair_last_loads: the second op must be a plain load");
  RPDE_REQUIRE((o1.code == OP_LOAD && o1.i0 == 0 && pg.arr[o1.arr].es == 1 && o1.b == 0) ||
                   (o1.code == OP_LOADX && o1.b == 0 && o1.i0 == 0),
               "pair_last_loads: the first op must be a plain load or a cross-line load");
  RPDE_REQUIRE(pg.nops < 3 || pg.ops[pg.nops - 3].b == 0 || (pg.ops[pg.nops - 3].code != OP_LOAD && pg.ops[pg.nops - 3].code != OP_LOADX),
               "pair_last_loads: the first op already belongs to a pair");
  o1.b = 1;
}
void ProgramBuilder::store(int a, int ar, int n, double s0, int half) {
  Op& o = push(OP_STORE); o.a = a; o.arr = ar; o.n = n; o.s0 = s0; o.i0 = half > 0; o.i1 = half;
}
void ProgramBuilder::guard_last_store(int* flag) {
  RPDE_REQUIRE(pg.nops > 0 && pg.ops[pg.nops - 1].code == OP_STORE && pg.ops[pg.nops - 1].i0 == 0,
               "guard_last_store: the last op must be a plain store");
  RPDE_REQUIRE(pg.arr[pg.ops[pg.nops - 1].arr].es == 1, "guard_last_store: contiguous lines only");
  pg.ops[pg.nops - 1].acc = 1;
  pg.nanflag = flag;
}
static void two_term_only(const AxisTables& ax) {
  RPDE_REQUIRE(!ax.base.is_composite() || ax.base.is_two_term(), "line programs hold two-term stencils only (pdma.h)");
}
void ProgramBuilder::sten(int d, int a, int n_ortho, const double* low) {
  Op& o = push(OP_STEN); o.d = d; o.a = a; o.n = n_ortho; o.tab = tab(low);
}
void ProgramBuilder::mv3(int d, int a, int n, const double* t0, const double* t1, const double* t2, long tabld) {
  const int i0 = tab(t0), i1 = tab(t1), i2 = tab(t2);
  RPDE_REQUIRE(i1 == i0 + 1 && i2 == i0 + 2, "mv3 tables must be registered consecutively");
  Op& o = push(OP_MV3); o.d = d; o.a = a; o.n = n; o.tab = i0; o.tabld = tabld;
}
void ProgramBuilder::cdiff(int d, int a, int n, double scale) {
  Op& o = push(OP_CDIFF); o.d = d; o.a = a; o.n = n; o.s0 = scale;
}
void ProgramBuilder::rec1(int d, int a, int n, const double* p, const double* q, int dir, long tabld) {
  const int ip = p ? tab(p) : -1, iq = tab(q);
  Op& o = push(OP_REC1); o.d = d; o.a = a; o.n = n; o.tab = ip; o.i0 = iq; o.i1 = dir; o.tabld = tabld;
}
void ProgramBuilder::rec2(int d, int a, int n, const double* p, const double* q, const double* r, long tabld) {
  const int ip = p ? tab(p) : -1, iq = tab(q), ir = tab(r);
  // the kernel parks the r coefficients in the LAST slot (the only one addressable up to T * EPT)
  const int scratch = pg.nslots - 1;
  RPDE_REQUIRE(scratch != d && scratch != a, "OP_REC2 needs the last slot as scratch");
  Op& o = push(OP_REC2); o.d = d; o.a = a; o.b = scratch; o.n = n; o.tab = ip; o.i0 = iq; o.i1 = ir; o.tabld = tabld;
}
// FFT path: the scalings are flags (tab = pre on/off, i0 = post on/off), a = first zeroed
// coefficient, s1 = 1/N; direct path: table indices as before
void ProgramBuilder::dct_flags(Op& o, int n, const double* pre, const double* post, int cut) {
  RPDE_REQUIRE(ax_ != nullptr && ax_->base.n == n, "OP_DCT: set_fft(axis) must name the transformed axis");
  RPDE_REQUIRE(pre == nullptr || pre == ax_->bwd_pre.p, "OP_DCT (FFT path): only the standard backward pre-scaling");
  RPDE_REQUIRE(post == nullptr || post == ax_->fwd_post.p || cut >= 0,
               "OP_DCT (FFT path): only the standard forward post-scaling (optionally cut)");
  o.tab = pre ? 0 : -1;
  o.i0 = post ? 0 : -1;
  o.a = (cut >= 0 && cut < n) ? cut : n;
  o.s1 = 1.0 / (double)(n - 1);
}
void ProgramBuilder::dct(int d, int n, const double* pre, const double* post, int cut) {
  RPDE_REQUIRE(d + 1 < pg.nslots, "OP_DCT needs slot d+1 as scratch");
  Op& o = push(OP_DCT); o.d = d; o.n = n; o.i1 = -1; o.arr = -1;
  if (pg.fft_n > 0) {
    dct_flags(o, n, pre, post, cut);
  } else {
    o.tab = pre ? tab(pre) : -1; o.i0 = post ? tab(post) : -1;
  }
}
void ProgramBuilder::dct_fused(int d, const AxisTables& ax, bool sten_, const double* pre,
                               const double* post, int store_arr, int nstore, double scale, int cut) {
  const int n = ax.base.n;
  if (ax.fft_n == 0) {   // direct transform: no fused forms
    if (sten_) to_ortho(d, ax);
    dct(d, n, pre, post, cut);
    if (store_arr >= 0) store(d, store_arr, nstore, scale);
    return;
  }
  RPDE_REQUIRE(d + 1 < pg.nslots, "OP_DCT needs slot d+1 as scratch");
  RPDE_REQUIRE(!sten_ || !ax.base.is_composite() || ax.base.is_two_term(), "line programs hold two-term stencils only (pdma.h)");
  // the Dirichlet stencil is the constant -1: no table
  const int ilow = !(sten_ && ax.base.is_composite()) ? -1 : (ax.base.kind == kChebDirichlet ? -2 : tab(ax.low.p));
  Op& o = push(OP_DCT); o.d = d; o.n = n; o.i1 = ilow;
  dct_flags(o, n, pre, post, cut);
  o.arr = store_arr; o.b = nstore; o.s0 = scale;
}
void ProgramBuilder::mul(int d, int a, int b, int n, double s0, bool acc) {
  Op& o = push(OP_MUL); o.d = d; o.a = a; o.b = b; o.n = n; o.s0 = s0; o.acc = acc;
}
void ProgramBuilder::axpby(int d, int a, double s0, int b, double s1, int n) {
  Op& o = push(OP_AXPBY); o.d = d; o.a = a; o.b = b; o.n = n; o.s0 = s0; o.s1 = s1;
}
void ProgramBuilder::stash(int a) { Op& o = push(OP_PUSH); o.a = a; }
void ProgramBuilder::unstash_axpy(int d, double s0, double s1, int n) {
  Op& o = push(OP_POPAXPY); o.d = d; o.n = n; o.s0 = s0; o.s1 = s1;
}
void ProgramBuilder::zero(int d, int from, int to) {
  Op& o = push(OP_ZERO); o.d = d; o.i0 = from; o.i1 = to;
}
void ProgramBuilder::tabdiv(int d, int a, int n, const double* t, int shift) {
  Op& o = push(OP_TABDIV); o.d = d; o.a = a; o.n = n; o.tab = tab(t); o.i0 = shift;
}
void ProgramBuilder::rfft_f(int d, int nx) { Op& o = push(OP_RFFT_F); o.d = d; o.n = nx; }
void ProgramBuilder::rfft_b(int d, int nx) { Op& o = push(OP_RFFT_B); o.d = d; o.n = nx; }
void ProgramBuilder::cik(int d, int a, int nc, double s0, int power) {
  Op& o = push(OP_CIK); o.d = d; o.a = a; o.n = nc; o.s0 = s0; o.i0 = power;
}
void ProgramBuilder::to_ortho(int d, const AxisTables& ax) {
  two_term_only(ax);
  if (ax.base.is_composite()) sten(d, d, ax.base.n, ax.low.p);
}
void ProgramBuilder::to_ortho_axpby(int d, double sd, int a, double sa, const AxisTables& ax) {
  RPDE_REQUIRE(d != a, "to_ortho_axpby: out of place only");
  two_term_only(ax);
  if (ax.base.is_composite()) {
    sten(d, a, ax.base.n, ax.low.p);
    Op& o = pg.ops[pg.nops - 1];
    o.acc = 1; o.s1 = sd; o.s0 = sa;
  } else {
    axpby(d, d, sd, a, sa, ax.base.n);
  }
}
void ProgramBuilder::to_ortho_from(int d, int a, const AxisTables& ax) {
  two_term_only(ax);
  if (ax.base.is_composite()) sten(d, a, ax.base.n, ax.low.p);
  else axpby(d, a, 1.0, a, 0.0, ax.base.n);
}
void ProgramBuilder::from_ortho(int d, const AxisTables& ax) {
  if (!ax.base.is_composite()) return;
  two_term_only(ax);
  const int m = ax.base.m;
  mv3(d, d, m, ax.fo_t0.p, ax.fo_t1.p, ax.fo_t2.p);
  rec1(d, d, m, ax.fo_pup.p, ax.fo_qup.p, +1);
  rec1(d, d, m, nullptr, ax.fo_qdn.p, -1);
}
void ProgramBuilder::fdma_solve(int d, int n, const FdmaDev& f) {
  // per-line tables hold the lines [row0, ...): the kernel indexes them with the global line number
  const long off = f.row0 * f.tabld;
  rec1(d, d, n, nullptr, f.q1.p - off, +1, f.tabld);
  rec2(d, d, n, f.p2.p - off, f.q2.p - off, f.r2.p - off, f.tabld);
}
void ProgramBuilder::pinv_matvec(int d, const AxisTables& ax) {
  mv3(d, d, ax.base.n - 2, ax.pv0.p, ax.pv1.p, ax.pv2.p);
}

// ------------------------------------------------------------------------------------------
Space2Ops::Space2Ops(const Base& b0, const Base& b1) {
  RPDE_REQUIRE(b1.is_cheb(), "axis 1 must be a Chebyshev-family base");
  ax_[0] = std::make_unique<AxisTables>(b0);
  ax_[1] = std::make_unique<AxisTables>(b1);
}

void Space2Ops::run_lines(Kind kind, const AxisTables& ax, const double* in, long ldi, int len_in,
                          double* out, long ldo, int len_out, int nlines, int ncomp, Stream& st,
                          int order, double scale, const FdmaDev* fd, const double* diag, const PdmaDev* pd) {
  if (ax.base.kind == kChebDirichletNeumann) {
    run_lines3(kind, ax, in, ldi, len_in, out, ldo, len_out, nlines, ncomp, st, order, scale, pd);
    return;
  }
  // ncomp = 2: the lines are interleaved complex but the op is real (acts on re and im alike)
  // Chebyshev kinds need the second slot (DCT work area, scratch of the banded solve); a Fourier
  // axis never does, and its longest configuration has room for one slot only
  ProgramBuilder pb(ax.base.is_cheb() ? 2 : 1, ax.slot_len, nlines, ncomp);
  pb.set_fft(ax);
  const int es = ncomp, coff = ncomp == 2 ? 1 : 0;
  const int ai = pb.arr(in, ldi, es, coff);
  const int ao = pb.arr(out, ldo, es, coff);
  const Base& b = ax.base;
  pb.load(0, ai, len_in);
  switch (kind) {
    case kToOrtho: pb.to_ortho(0, ax); break;
    case kFromOrtho: pb.from_ortho(0, ax); break;
    case kForwardOrtho:
    case kForward:
      if (b.is_cheb()) {
        pb.dct(0, b.n, nullptr, ax.fwd_post.p);
        if (kind == kForward) pb.from_ortho(0, ax);
      } else {
        pb.rfft_f(0, b.n);
      }
      break;
    case kBackwardOrtho:
    case kBackward:
      if (b.is_cheb()) {
        if (kind == kBackward) pb.to_ortho(0, ax);
        pb.dct(0, b.n, ax.bwd_pre.p, nullptr);
      } else {
        pb.rfft_b(0, b.n);
      }
      break;
    case kDiff:
      if (b.is_cheb()) {
        pb.to_ortho(0, ax);
        for (int o = 0; o < order; ++o) pb.cdiff(0, 0, b.n, 1.0 / scale);
      } else {
        if (order > 0) pb.cik(0, 0, b.m, 1.0 / scale, order);
      }
      break;
    case kDiffBackward:
      if (b.is_cheb()) {
        pb.to_ortho(0, ax);
        for (int o = 0; o < order; ++o) pb.cdiff(0, 0, b.n, 1.0 / scale);
        pb.dct(0, b.n, ax.bwd_pre.p, nullptr);
      } else {
        if (order > 0) pb.cik(0, 0, b.m, 1.0 / scale, order);
        pb.rfft_b(0, b.n);
      }
      break;
    case kPinvMatvec: pb.pinv_matvec(0, ax); break;
    case kFdmaSolve: pb.fdma_solve(0, len_in, *fd); break;
    case kDiagSolve: pb.tabdiv(0, 0, len_in, diag, b.is_cheb() ? 0 : 1); break;
  }
  pb.store(0, ao, len_out);
  pb.run(st);
}

// A three-term axis (cheb_dirichlet_neumann): the stencil and the banded solves are the pdma.h kernels, everything else
// is a line program of the orthonormal parent.  Setup / diagnostics / operator API only -- the time step has its own
// YX-layout kernels (engine.cc).
void Space2Ops::run_lines3(Kind kind, const AxisTables& ax, const double* in, long ldi, int len_in, double* out, long ldo,
                           int len_out, int nlines, int ncomp, Stream& st, int order, double scale, const PdmaDev* pd) {
  const AxisTables& ox = *ax.ortho;
  const int n = ax.base.n, m = ax.base.m;
  const long ldt = pitch((long)n * ncomp);
  auto sten3 = [&](const double* src, long lds_, double* dst, long ldd) {
    launch_sten3_lines(Sten3LinesArgs{src, lds_, dst, ldd, nlines, m, ncomp, ncomp, ax.low1.p, ax.low.p}, st);
  };
  auto solve = [&](const double* src, long lds_, double* dst, long ldd, bool normal_eq, const PdmaDev& f) {
    launch_pdma_lines(PdmaLinesArgs{src, lds_, dst, ldd, nlines, m, ncomp, ncomp, normal_eq ? ax.low1.p : nullptr,
                                    normal_eq ? ax.low.p : nullptr, f.tabs()}, st);
  };
  switch (kind) {
    case kToOrtho: sten3(in, ldi, out, ldo); return;
    case kFromOrtho: solve(in, ldi, out, ldo, true, ax.fo_pdma); return;
    case kForward: {
      Arr2& t = scr_.get(3, nlines, n, ncomp);
      run_lines(kForwardOrtho, ox, in, ldi, n, t.p(), ldt, n, nlines, ncomp, st, 0, 1.0, nullptr, nullptr);
      solve(t.p(), ldt, out, ldo, true, ax.fo_pdma);
      return;
    }
    case kBackward: {
      Arr2& t = scr_.get(3, nlines, n, ncomp);
      sten3(in, ldi, t.p(), ldt);
      run_lines(kBackwardOrtho, ox, t.p(), ldt, n, out, ldo, n, nlines, ncomp, st, 0, 1.0, nullptr, nullptr);
      return;
    }
    case kDiff: {
      Arr2& t = scr_.get(3, nlines, n, ncomp);
      sten3(in, ldi, t.p(), ldt);
      run_lines(kDiff, ox, t.p(), ldt, n, out, ldo, n, nlines, ncomp, st, order, scale, nullptr, nullptr);
      return;
    }
    case kDiffBackward: {
      Arr2& t = scr_.get(3, nlines, n, ncomp);
      sten3(in, ldi, t.p(), ldt);
      run_lines(kDiffBackward, ox, t.p(), ldt, n, out, ldo, n, nlines, ncomp, st, order, scale, nullptr, nullptr);
      return;
    }
    case kForwardOrtho: case kBackwardOrtho: case kPinvMatvec:
      run_lines(kind, ox, in, ldi, len_in, out, ldo, len_out, nlines, ncomp, st, order, scale, nullptr, nullptr);
      return;
    case kFdmaSolve:
      RPDE_REQUIRE(pd != nullptr && pd->n == m, "banded solve along a three-term axis needs its PdmaPlus2 tables");
      solve(in, ldi, out, ldo, false, *pd);
      return;
    case kDiagSolve: break;
  }
  fail("operator not defined for a cheb_dirichlet_neumann axis");
}

void Space2Ops::apply_axis(Kind kind, int axis, const Arr2& in, Arr2& out, Stream& st, int order,
                           double scale, const FdmaDev* fd, const double* diag, const PdmaDev* pd) {
  const AxisTables& ax = *ax_[axis];
  const Base& b = ax.base;
  // element counts along the axis, in and out, and element types
  auto len_of = [&](bool input) -> int {
    const int n = b.n, m = b.m, no = b.n_ortho();
    switch (kind) {
      case kToOrtho: return input ? m : no;
      case kFromOrtho: return input ? no : m;
      case kForwardOrtho: return input ? n : no;
      case kForward: return input ? n : m;
      case kBackwardOrtho: return input ? no : n;
      case kBackward: return input ? m : n;
      case kDiff: return input ? m : no;
      case kDiffBackward: return input ? m : n;
      case kPinvMatvec: return input ? n : n - 2;
      case kFdmaSolve: case kDiagSolve: return m;
    }
    return 0;
  };
  const int li = len_of(true), lo = len_of(false);
  if (axis == 1) {
    RPDE_REQUIRE(in.cols == li && out.cols == lo && in.rows == out.rows && in.elem == out.elem,
                 "shape mismatch in axis-1 operator");
    run_lines(kind, ax, in.p(), in.ld, li, out.p(), out.ld, lo, in.rows, in.elem, st, order, scale,
              fd, diag, pd);
    return;
  }
  // axis 0: transpose, run along the now contiguous axis, transpose back
  const bool fourier = !b.is_cheb();
  const bool real_to_cplx = fourier && (kind == kForwardOrtho || kind == kForward);
  const bool cplx_to_real = fourier && (kind == kBackwardOrtho || kind == kBackward || kind == kDiffBackward);
  RPDE_REQUIRE(in.rows == li && out.rows == lo && in.cols == out.cols, "shape mismatch in axis-0 operator");
  const int ncols = in.cols;
  scr_.enter(st);
  Arr2 &tin = scr_.get(0, ncols, li, in.elem), &tout = scr_.get(1, ncols, lo, out.elem);
  launch_transpose(in.p(), in.ld, tin.p(), tin.ld, in.rows, in.cols, in.elem, st);
  if (fourier) {
    // lines are genuinely complex (or real <-> complex): one component, element stride 1
    ProgramBuilder pb(1, ax.slot_len, ncols, 1);
    pb.set_fft(ax);
    const int ai = pb.arr(tin.p(), tin.ld), ao = pb.arr(tout.p(), tout.ld);
    pb.load(0, ai, li * in.elem);
    if (real_to_cplx) pb.rfft_f(0, b.n);
    else if (cplx_to_real) { if (kind == kDiffBackward && order > 0) pb.cik(0, 0, b.m, 1.0 / scale, order); pb.rfft_b(0, b.n); }
    else if (kind == kDiff) { if (order > 0) pb.cik(0, 0, b.m, 1.0 / scale, order); }
    else if (kind == kDiagSolve) pb.tabdiv(0, 0, 2 * b.m, diag, 1);
    else if (kind == kToOrtho || kind == kFromOrtho) {}
    else fail("operator not defined for a Fourier axis");
    pb.store(0, ao, lo * out.elem);
    pb.run(st);
  } else {
    RPDE_REQUIRE(in.elem == out.elem, "element type mismatch");
    run_lines(kind, ax, tin.p(), tin.ld, li, tout.p(), tout.ld, lo, ncols, in.elem, st, order, scale,
              fd, diag, pd);
  }
  launch_transpose(tout.p(), tout.ld, out.p(), out.ld, tout.rows, tout.cols, out.elem, st);
}

void Space2Ops::forward(const Arr2& v, Arr2& vhat, Stream& st) {
  RPDE_REQUIRE(v.rows == phys_rows() && v.cols == phys_cols() && v.elem == 1, "forward: bad input shape");
  RPDE_REQUIRE(vhat.rows == spec_rows() && vhat.cols == spec_cols() && vhat.elem == elem(),
               "forward: bad output shape");
  scr_.enter(st);
  Arr2& t = scr_.get(2, phys_rows(), spec_cols(), 1);
  apply_axis(kForward, 1, v, t, st);
  apply_axis(kForward, 0, t, vhat, st);
}
void Space2Ops::backward(const Arr2& vhat, Arr2& v, Stream& st) {
  RPDE_REQUIRE(v.rows == phys_rows() && v.cols == phys_cols() && v.elem == 1, "backward: bad output shape");
  RPDE_REQUIRE(vhat.rows == spec_rows() && vhat.cols == spec_cols() && vhat.elem == elem(),
               "backward: bad input shape");
  scr_.enter(st);
  Arr2& t = scr_.get(2, phys_rows(), spec_cols(), 1);
  apply_axis(kBackward, 0, vhat, t, st);
  apply_axis(kBackward, 1, t, v, st);
}
void Space2Ops::to_ortho(const Arr2& vhat, Arr2& out, Stream& st) {
  RPDE_REQUIRE(out.rows == ortho_rows() && out.cols == ortho_cols() && out.elem == elem(),
               "to_ortho: bad output shape");
  scr_.enter(st);
  Arr2& t = scr_.get(2, ortho_rows(), spec_cols(), elem());
  apply_axis(kToOrtho, 0, vhat, t, st);
  apply_axis(kToOrtho, 1, t, out, st);
}
void Space2Ops::from_ortho(const Arr2& in, Arr2& vhat, Stream& st) {
  RPDE_REQUIRE(in.rows == ortho_rows() && in.cols == ortho_cols() && in.elem == elem(),
               "from_ortho: bad input shape");
  scr_.enter(st);
  Arr2& t = scr_.get(2, spec_rows(), ortho_cols(), elem());
  apply_axis(kFromOrtho, 0, in, t, st);
  apply_axis(kFromOrtho, 1, t, vhat, st);
}
void Space2Ops::gradient(const Arr2& vhat, int d0, int d1, double s0, double s1, Arr2& out, Stream& st) {
  RPDE_REQUIRE(out.rows == ortho_rows() && out.cols == ortho_cols() && out.elem == elem(),
               "gradient: bad output shape");
  scr_.enter(st);
  Arr2& t = scr_.get(2, ortho_rows(), spec_cols(), elem());
  apply_axis(kDiff, 0, vhat, t, st, d0, s0);
  apply_axis(kDiff, 1, t, out, st, d1, s1);
}

void Space2Ops::gradient_backward(const Arr2& vhat, int d0, int d1, double s0, double s1, Arr2& phys, Stream& st) {
  RPDE_REQUIRE(phys.rows == phys_rows() && phys.cols == phys_cols() && phys.elem == 1, "gradient_backward: bad output shape");
  RPDE_REQUIRE(vhat.rows == spec_rows() && vhat.cols == spec_cols() && vhat.elem == elem(), "gradient_backward: bad input shape");
  scr_.enter(st);
  Arr2& t = scr_.get(2, phys_rows(), spec_cols(), 1);
  apply_axis(kDiffBackward, 0, vhat, t, st, d0, s0);   // axis 0 first: a Fourier axis turns the complex coefficients into real rows
  apply_axis(kDiffBackward, 1, t, phys, st, d1, s1);
}

// ------------------------------------------------------------------------------------------
void ColHhDev::upload(const ColHhHost& h) {
  n = h.n; BR = h.BR; NB = h.NB;
  t0.upload(h.t0); t1.upload(h.t1); t2.upload(h.t2); q1.upload(h.q1); m1.upload(h.m1);
  p2.upload(h.p2); q2.upload(h.q2); r2.upload(h.r2); m2.upload(h.m2); g.upload(h.g);
  if (!h.w.empty()) { w.upload(h.w); hr.upload(h.h); }
  if (!h.rk.empty()) rk.upload(h.rk);
}

void ColHhDev::upload1(const ColHh1Host& h) {
  W = h.W; NSB = h.NSB;
  F.upload(h.F); H0.upload(h.H0); H1.upload(h.H1); m1w.upload(h.m1w); m2w.upload(h.m2w); gw.upload(h.gw);
}

HholtzAdiOp::HholtzAdiOp(Space2Ops& s, double c0, double c1) : sp(s) {
  const double c[2] = {c0, c1};
  for (int axis = 0; axis < 2; ++axis) {
    const Base& b = sp.base(axis);
    if (b.is_cheb()) {
      RPDE_REQUIRE(b.is_composite(), "HholtzAdi: orthonormal Chebyshev base is not supported");
      if (b.kind == kChebDirichletNeumann) {   // BaseKind::ChebDirichletNeumann => PdmaPlus2 (hholtz_adi.rs:62-64)
        pdma[axis].upload(pdma_factor(bands7_axpy(hholtz7_mat_a(b), -c[axis], hholtz7_mat_b(b))));
        continue;
      }
      Bands mtx = bands_axpy(hholtz_mat_a(b), -c[axis], hholtz_mat_b(b));
      fdma_sweep(mtx);
      host[axis] = fdma_tables(mtx);
      fdma[axis] = upload_fdma(host[axis], sp.axis(axis).slot_len);
    } else {
      Vec d(b.m);
      for (int k = 0; k < b.m; ++k) d[k] = 1.0 - (-(double)k * (double)k) * c[axis];
      diag0.upload(d);
    }
  }
}

void HholtzAdiOp::solve(const Arr2& in, Arr2& out, Stream& st) {
  const int e = sp.elem();
  RPDE_REQUIRE(in.rows == sp.ortho_rows() && in.cols == sp.ortho_cols() && in.elem == e,
               "HholtzAdi: input must have the orthonormal shape");
  RPDE_REQUIRE(out.rows == sp.spec_rows() && out.cols == sp.spec_cols() && out.elem == e,
               "HholtzAdi: output must have the composite shape");
  const bool cheb0 = sp.base(0).is_cheb();
  scr_.enter(st);
  Arr2 &t0 = scr_.get(0, sp.spec_rows(), sp.ortho_cols(), e), &t1 = scr_.get(1, sp.spec_rows(), sp.spec_cols(), e),
       &t2 = scr_.get(2, sp.spec_rows(), sp.spec_cols(), e);
  const Arr2* cur = &in;
  if (cheb0) { sp.apply_axis(Space2Ops::kPinvMatvec, 0, in, t0, st); cur = &t0; }
  sp.apply_axis(Space2Ops::kPinvMatvec, 1, *cur, t1, st);
  if (cheb0) sp.apply_axis(Space2Ops::kFdmaSolve, 0, t1, t2, st, 0, 1.0, &fdma[0], nullptr, &pdma[0]);
  else sp.apply_axis(Space2Ops::kDiagSolve, 0, t1, t2, st, 0, 1.0, nullptr, diag0.p);
  sp.apply_axis(Space2Ops::kFdmaSolve, 1, t2, out, st, 0, 1.0, &fdma[1], nullptr, &pdma[1]);
}

// ------------------------------------------------------------------------------------------
static Arr2 upload_dense(const double* src, int rows, int cols) {
  Arr2 a(rows, cols, 1);
  dev_upload2d(a.p(), a.ld, src, rows, cols);
  return a;
}

static thread_local const Vec* g_pending_x_spectrum = nullptr;
void set_pending_x_spectrum(const Vec* lam) { g_pending_x_spectrum = lam; }

PoissonOp::PoissonOp(Space2Ops& s, double c0, double c1, int row_begin, int row_end, double alpha, bool singular_fix) : sp(s) {
  const Base& b0 = sp.base(0);
  const Base& b1 = sp.base(1);
  RPDE_REQUIRE(b1.is_two_term(), "Poisson: axis 1 must be a composite Chebyshev base with a two-term stencil");
  const int m0 = b0.m, m1 = b1.m;
  if (b0.is_cheb()) {
    RPDE_REQUIRE(b0.is_two_term(), "Poisson: axis 0 must be composite Chebyshev (two-term stencil) or Fourier");
    Bands ax = bands_axpy(Bands{Vec(m0, 0.0), Vec(m0, 0.0), Vec(m0, 0.0), Vec(m0, 0.0)}, c0,
                          hholtz_mat_b(b0));
    Bands cx = hholtz_mat_a(b0);
    const Vec* given = (alpha == 0.0 && singular_fix) ? g_pending_x_spectrum : nullptr;
    if (given) RPDE_REQUIRE((int)given->size() == m0, "the supplied x spectrum must hold nx - 2 eigenvalues ([even block | odd block])");
    from_spectrum = given != nullptr;
    EigenX eg = given ? eigenbasis_from_spectrum(ax, cx, *given) : eigen_decomposition_parity(ax, cx);
    me = eg.me; mo = eg.mo;
    lam = eg.lam;
    fwd_e = upload_dense(eg.fwd.data(), me, me);
    bwd_e = upload_dense(eg.bwd.data(), me, me);
    fwd_o = upload_dense(eg.fwd.data() + (size_t)me * me, mo, mo);
    bwd_o = upload_dense(eg.bwd.data() + (size_t)me * me, mo, mo);
  } else {
    lam.resize(m0);
    for (int k = 0; k < m0; ++k) lam[k] = -((double)k * (double)k) * c0;
  }
  half = (me + 1) & ~1;
  lam_raw = lam;
  // singularity fix (src/solver/poisson.rs:84-87): lam[0] of the descending list is the largest
  const double lmax = *std::max_element(lam.begin(), lam.end());
  if (singular_fix && std::fabs(lmax) < 1e-10)
    for (double& l : lam) l -= 1e-10;
  // per-row factorised y systems (A_y + lam_r C_y)
  const Bands ay = bands_axpy(Bands{Vec(m1, 0.0), Vec(m1, 0.0), Vec(m1, 0.0), Vec(m1, 0.0)}, c1,
                              hholtz_mat_b(b1));
  const Bands cy = hholtz_mat_a(b1);
  const LineClass lc = line_class_for(sp.axis(1).slot_len);
  const long ld = (long)lc.T * lc.C;   // one chunk-major table row per x-row
  const int rb = std::max(0, row_begin), re = (row_end < 0 || row_end > m0) ? m0 : row_end;
  const size_t nr = (size_t)std::max(0, re - rb);
  Vec q1(nr * ld, 0.0), p2(nr * ld, 0.0), q2(nr * ld, 0.0), r2(nr * ld, 0.0);
  for (int r = rb; r < re; ++r) {
    Bands mtx = bands_axpy(ay, lam[r] + alpha, cy);   // (A_y + (lam_i + alpha) C_y), fdma_tensor.rs:219-221
    fdma_sweep(mtx);
    FdmaTables t = fdma_tables(mtx);
    const Vec a = chunk_major(t.q1, lc, +1), b = chunk_major(t.p2, lc, -1, 1.0),
              c = chunk_major(t.q2, lc, -1), d = chunk_major(t.r2, lc, -1);
    std::copy(a.begin(), a.end(), q1.begin() + (size_t)(r - rb) * ld);
    std::copy(b.begin(), b.end(), p2.begin() + (size_t)(r - rb) * ld);
    std::copy(c.begin(), c.end(), q2.begin() + (size_t)(r - rb) * ld);
    std::copy(d.begin(), d.end(), r2.begin() + (size_t)(r - rb) * ld);
  }
  rows.row0 = rb;
  rows.n = m1;
  rows.tabld = ld;
  rows.q1.upload(q1); rows.p2.upload(p2); rows.q2.upload(q2); rows.r2.upload(r2);
  rows_c1_ = c1; rows_alpha_ = alpha; rows_rb_ = rb; rows_re_ = re;
}

// The same factors a second time, chunk-major for 16 elements per thread: only the whole-line form of S6 reads them
// (Navier2DEngine::add_prow_line), so only that caller builds them -- 4 x (nx - 2) x 4096 doubles = 0.54 GB at 4097^2 that the
// tensor Helmholtz operators of the adjoint solver, the generic operator API and RPDE_S6_LINE=0 never carried a use for.
bool PoissonOp::ensure_rows16(bool derive) {
  const Base& b1 = sp.base(1);
  const int m1 = b1.m, N16 = m1 + 1;
  // (a Fourier x axis too: the rows of the periodic step's S6 are the wavenumbers)
#ifdef RPDE_EMU
  const bool want16 = N16 == 256 || N16 == 1024 || N16 == 2048 || N16 == 4096;
#else
  const bool want16 = N16 == 1024 || N16 == 2048 || N16 == 4096;
#endif
  const int rb = rows_rb_, re = rows_re_;
  if (!want16 || re <= rb) return false;
  const bool need_p2 = rows16.n == 0, need_full = !derive && !rows16_full_, need_der = derive && !rows16d.built;
  if (!need_p2 && !need_full && !need_der) return true;
  const Bands ay = bands_axpy(Bands{Vec(m1, 0.0), Vec(m1, 0.0), Vec(m1, 0.0), Vec(m1, 0.0)}, rows_c1_, hholtz_mat_b(b1));
  const Bands cy = hholtz_mat_a(b1);
  const size_t nr = (size_t)(re - rb);
  const long ld16 = N16;
  const int T16 = N16 / 16;
  if (need_p2 || need_full) {
    Vec q1w, q2w, r2w, p2w;
    if (need_p2) p2w.assign(nr * ld16, 0.0);
    if (need_full) { q1w.assign(nr * ld16, 0.0); q2w.assign(nr * ld16, 0.0); r2w.assign(nr * ld16, 0.0); }
    for (int r = rb; r < re; ++r) {
      Bands mtx = bands_axpy(ay, lam[r] + rows_alpha_, cy);
      fdma_sweep(mtx);
      FdmaTables t = fdma_tables(mtx);
      if (need_p2) { const Vec bw = chunk_major16(t.p2, T16, -1, 1.0); std::copy(bw.begin(), bw.end(), p2w.begin() + (size_t)(r - rb) * ld16); }
      if (need_full) {
        const Vec aw = chunk_major16(t.q1, T16, +1), cw = chunk_major16(t.q2, T16, -1), dw = chunk_major16(t.r2, T16, -1);
        std::copy(aw.begin(), aw.end(), q1w.begin() + (size_t)(r - rb) * ld16);
        std::copy(cw.begin(), cw.end(), q2w.begin() + (size_t)(r - rb) * ld16);
        std::copy(dw.begin(), dw.end(), r2w.begin() + (size_t)(r - rb) * ld16);
      }
    }
    rows16.row0 = rb;
    rows16.n = m1;
    rows16.tabld = ld16;
    if (need_p2) rows16.p2.upload(p2w);
    if (need_full) { rows16.q1.upload(q1w); rows16.q2.upload(q2w); rows16.r2.upload(r2w); rows16_full_ = true; }
  }
  if (need_der) {
    // the row's matrix is ay + mu cy with ay = c1 (peye . S): nothing below the diagonal, nothing two above (prow_line.h DERIVE)
    for (int k = 0; k < m1; ++k) RPDE_REQUIRE(ay.low[k] == 0.0 && ay.up2[k] == 0.0, "Poisson rows: the band structure the derived factors assume");
    Vec mu(nr), sh(m1, 0.0);
    for (int r = rb; r < re; ++r) mu[r - rb] = lam[r] + rows_alpha_;
    for (int k = 2; k < m1; ++k) sh[k] = cy.up2[k - 2];
    rows16d.mu.upload(mu);
    rows16d.aLa.upload(chunk_major16(cy.low, T16, +1));
    rows16d.aLd.upload(chunk_major16(cy.low, T16, -1));
    rows16d.aU1d.upload(chunk_major16(cy.up1, T16, -1));
    rows16d.aU2d.upload(chunk_major16(cy.up2, T16, -1));
    rows16d.aU2sd.upload(chunk_major16(sh, T16, -1));
    rows16d.b1d.upload(chunk_major16(ay.up1, T16, -1));
    rows16d.built = true;
  }
  return true;
}

void PoissonOp::export_eigenbasis(double* lam_out, double* fwd_out, double* bwd_out) const {
  RPDE_REQUIRE(sp.base(0).is_cheb(), "the Fourier x axis is diagonal: no eigenbasis");
  const int m = me + mo;
  std::copy(lam_raw.begin(), lam_raw.end(), lam_out);
  std::fill(fwd_out, fwd_out + (size_t)m * m, 0.0);
  std::fill(bwd_out, bwd_out + (size_t)m * m, 0.0);
  for (int par = 0; par < 2; ++par) {
    const int mb = par ? mo : me, off = par ? me : 0;
    const Arr2& f = par ? fwd_o : fwd_e;
    const Arr2& b = par ? bwd_o : bwd_e;
    Vec hf((size_t)mb * mb), hb((size_t)mb * mb);
    dev_download2d(hf.data(), f.p(), f.ld, mb, mb);
    dev_download2d(hb.data(), b.p(), b.ld, mb, mb);
    for (int k = 0; k < mb; ++k)
      for (int i = 0; i < mb; ++i) {
        fwd_out[(size_t)(off + k) * m + (par + 2 * i)] = hf[(size_t)k * mb + i];   // fwd[eigen k, coefficient]
        bwd_out[(size_t)(par + 2 * i) * m + (off + k)] = hb[(size_t)i * mb + k];   // bwd[coefficient, eigen k]
      }
  }
}

void PoissonOp::solve(const Arr2& in, Arr2& out, Stream& st) {
  const int e = sp.elem();
  RPDE_REQUIRE(in.rows == sp.ortho_rows() && in.cols == sp.ortho_cols() && in.elem == e,
               "Poisson: input must have the orthonormal shape");
  RPDE_REQUIRE(out.rows == sp.spec_rows() && out.cols == sp.spec_cols() && out.elem == e,
               "Poisson: output must have the composite shape");
  const bool cheb0 = sp.base(0).is_cheb();
  const int m0 = sp.spec_rows(), m1 = sp.spec_cols();
  scr_.enter(st);
  Arr2 &t0 = scr_.get(0, m0, sp.ortho_cols(), e), &t1 = scr_.get(1, m0, m1, e), &t2 = scr_.get(2, m0, m1, e), &t3 = scr_.get(3, m0, m1, e);
  const Arr2* cur = &in;
  if (cheb0) { sp.apply_axis(Space2Ops::kPinvMatvec, 0, in, t0, st); cur = &t0; }
  sp.apply_axis(Space2Ops::kPinvMatvec, 1, *cur, t1, st);
  if (cheb0) {
    // ghat[k, :] = sum_i fwd[k, i] rhs[i, :]  per parity block (rows of one parity: stride 2 ld)
    // (both parity blocks of a transform in ONE launch, like the step's G1 / G2: twice the tiles to fill the chip with)
    launch_gemm_pair(true, GemmProblem{me, m1, me, fwd_e.p(), fwd_e.ld, t1.p(), 2 * t1.ld, t2.p(), t2.ld},
                     GemmProblem{mo, m1, mo, fwd_o.p(), fwd_o.ld, t1.p() + t1.ld, 2 * t1.ld, t2.p() + (size_t)me * t2.ld, t2.ld}, st);
    sp.apply_axis(Space2Ops::kFdmaSolve, 1, t2, t3, st, 0, 1.0, &rows);
    launch_gemm_pair(true, GemmProblem{me, m1, me, bwd_e.p(), bwd_e.ld, t3.p(), t3.ld, out.p(), 2 * out.ld},
                     GemmProblem{mo, m1, mo, bwd_o.p(), bwd_o.ld, t3.p() + (size_t)me * t3.ld, t3.ld, out.p() + out.ld, 2 * out.ld}, st);
  } else {
    sp.apply_axis(Space2Ops::kFdmaSolve, 1, t1, out, st, 0, 1.0, &rows);
  }
}

}  // namespace rpde
