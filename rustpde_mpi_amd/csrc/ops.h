// Operator layer: device tables per basis, a builder for line programs, and the generic
// (canonical row-major "XY" layout, like the reference's ndarray) spectral operators
// forward / backward / to_ortho / from_ortho / gradient and the HholtzAdi / Poisson solvers.
// The generic operators mirror funspace's Space2 methods one-to-one (src/field.rs:103-129) and
// the `Solve` trait (src/solver.rs:59-82): out-of-place, shapes checked, errors thrown.
// The fused time step (engine.cc) builds its own, longer programs from the same pieces.
#pragma once
#include <memory>
#include <vector>

#include "hostmath.h"
#include "kernels.h"

namespace rpde {

inline long pitch(long n) { return (n + 15) & ~15L; }

// factorised seven-diagonal system on the device (hostmath.h PdmaTables, pdma.h)
struct PdmaDev {
  DBuf l2, ka, imu, al, be, ga, de;
  DBuf phi1, phi2, fm, psi1, psi2, psi3, psi4, bm;   // the blocked column form (pdma.h PdmaBlkTabs), uploaded by upload_blocks
  int n = 0, NB = 0;
  PdmaTables host;                                   // kept for upload_blocks
  void upload(const PdmaTables& t);
  void upload_blocks();
  PdmaTabs tabs() const { return PdmaTabs{l2.p, ka.p, imu.p, al.p, be.p, ga.p, de.p, n}; }
  PdmaBlkTabs blk() const { return PdmaBlkTabs{phi1.p, phi2.p, fm.p, psi1.p, psi2.p, psi3.p, psi4.p, bm.p, NB}; }
};

// device-resident tables of one 1-D basis
struct AxisTables {
  Base base{};
  int fft_n = 0;       // complex FFT length of the power-of-two plans (Chebyshev n = 2^k + 1, Fourier nx = 2^k); 0: none
  int blu_m = 0;       // > 0: every other length -- Bluestein's algorithm through FFTs of this power-of-two length (line_vm.h);
                       // fft_n = blu_m = 0: the direct O(n^2) DCT (RPDE_DCT_DIRECT=1, n <= 500: a cross-check, not a product path)
  DBuf blu;            // bluestein_*_tables (chirp, filter spectra); tw = the twiddles of the length-blu_m FFT then
  int slot_len = 0;    // slot length for lines of this axis (doubles)
  DBuf tw, tw2, fwd_post, bwd_pre;             // transforms
  DBuf low;                                     // stencil (composite): S[k+2,k]
  DBuf low1;                                    // S[k+1,k]: the three-term stencil of cheb_dirichlet_neumann only
  PdmaDev fo_pdma;                              // its from_ortho: the factorised normal equations S^T S
  std::unique_ptr<AxisTables> ortho;            // its orthonormal parent (same n): transforms, derivatives and B2 rows of a
                                                // three-term axis run as line programs of the parent around the pdma.h kernels
  DBuf fo_t0, fo_t1, fo_t2, fo_pup, fo_qup, fo_qdn;  // from_ortho
  DBuf pv0, pv1, pv2;                           // B2 pseudo-inverse rows (Chebyshev family)
  explicit AxisTables(const Base& b);
  int n_phys() const { return base.n; }
  int n_ortho() const { return base.n_ortho(); }   // complex count for Fourier
  int n_spec() const { return base.m; }
  int ortho_doubles() const { return base.is_cheb() ? base.n : 2 * base.m; }
  int spec_doubles() const { return base.is_cheb() ? base.m : 2 * base.m; }
};

// swept Fdma solve tables on device: constant (tabld = 0) or one row of tables per line
struct FdmaDev {
  DBuf q1, p2, q2, r2;
  long row0 = 0;   // per-line tables: global index of the first line they hold (pencil-sharded engines keep their own rows only)
  long tabld = 0;
  int n = 0;
};
FdmaDev upload_fdma(const FdmaTables& t, int slot_len);

// ------------------------------------------------------------------------------------------
class ProgramBuilder {
 public:
  Program pg{};
  ProgramBuilder(int nslots, int slot_len, int nlines, int ncomp = 1);
  void set_fft(const AxisTables& ax);
  int arr(double* p, long ld, int es = 1, long coff = 0);
  int arr(const double* p, long ld, int es = 1, long coff = 0) {
    return arr(const_cast<double*>(p), ld, es, coff);
  }
  int tab(const double* t);
  // ops (slot indices d, a, b)
  void load(int d, int arr, int n, double s0 = 1.0, bool acc = false, int interleave_half = 0);
  void loadmul(int d, int arr, int n, double s0 = 1.0);   // d[k] *= s0 * A[line][k]
  // the last two ops (load / loadmul of contiguous lines; the first may be a loadx) run as ONE
  // op: both lines' global loads are in flight together, one HBM round trip instead of two
  void pair_last_loads();
  // interleaved complex line (n doubles) times i*kappa while loading: d (+)= s0 * (i kappa) * A
  void load_cik(int d, int arr, int n, double s0 = 1.0, bool acc = false);
  void set_line0(int line0) { pg.line0 = line0; }
  void loadx(int d, int arr, int n, int rows, const double* lowtab, double s0 = 1.0, bool acc = false, int deinterleave_half = 0);
  void store(int a, int arr, int n, double s0 = 1.0, int deinterleave_half = 0);
  // the last op (a plain OP_STORE) raises *flag when it stores a NaN: the device-side form of
  // Integrate::exit (navier.rs:482-489) -- no extra pass over the fields, no allocation
  void guard_last_store(int* flag);
  void sten(int d, int a, int n_ortho, const double* low);
  void mv3(int d, int a, int n, const double* t0, const double* t1, const double* t2, long tabld = 0);
  void cdiff(int d, int a, int n, double scale);
  void rec1(int d, int a, int n, const double* p, const double* q, int dir, long tabld = 0);
  void rec2(int d, int a, int n, const double* p, const double* q, const double* r, long tabld = 0);
  // `pre` / `post`: nullptr or the STANDARD scalings of the axis given to set_fft (bwd_pre /
  // fwd_post, the latter possibly zeroed from index `cut` on -- the 2/3 rule): on the FFT path the
  // kernel evaluates them arithmetically ((-1)^k, 1/2, 1/N, the cut) instead of fetching tables;
  // the tables themselves are only read by the direct O(n^2) transform of small lines.
  void dct(int d, int n, const double* pre, const double* post, int cut = -1);
  // fused forms: `sten` = composite->ortho stencil of `ax` applied while packing (slot d holds the
  // composite coefficients); `store_arr` >= 0 = results written straight to that array (nstore
  // values, scaled).  Fall back to separate ops when the line uses the direct (non-FFT) transform.
  void dct_fused(int d, const AxisTables& ax, bool sten, const double* pre, const double* post,
                 int store_arr = -1, int nstore = 0, double scale = 1.0, int cut = -1);
  void mul(int d, int a, int b, int n, double s0 = 1.0, bool acc = false);
  void axpby(int d, int a, double s0, int b, double s1, int n);
  // per-thread register copy of slot a (kept until the end of the program), and
  // d = s0 * d + s1 * stash: an accumulator that costs no LDS slot
  void stash(int a);
  void unstash_axpy(int d, double s0, double s1, int n);
  void zero(int d, int from, int to);
  void tabdiv(int d, int a, int n, const double* t, int shift);
  void rfft_f(int d, int nx);
  void rfft_b(int d, int nx);
  void cik(int d, int a, int ncomplex, double s0, int power);
  // composites
  void to_ortho(int d, const AxisTables& ax);                 // slot d: composite -> ortho (in place)
  void to_ortho_axpby(int d, double sd, int a, double sa, const AxisTables& ax);   // d = sd * d + sa * (composite -> ortho of slot a), d != a
  void to_ortho_from(int d, int a, const AxisTables& ax);     // slot d = composite -> ortho of slot a (d != a: one phase)
  void from_ortho(int d, const AxisTables& ax);               // slot d: ortho -> composite (in place)
  void fdma_solve(int d, int n, const FdmaDev& f);            // slot d in place
  void pinv_matvec(int d, const AxisTables& ax);              // slot d: ortho (n) -> (n-2), in place
  void run(Stream& st) { launch_line_program(pg, st); }

 private:
  Op& push(int code);
  int narr_ = 0, ntab_ = 0;
  const AxisTables* ax_ = nullptr;   // axis of set_fft
  void dct_flags(Op& o, int n, const double* pre, const double* post, int cut);
};

// ------------------------------------------------------------------------------------------
// a 2-D device array in canonical (XY) layout: rows x cols elements of `elem` doubles, pitch ld doubles
struct Arr2 {
  DBuf buf;
  int rows = 0, cols = 0, elem = 1;
  long ld = 0;
  Arr2() = default;
  Arr2(int r, int c, int e = 1) { alloc(r, c, e); }
  void alloc(int r, int c, int e = 1) {
    rows = r; cols = c; elem = e; ld = pitch((long)c * e);
    buf.alloc((size_t)r * ld);
  }
  double* p() const { return buf.p; }
  size_t bytes() const { return (size_t)rows * ld * sizeof(double); }
};

// Work arrays an operator keeps between calls (grow-only): the generic operators allocate nothing and wait for nothing once
// warm, so a solver composed of them (adjoint.cc: several hundred launches per update) queues its whole step without a host
// round trip.  All calls of one operator are expected on ONE stream; a call on another stream first waits for the previous one.
class ScratchPool {
 public:
  Arr2& get(int slot, int rows, int cols, int elem) {
    RPDE_REQUIRE(slot >= 0 && slot < kSlots, "scratch slot");
    Arr2& a = a_[slot];
    const long ld = pitch((long)cols * elem);
    const size_t need = (size_t)rows * ld;
    if (a.buf.n < need) { wait(); a.buf.alloc(need); }
    a.rows = rows; a.cols = cols; a.elem = elem; a.ld = ld;
    return a;
  }
  void enter(Stream& st) {
#ifndef RPDE_EMU
    if (used_ && last_ != st.s) (void)hipStreamSynchronize(last_);
    last_ = st.s; used_ = true;
#else
    (void)st;
#endif
  }
 private:
  static constexpr int kSlots = 6;
  Arr2 a_[kSlots];
  void wait() {
#ifndef RPDE_EMU
    if (used_) (void)hipStreamSynchronize(last_);   // the buffer that is about to be replaced may still be read
#endif
  }
#ifndef RPDE_EMU
  hipStream_t last_ = nullptr;
#endif
  bool used_ = false;
};

// funspace Space2<B0, B1> on device (canonical layout)
class Space2Ops {
 public:
  Space2Ops(const Base& b0, const Base& b1);
  const Base& base(int axis) const { return ax_[axis]->base; }
  AxisTables& axis(int a) { return *ax_[a]; }
  bool complex_spectral() const { return !ax_[0]->base.is_cheb(); }
  int elem() const { return complex_spectral() ? 2 : 1; }
  // shapes (rows, cols) in elements
  int phys_rows() const { return ax_[0]->base.n; }
  int phys_cols() const { return ax_[1]->base.n; }
  int spec_rows() const { return ax_[0]->base.m; }
  int spec_cols() const { return ax_[1]->base.m; }
  int ortho_rows() const { return ax_[0]->base.n_ortho(); }
  int ortho_cols() const { return ax_[1]->base.n; }

  void forward(const Arr2& v, Arr2& vhat, Stream& st);
  void backward(const Arr2& vhat, Arr2& v, Stream& st);
  void to_ortho(const Arr2& vhat, Arr2& out, Stream& st);
  void from_ortho(const Arr2& in, Arr2& vhat, Stream& st);
  void gradient(const Arr2& vhat, int d0, int d1, double s0, double s1, Arr2& out, Stream& st);
  // backward_ortho(gradient(vhat, [d0, d1], [s0, s1])) -- what `conv_term` needs of a field (functions.rs:56-69): the two
  // factors of an axis are one line program (stencil, derivative recurrence, inverse transform), so the pair costs two
  // line programs and two transposes instead of four and four
  void gradient_backward(const Arr2& vhat, int d0, int d1, double s0, double s1, Arr2& phys, Stream& st);

  // single-axis building blocks on canonical arrays (axis 0 goes through a transposed copy)
  enum Kind { kToOrtho, kFromOrtho, kForwardOrtho, kBackwardOrtho, kForward, kBackward, kDiff,
              kPinvMatvec, kFdmaSolve, kDiagSolve,
              kDiffBackward };   // derivative (order, scale) and backward transform of a composite line in one line program
  // kFdmaSolve along a three-term axis takes `pd` (PdmaPlus2) instead of `fd`
  void apply_axis(Kind kind, int axis, const Arr2& in, Arr2& out, Stream& st, int order = 0,
                  double scale = 1.0, const FdmaDev* fd = nullptr, const double* diag = nullptr,
                  const PdmaDev* pd = nullptr);

 private:
  std::unique_ptr<AxisTables> ax_[2];
  ScratchPool scr_;   // 0, 1: the transposed copies of an axis-0 operator; 2: between the two axes of a 2-D operator; 3: three-term axis
  void run_lines(Kind kind, const AxisTables& ax, const double* in, long ldi, int len_in,
                 double* out, long ldo, int len_out, int nlines, int ncomp, Stream& st, int order,
                 double scale, const FdmaDev* fd, const double* diag, const PdmaDev* pd = nullptr);
  void run_lines3(Kind kind, const AxisTables& ax, const double* in, long ldi, int len_in,
                  double* out, long ldo, int len_out, int nlines, int ncomp, Stream& st, int order,
                  double scale, const PdmaDev* pd);
};

// device tables of the column-scan form of one Helmholtz-y solve (colscan.h)
struct ColHhDev {
  DBuf t0, t1, t2, q1, m1, p2, q2, r2, m2, g, w, hr, rk;
  DBuf F, H0, H1, m1w, m2w, gw;   // single-pass form (colscan1.h), uploaded by upload1
  int n = 0, BR = 0, NB = 0, W = 0, NSB = 0;
  void upload(const ColHhHost& h);
  void upload1(const ColHh1Host& h);
  ColHh1Tabs tabs1() const { return ColHh1Tabs{F.p, H0.p, H1.p, m1w.p, m2w.p, gw.p}; }
  ColHhTabs tabs() const { return ColHhTabs{t0.p, t1.p, t2.p, q1.p, m1.p, p2.p, q2.p, r2.p, m2.p, g.p, w.p, hr.p, rk.p}; }
};
constexpr int kColBlockRows = kColBR; // rows per block of the column scans (colscan.h)

// HholtzAdi (src/solver/hholtz_adi.rs:48-76,149-169) on canonical arrays
class HholtzAdiOp {
 public:
  HholtzAdiOp(Space2Ops& sp, double c0, double c1);
  void solve(const Arr2& in_ortho, Arr2& out, Stream& st);
  FdmaDev fdma[2];      // Chebyshev axes with a two-term stencil
  FdmaTables host[2];   // the same tables on the host, natural order (other kernels re-order them for their own chunking)
  PdmaDev pdma[2];      // Chebyshev axes with the three-term stencil: PdmaPlus2 (hholtz_adi.rs:62-64)
  DBuf diag0;           // Fourier axis 0: 1 + c0 k^2
  Space2Ops& sp;
 private:
  ScratchPool scr_;
};

// Setup data supplied by the host: the x eigenvalues of the next PLAIN Poisson solver built on this thread (alpha = 0; the
// tensor Helmholtz solvers ignore it).  With it PoissonOp builds its eigenbasis with hostmath's eigenbasis_from_spectrum
// (bit-reproducible, no LAPACK) instead of dgeev.  Null clears.  The C ABI sets it around the construction of one engine
// (rpde_navier2d_create_confined_with_spectrum) or operator (rpde_poisson_create_with_spectrum).
void set_pending_x_spectrum(const Vec* lam);

// Poisson (src/solver/poisson.rs:54-94,195-236) on canonical arrays
class PoissonOp {
 public:
  // row_begin / row_end: the x-rows (eigen index, or wavenumber when periodic) whose factorised y-systems
  // are kept on this device; the default keeps all of them.  A pencil-sharded rank only solves its own rows.
  // alpha / singular_fix: the same tensor construction serves `Hholtz` (src/solver/hholtz.rs:72-106:
  // (I - c D2) vhat = A f, i.e. laplacian = -c mat_b, alpha = 1, no singularity shift) -- TensorHholtzOp below
  PoissonOp(Space2Ops& sp, double c0, double c1, int row_begin = 0, int row_end = -1, double alpha = 0.0,
            bool singular_fix = true);
  void solve(const Arr2& in_ortho, Arr2& out, Stream& st);
  Space2Ops& sp;
  // x direction
  int me = 0, mo = 0, half = 0;  // parity block sizes; column offset of the odd block in split arrays
  Vec lam;                       // eigenvalues (after the singularity shift), engine order
  Vec lam_raw;                   // the same before the shift (what LAPACK returned)
  bool from_spectrum = false;    // the eigenbasis was built from host-supplied eigenvalues (set_pending_x_spectrum)
  // the x eigen-decomposition in the reference's form (fdma_tensor.rs:123-127): m eigenvalues
  // (unshifted, order [even block | odd block]), fwd = Q^-1 C^-1 and bwd = Q as dense m x m row-major
  // matrices over the natural coefficient index.  Setup data, exported so that a checker can run
  // the same algorithm on the same decomposition (the Poisson solve amplifies the round-off of
  // dgeev itself).  Confined (Chebyshev x) only.
  void export_eigenbasis(double* lam_out, double* fwd_out, double* bwd_out) const;
  Arr2 fwd_e, fwd_o, bwd_e, bwd_o;
  // y direction: per-x-row swept tables, row index = eigen index (confined) or wavenumber (periodic)
  FdmaDev rows;
  // the same factors a second time, chunk-major for 16 elements per thread (rhs_line.h chunk_major16: ascending q1, descending
  // p2 / q2 / r2; one row of ny - 1 = 16 T entries per x-row) -- the whole-line form of the stage (prow_line.h, S6 of the
  // confined step).  Built ON FIRST USE (ensure_rows16) for real (Chebyshev x) operators whose y-lines have a whole-line length; n = 0: not built.
  FdmaDev rows16;
  // `derive` (prow_line.h DERIVE): only p2 is tabulated per row; q1 / q2 / r2 follow in the kernel from p2, mu = lam_r + alpha and
  // the one-dimensional bands of the row's matrix c1 B + mu A below (a quarter of the table memory: 0.13 GB instead of 0.54 at 4097^2)
  struct Rows16Derived { DBuf mu, aLa, aLd, aU1d, aU2d, aU2sd, b1d; bool built = false; } rows16d;
  bool ensure_rows16(bool derive = false);   // builds them on first use (Navier2DEngine::add_prow_line); false: this operator has no such form
 private:
  ScratchPool scr_;
  double rows_c1_ = 0.0, rows_alpha_ = 0.0;
  int rows_rb_ = 0, rows_re_ = 0;
  bool rows16_full_ = false;     // q1 / q2 / r2 of rows16 are built too
};

// Hholtz<f64, 2> (src/solver/hholtz.rs:29-37, 72-106, 164-187): (I - c0 Dxx - c1 Dyy) vhat = A f by diagonalising x --
// FdmaTensor::from_matrix(laplacians = -c mat_b, masses = mat_a, alpha = 1).  solve(): preconditioner along both axes,
// eigen-transform (two parity GEMMs), one banded solve per x-row with pre-factorised rows, back-transform.
class TensorHholtzOp : public PoissonOp {
 public:
  TensorHholtzOp(Space2Ops& sp, double c0, double c1) : PoissonOp(sp, -c0, -c1, 0, -1, 1.0, false) {}
};

}  // namespace rpde
