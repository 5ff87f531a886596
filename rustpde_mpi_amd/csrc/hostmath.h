// Host-side (setup only) mathematics of the function spaces and solver matrices.
// Everything here runs once at construction; results are uploaded as constant tables.
//
// Reference: funspace 0.3.0 bases/matrices as used by rustpde (call sites
// src/field.rs:195-216), Fdma::sweep (src/solver/fdma.rs:73-82), HholtzAdi::new
// (src/solver/hholtz_adi.rs:48-76), Poisson::new (src/solver/poisson.rs:54-94),
// FdmaTensor::from_matrix (src/solver/fdma_tensor.rs:106-154), utils::eig/inv
// (src/solver/utils.rs:67-107).
#pragma once
#include <string>
#include <vector>

namespace rpde {

enum BaseKind : int { kChebyshev = 0, kChebDirichlet = 1, kChebNeumann = 2, kFourierR2c = 3,
                      kChebDirichletNeumann = 4 };   // the "hc" temperature along y (navier.rs:245-248): three-term stencil

using Vec = std::vector<double>;

struct Base {
  BaseKind kind;
  int n;  // physical points
  int m;  // spectral coefficients (complex count for Fourier)
  bool is_cheb() const { return kind != kFourierR2c; }
  bool is_composite() const { return kind == kChebDirichlet || kind == kChebNeumann || kind == kChebDirichletNeumann; }
  // two-term stencil (T_k + low_k T_{k+2}): even and odd coefficients decouple -- what the stride-2 scans, the
  // parity-block GEMMs and the column scans rely on; cheb_dirichlet_neumann has S[k+1,k] != 0 and goes through pdma.h
  bool is_two_term() const { return kind == kChebDirichlet || kind == kChebNeumann; }
  int n_ortho() const { return kind == kFourierR2c ? m : n; }
};
Base make_base(BaseKind kind, int n);

Vec base_coords(const Base& b);                 // grid points (unscaled)
Vec base_dx(const Base& b, const Vec& x);       // src/field.rs:135-163
// Rust's number formatting for the info lines (defined in engine.cc): {:.Ne} without exponent padding; `{}` of an f64
std::string rust_exp(double v, int prec);
std::string rust_display(double v);
Vec stencil_low(const Base& b);                 // S[k+2,k]  (S[k,k] = 1), length m
Vec stencil_low1(const Base& b);                // S[k+1,k]: zero for the two-term stencils, length m

// DCT-I scaling tables for a Chebyshev line of n points (N = n-1)
Vec cheb_fwd_post(int n);   // (-1)^k / N, halved at both ends
Vec cheb_bwd_pre(int n);    // (-1)^k * (1/2 interior, 1 at the ends)

// twiddles
Vec fft_twiddles(int nfft);            // (cos, -sin)(2 pi k / nfft), k < nfft
Vec dct_split_twiddles(int N);         // (cos, sin)(pi k / N), k <= N
Vec rfft_split_twiddles(int nx);       // (cos, sin)(2 pi k / nx), k <= nx/2
Vec dct_direct_costab(int N);          // cos(pi m / N), m < 2N
// Bluestein (chirp-z) transforms of arbitrary length through a power-of-two FFT of length M (hostmath.cc)
int bluestein_len(int lags);                    // smallest power of two >= lags
Vec bluestein_dct_tables(int N, int M);         // chirp (cos, sin)(pi j^2 / (2N)), j <= N | filter spectrum / M  (M >= 2N + 1)
Vec bluestein_rfft_tables(int nx, int M);       // chirp (cos, sin)(pi j^2 / nx), j < nx | forward | backward filter spectra / M  (M >= nx + nx/2)

// from_ortho as MV3 + ascending REC1 + descending REC1 (tables of length m, padded with zeros)
struct FromOrthoTables {
  Vec t0, t1, t2;   // rhs_k = t0 c_k + t1 c_{k+2}
  Vec p_up, q_up;   // g_k = p b_k + q g_{k-2}
  Vec q_dn;         // x_k = g_k + q x_{k+2}
};
FromOrthoTables from_ortho_tables(const Base& b);

// B2 pseudo-inverse rows 2.. as MV3 tables (length m = n-2)
struct Mv3Tables { Vec t0, t1, t2; };
Mv3Tables pinv_tables(const Base& b);

// four-diagonal matrix in row-indexed band form (length m each; out-of-range entries are 0)
struct Bands { Vec low, dia, up1, up2; };
Bands hholtz_mat_a(const Base& b);    // pinv . S
Bands hholtz_mat_b(const Base& b);    // peye . S   (offsets 0, +2 only)
Bands bands_axpy(const Bands& a, double c, const Bands& b);  // a + c b
void fdma_sweep(Bands& m);            // in place, reference op order

// solve tables for a swept Fdma: ascending REC1 (q1) then descending REC2 (p2, q2, r2)
struct FdmaTables { Vec q1, p2, q2, r2; };
FdmaTables fdma_tables(const Bands& swept);

// Seven-diagonal systems (offsets -2 .. +4) of the three-term stencil: `PdmaPlus2` (src/solver/pdma_plus2.rs:45-157).
// Row-indexed bands: d[o + 2][r] = a[r, r + o], zero where the column is out of range.
struct Bands7 { Vec d[7]; };
Bands7 hholtz7_mat_a(const Base& b);                      // pinv . S
Bands7 hholtz7_mat_b(const Base& b);                      // peye . S (offsets 0, +1, +2)
Bands7 bands7_axpy(const Bands7& a, double c, const Bands7& b);
Bands7 from_ortho7(const Base& b);                        // S^T S (offsets -2 .. +2): the normal equations of from_ortho
// the factorisation PdmaPlus2::from_matrix precomputes (same order of operations; 1 / mu instead of mu, and every
// table padded with four zeros so that the four-term backward recurrence needs no end cases): a solve is
//   ze_i = (rhs_i - l2_{i-2} ze_{i-2} - ka_i ze_{i-1}) imu_i ,   x_i = ze_i - al_i x_{i+1} - be_i x_{i+2} - ga_i x_{i+3} - de_i x_{i+4}
struct PdmaTables { int n = 0; Vec l2, ka, imu, al, be, ga, de; };   // l2 is shifted: l2[i] = a[i, i-2]
PdmaTables pdma_factor(const Bands7& m);
// tables of the blocked column form (pdma.h PdmaBlkTabs): homogeneous solutions of the two recurrences restarted at every block
// of BR rows, and the block transfer matrices of the end states
struct PdmaBlockTables { int NB = 0; Vec phi1, phi2, fm, psi1, psi2, psi3, psi4, bm; };
PdmaBlockTables pdma_block_tables(const PdmaTables& t, int BR);

// Tables of the column-scan Helmholtz solve (colscan.h) for one swept Fdma and its B2 preconditioner,
// rows cut into blocks of BR: per-row coefficients (zero-padded to NB * BR + 4), the block transfer
// factors / matrices of the forward (first-order) and backward (second-order) chains, and `g`: the backward
// end state a block reaches from a zero backward inflow per unit of its forward inflow.
struct ColHhHost {
  int n = 0, BR = 0, NB = 0;
  Vec t0, t1, t2, q1, m1, p2, q2, r2, m2, g;
  Vec w, h;   // optional rank-one term (colscan.h): weights of the column sum, response; empty: none
  Vec rk;     // pencil-sharded runs: [nranks][14] transfer of every rank's rows taken as one block; empty: one rank
};
// single-pass form of the column scan (colscan1.h, one rank): per-row responses of a block's solution to its inflow
// states, and the transfers of super-blocks of W blocks
struct ColHh1Host {
  int W = 0, NSB = 0;
  Vec F, H0, H1;      // [rows, padded]: x_j per unit of forward inflow / of the backward inflow states (1,0) and (0,1) of its block
  Vec m1w, m2w, gw;   // [NSB][2], [NSB][2][4], [NSB][2][2]: m1 / m2 / g of W consecutive blocks taken as one
};
ColHh1Host build_colhh1_tables(const ColHhHost& h, int W);
// blocks cover the rows [row0, jend) of the system (this rank's rows; jend < 0: to the end); `ranks`: the row
// partition (nranks + 1 boundaries, all even) when the rows are split over several ranks
ColHhHost build_colhh_tables(const Mv3Tables& pv, const FdmaTables& f, int BR, int row0 = 0, int jend = -1,
                             const std::vector<int>* ranks = nullptr);
// y part of the velocity correction (navier_eq.rs:117-125) as two banded column problems on the pseudo-pressure
// (Neumann-composite rows, base `bn`), results in the Dirichlet-composite base `bd` of the velocities:
//   a = from_ortho_D( to_ortho_N(ps) )                          taps ps_{k-2}, ps_k, ps_{k+2}          (shift 2)
//   b = from_ortho_D( dscale * d/dy to_ortho_N(ps) )            taps ps_{k-1}, ps_{k+1} + rank-one term (shift 1)
// For the Dirichlet stencil (c_k = a_k - a_{k-2}) the right-hand side of the projection, S^T d = d_k - d_{k+2}, is the
// LOCAL term 2 (k + 1) c_{k+1} of the derivative's recurrence d_k = d_{k+2} + 2 (k + 1) c_{k+1}; only row 0 (d_0 is
// halved) keeps a sum over the column: rhs_0 = dscale (c_1 - d_2 / 2) -- the rank-one term.
struct ColCorrHost { ColHhHost a, b; };
ColCorrHost build_colcorr_tables(const Base& bd, const Base& bn, double dscale, int BR, int row0 = 0, int jend = -1,
                                 const std::vector<int>* ranks = nullptr);

// dense helpers (row-major) + LAPACK (loaded at run time from the OpenBLAS that ships with SciPy,
// the same library family the reference links: Cargo.toml:39,45-46)
struct EigenX {
  Vec lam;   // m eigenvalues; index order = [even-parity block | odd-parity block], each descending
  Vec fwd;   // block-diagonal Q^-1 C^-1 : fwd_e (me x me) then fwd_o (mo x mo), row-major
  Vec bwd;   // block-diagonal Q         : bwd_e (me x me) then bwd_o (mo x mo), row-major
  int me, mo;
};
// Diagonalise inv(C) A per parity block (all bands have even offsets, so even and odd
// coefficients decouple exactly).  Throws if LAPACK cannot be loaded.
EigenX eigen_decomposition_parity(const Bands& a, const Bands& c);
std::string lapack_library_path();
// The eigenvalues alone (dgeev without vectors), same order as EigenX::lam.
Vec eigen_spectrum_parity(const Bands& a, const Bands& c);
// The eigenbasis that belongs to GIVEN eigenvalues, without LAPACK and bit-reproducible: per eigenvalue one right and one
// left null vector of the banded pencil A - lam C of a parity block (inverse iteration on a banded LU with partial pivoting:
// O(m) per vector), bwd = the right vectors (unit 2-norm, largest component positive), fwd = the left vectors scaled to
// fwd C bwd = I (the rows of Q^-1 C^-1 ARE the left eigenvectors of the pencil: F A = Lam F C).  Plain scalar loops in a
// fixed order: the same library binary returns the same bits on any host -- which is what lets a checker on another
// machine run the reference algorithm on EXACTLY the engine's setup data (tests/golden/make_shared_basis_golden.py): dgeev's
// own output is not reproducible across thread counts or CPU models, and the Poisson solve amplifies that by 1e10
// (DESIGN.md section 4).  `lam`: m values in EigenX order.
EigenX eigenbasis_from_spectrum(const Bands& a, const Bands& c, const Vec& lam);

}  // namespace rpde
