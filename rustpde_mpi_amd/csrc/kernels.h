// Host-callable launchers of the device kernels (HIP build) or their host emulation (EMU build).
#pragma once
#include "colscan.h"
#include "colscan1.h"
#include "rfft_line.h"
#include "line_vm.h"
#include "pdma.h"
#include "rhs_line.h"
#include "corr_line.h"
#include "div_line.h"
#include "prow_line.h"
#include "pres_line.h"
#include "per_rows.h"

namespace rpde {

// slot length (doubles) needed for lines of `maxlen` doubles (room for the +2/+4 stencil reads)
// and for the padded FFT work area (two consecutive slots hold fft_work_doubles(N) doubles)
inline int slot_len_for(int maxlen) {
  int need = maxlen + 4;
  int n = 1;
  while (2 * n < maxlen) n *= 2;                  // largest FFT that can be asked for on this line
  const int half_work = (fft_work_doubles(n) + 1) / 2;
  if (half_work > need) need = half_work;
  return (need + 1) & ~1;
}

// kernel configuration (threads per line, scan chunk per thread) used for a given slot length
struct LineClass { int T, C; };
LineClass line_class_for(int slot_len);

// Re-order a recurrence table for the chunked scans: out[i * T + t] = tab[tau(t) * C + i] with
// tau(t) = t (dir > 0) or T - 1 - t (dir < 0); entries past the end of `tab` are `pad`.
std::vector<double> chunk_major(const std::vector<double>& tab, LineClass lc, int dir, double pad = 0.0);

// run a line program: grid = (nlines, ncomp), one workgroup per line
void launch_line_program(const Program& pg, Stream& st);

// out[c * ldo + r] = in[r * ldi + c], r < rows, c < cols; elem = 1 (double) or 2 (interleaved complex)
void launch_transpose(const double* in, long ldi, double* out, long ldo, int rows, int cols,
                      int elem, Stream& st);

// up to six arrays of one shape in one launch
constexpr int kMaxTransposeBatch = 6;
struct TransposeBatch { const double* in[kMaxTransposeBatch]; double* out[kMaxTransposeBatch]; };
void launch_transpose_batch(const TransposeBatch& b, int n, long ldi, long ldo, int rows, int cols, int elem, Stream& st);

// C[m, n] = sum_k A[m, k] * B[n, k]   (A: M x K lda, B: N x K ldb, C: M x N ldc), f64
void launch_gemm_nt(int M, int N, int K, const double* A, long lda, const double* B, long ldb,
                    double* C, long ldc, Stream& st);
// C[m, n] = sum_k A[m, k] * B[k, n]   (A: M x K lda, B: K x N ldb, C: M x N ldc), f64
void launch_gemm_nn(int M, int N, int K, const double* A, long lda, const double* B, long ldb,
                    double* C, long ldc, Stream& st);
// two independent products of the same kind (nn: both as launch_gemm_nn, else as launch_gemm_nt) in ONE launch
struct GemmProblem { int M = 0, N = 0, K = 0; const double* A = nullptr; long lda = 0; const double* B = nullptr; long ldb = 0;
                     double* C = nullptr; long ldc = 0;
                     bool ct = false;      // ct: store the product transposed, C[n * ldc + m] (launch_gemm_pair only)
                     bool zero00 = false; };   // the element (m, n) = (0, 0) of the product is stored as 0: `pseu[0, 0] = 0` (solve_pres,
                                               // navier_eq.rs:158-162) rides in the store of the GEMM that produces pseu (launch_gemm_pair only)
void launch_gemm_pair(bool nn, const GemmProblem& p0, const GemmProblem& p1, Stream& st);

// out[r * ldo + c] = in[r * ldi + c], r < rows, c < cols (doubles)
void launch_copy2d(const double* in, long ldi, double* out, long ldo, int rows, int cols, Stream& st);

// pack / unpack of one batched pencil exchange (P ranks, up to 6 arrays), one launch each:
//  pack  : for every destination q and array a, the block in_a[0:rl, c0[q]:c0[q+1]] is written
//          transposed (cq x rl, contiguous) to send + soff[q] + a * cq * rl * elem
//  unpack: for every source s and array a, the segment recv + roff[s] + a * cl * rs * elem
//          (cl x rs, contiguous) is written to out_a[0:cl, r0[s]:r0[s+1]]
struct XchgDesc {
  int P, nA, rl, cl, elem;
  int c0[9], r0[9];
  long soff[9], roff[9];
  long ldi, ldo;
  const double* in[6];
  double* out[6];
};
void launch_xchg_pack(const XchgDesc& d, double* send, Stream& st);
void launch_xchg_unpack(const XchgDesc& d, const double* recv, Stream& st);

// column scans (colscan.h): the Helmholtz solve along y of up to three YX fields (five launches: pass A,
// carry, pass B, carry, pass C) and the Chebyshev y-derivative of one YX array (three launches)
// phase 0: block summaries; 1: carry (one rank: final; sharded: this rank's summary); 2: carry from the gathered
// summaries (sharded only, after the exchange); 3: final pass
void launch_col_hholtz_phase(const ColHhArgs& a, int phase, Stream& st);
void launch_col_diff_phase(const ColDiffArgs& a, int phase, Stream& st);
// single-pass form (colscan1.h): one kernel; every launch site owns its ticket / arrival counters and never resets them
// (64-bit, a launch is an epoch)
void launch_col_hholtz1(const ColHh1Args& a, Stream& st);
// workgroups of that kernel the current device keeps resident at once (occupancy per CU x CUs): the engine's ctor compares it
// with what a tile's partners need before it chooses the single-pass form
int col_hholtz1_resident_workgroups(int W, int NSB);
void launch_col_diff1(const ColDiff1Args& a, Stream& st);   // colscan1.h: the y-derivative in one pass (one rank)
inline void launch_col_hholtz(const ColHhArgs& a, Stream& st) { for (int ph : {0, 1, 3}) launch_col_hholtz_phase(a, ph, st); }   // one rank
inline void launch_col_diff(const ColDiffArgs& a, Stream& st) { for (int ph : {0, 1, 3}) launch_col_diff_phase(a, ph, st); }

// backward Chebyshev transform of whole lines with four workgroups per CU (dct_line.h); false: shape / alignment
// not covered (the caller runs the line program instead)
struct DctLineArgs;
bool launch_dct_line(const DctLineArgs& a, Stream& st, long long* trace = nullptr);   // trace: diagnostics record (kTraceStride words per workgroup)
// two transforms of the same input lines in one launch (value and x-derivative of a state line, S1 of the step):
// the second read of a line comes from L2
bool launch_dct_line2(const DctLineArgs& a0, const DctLineArgs& a1, Stream& st);
// rfft_line.h: the Fourier lines of the periodic step.  false: shape not covered (the caller runs the line programs)
bool launch_rfft_pair(const RfftLineArgs& a0, const RfftLineArgs& a1, Stream& st);   // S1: value and x-derivative of a spectral line
bool launch_four_rhs(const FourRhsArgs& a, Stream& st);                               // S3: forward FFT + right-hand side + diagonal factor
// Small grids (lines of 1025 points, one wave per line): a launch of one field's lines leaves most of the chip idle and
// costs a kernel's latency; the launches of the three fields of a stage go out as ONE launch (blockIdx.y = field).
constexpr int kLineBatch = 3;
struct LineBatch {
  int n = 0, kind = 0;                       // kind: 0 transform, 1 transform pair, 2 convection term, 3 rhs + hholtz-x
  DctLineArgs d0[kLineBatch], d1[kLineBatch];
  ConvLineArgs c[kLineBatch];
  RhsLineArgs r[kLineBatch];
};
bool line_batch_ok(int N);                   // the line lengths launch_line_batch covers
void launch_line_batch(const LineBatch& b, Stream& st);
bool launch_div_line(const DivLineArgs& a, Stream& st);     // div_line.h: S5 of the confined step per x-line
bool launch_corr_line(const CorrLineArgs& a, Stream& st);   // corr_line.h: S8 of the confined step per x-line
bool launch_prow_line(const ProwLineArgs& a, Stream& st);   // prow_line.h: S6 of the confined step per eigen row
bool launch_pres_line(const PresLineArgs& a, Stream& st);   // pres_line.h: S9 of the confined step per x-line
void launch_per_rows(const PerRowsArgs& a, Stream& st);     // per_rows.h: S5 / S8 / S9 of the periodic step, one thread per complex number
bool launch_rhs_line(const RhsLineArgs& a, Stream& st, long long* trace = nullptr);   // rhs_line.h: S3 of the confined step per x-line
// one y-line of a convection term: two backward transforms, the physical products, the forward transform with the
// 2/3 rule, all in registers + one exchange buffer (dct_line.h conv_line; three workgroups per CU)
struct ConvLineArgs;
bool launch_conv_line(const ConvLineArgs& c, Stream& st);

// weighted averages of the callback diagnostics on the device (field/average.rs:26-59 applied to
// eval_nu / eval_nuvol / eval_re, functions.rs:146-233).  Inputs are physical (nx x ny, pitch ld) arrays:
// total temperature T, its unscaled y-derivative dT, ux, uy; wx / wy = dx / length of the two axes.
//   out[0] = sum_i wx_i c_nu dT(i, 0)       out[1] = sum_i wx_i c_nu dT(i, ny-1)
//   out[2] = sum_ij wx_i wy_j (c_v1 dT + c_v2 T uy)      out[3] = sum_ij wx_i wy_j c_re sqrt(ux^2 + uy^2)
// two launches (per-row partial sums, then one workgroup), fixed summation order; `partial`: 4 nx doubles
void launch_diag_reduce(const double* T, const double* dT, const double* ux, const double* uy, long ld, int nx, int ny,
                        const double* wx, const double* wy, double c_nu, double c_v1, double c_v2, double c_re,
                        double* partial, double* out4, Stream& st);

// measurement only: `blocks` workgroups x 4 waves x `iters` x 8 independent v_mfma_f64_16x16x4_f64 chains
void launch_mfma_peak(double* out, int blocks, int iters, Stream& st, int kind = 0);

// the three-term stencil (cheb_dirichlet_neumann: the "hc" temperature along y) and its seven-diagonal solves, pdma.h
void launch_sten3_rows(const Sten3RowsArgs& a, Stream& st);     // YX array, composite -> orthonormal rows
void launch_pdma_cols(const PdmaColsArgs& a, Stream& st);       // YX array, B2 rows + PdmaPlus2 along y (one thread per column)
void launch_sten3_lines(const Sten3LinesArgs& a, Stream& st);   // contiguous lines (generic operators)
void launch_pdma_lines(const PdmaLinesArgs& a, Stream& st);

// p[idx] = value (single element; used for pseu[0,0] = 0)
void launch_set_element(double* p, long idx, double value, Stream& st);

// sum of squares + NaN flag of a pitched 2-D array into out[0] (sum), out[1] (nan count)
void launch_sumsq(const double* a, long ld, int rows, int cols, double* out2, Stream& st);

}  // namespace rpde
