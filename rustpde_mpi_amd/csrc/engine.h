// Navier2D engine: device-resident mirror of rustpde's `Navier2D<T, S>` for 2-D Rayleigh-Benard
// convection (src/navier_stokes/navier.rs:49-89, 215-308, 336-428, 438-466).  All fields stay in
// HBM; `update(n)` runs n time steps as a fixed sequence of line programs (x-lines on "YX"
// arrays, y-lines on "XY" arrays), transposes and, for the confined case, four f64 MFMA GEMMs
// per step (the eigen-decomposition Poisson solve, src/solver/poisson.rs:195-236).
#pragma once
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>

#include "dct_line.h"
#include "ops.h"

namespace rpde {

// personalised all-to-all on buffers of doubles that live where the engine's arrays live (HBM in
// the HIP build): segment q of `send` (sendcounts[q] doubles) goes to rank q, segment s of `recv`
// arrives from rank s.  Returns 0 on success.  Called from inside update() with the engine's
// stream drained; the data must have landed when it returns.  (funspace Decomp2d::transpose_* =
// MPI_Alltoallv in the reference, src/field_mpi.rs:456-477.)
typedef int (*AllToAllvFn)(void* user, const double* send, const int64_t* sendcounts, double* recv,
                           const int64_t* recvcounts);
struct RcclComm;
struct CommCb {
  int rank = 0, size = 1;
  AllToAllvFn fn = nullptr;      // host callback transport (stream drained before every call), or
  void* user = nullptr;
  RcclComm* rccl = nullptr;      // native RCCL transport on the engine's stream (rccl_transport.h); owned
};

class Navier2DEngine {
 public:
  // buoyancy_lift = false: the buoyancy term of solve_vely is temp.to_ortho() alone, without the lift -- the forward step inside
  // Navier2DAdjoint::update (steady_adjoint_eq.rs:147-160; everything else of that step is Navier2D's, steady_adjoint.rs:547-585)
  Navier2DEngine(int nx, int ny, double ra, double pr, double dt, double aspect,
                 const std::string& bc, bool periodic, const CommCb* comm = nullptr, bool buoyancy_lift = true, int lnse = 0);
  // lnse = true: the step of Navier2DLnse (lnse.rs:263-288, lnse_eq.rs) on this schedule -- no lift anywhere (the temperature is the
  // deviation from the mean), and the convection terms linearised about mean fields: U d/dx f + V d/dy f + u d/dx M + v d/dy M
  // (lnse_eq.rs:59-110) with the physical mean velocities and the physical gradients of the three mean fields as time-independent
  // device arrays (set_lnse_mean_device).  One rank, bc = "rbc", y-lines the whole-line convection kernel covers -- the constructor
  // throws otherwise and Navier2DLnseEngine keeps its composition of generic operators.
  // which: 0 U, 1 V (physical mean velocities); 2 + 2 f + d: d/dx (d = 0) or d/dy (d = 1) of mean field f (0 velx, 1 vely, 2 temp),
  // physical; `phys` = (nx x ny) canonical
  void set_lnse_mean_device(int which, const Arr2& phys);
  // lnse = 2: the step of Navier2DNonLin (nonlin.rs:264-296, nonlin_eq.rs) -- the convection terms (U + u) . grad(M + f) (the classic kernel
  // with the mean velocities added to u, v and the mean gradients in the lift's place, for all three fields), and what the mean fields
  // add to the right-hand sides (their diffusion, the mean temperature in the buoyancy) as three time-independent composite arrays
  // H^-1 c added behind the Helmholtz solves (the solve is linear): fields "nl_velx", "nl_vely", "nl_temp" (set_field_spectral_device).
  // lnse = 3: Navier2DLnse::update_adjoint (lnse_adj_grad.rs:71-99, lnse_adj_eq.rs): the convection terms
  // -(U d/dx f + V d/dy f) + u* d_j U + v* d_j V + T* d_j T (conv_line<N, 3>; the caller hands over MINUS the mean velocities as arrays
  // 0 / 1), no buoyancy in the vely equation, dt vely.to_ortho() of the step's start in the temperature equation.
  ~Navier2DEngine();

  // initial conditions (src/navier_stokes/navier.rs:161-182, functions.rs:85-126)
  void set_velocity(double amp, double m, double n);
  void set_temperature(double amp, double m, double n);
  void init_random(double amp, unsigned long long seed);
  void reset_time() { time_ = 0.0; }

  // field access; name in {velx, vely, temp, pres, pseu}; physical arrays are (nx x ny) row-major
  // f64, spectral arrays have the reference's `vhat` shape (complex interleaved when periodic)
  void set_field_physical(const std::string& name, const double* host, size_t len);
  void get_field_physical(const std::string& name, double* host, size_t len);
  void set_field_spectral(const std::string& name, const double* host, size_t len);
  void get_field_spectral(const std::string& name, double* host, size_t len);
  void spectral_shape(const std::string& name, int* rows, int* cols, int* elem);
  // the same between device arrays in the canonical layout (an engine embedded in another solver, adjoint.cc): no host copy
  void set_field_spectral_device(const std::string& name, const Arr2& canonical);
  void get_field_spectral_device(const std::string& name, Arr2& canonical);

  void update(int nsteps);           // n x Integrate::update
  double div_norm();                 // ||div||_2 of the current velocity (navier_eq.rs:33-51)
  // Nusselt number at the plates, volumetric Nusselt number, Reynolds number
  // (eval_nu / eval_nuvol / eval_re, src/navier_stokes/functions.rs:146-233; callback cadence only)
  void diagnostics(double* nu, double* nuvol, double* re);
  // NaN guard of Integrate::exit (navier.rs:482-489).  The reference recomputes the divergence
  // every step just to test its norm for NaN (src/lib.rs:214); here the stores of the corrected
  // velocities, the pressure and (confined) the temperature raise a device flag, and exit() is one
  // 4-byte read of it after the stream drained: no allocation, no extra pass over the fields.
  // (A NaN spreads to every coefficient of every field within one step, so the flag and the
  // reference's test agree from the step after the first NaN at the latest.)  After a host
  // write to a field (set_field / init_random) and before the next update() the divergence is
  // evaluated like the reference does.
  bool exit();
  // Snapshots in the reference's HDF5 layout (navier_io.rs:21-62, field/io.rs:74-110):
  // groups ux, uy, temp, pres, tempbc with x, dx, y, dy, v, vhat (vhat_re / vhat_im when complex), root
  // scalars time, ra, pr, nu, ka.  read() restores vhat of ux, uy, temp, pres and the time; a snapshot
  // of another resolution is truncated / zero-padded in spectral space (field/io.rs:151-176).
  // Written and parsed by csrc/h5lite (no libhdf5 in this image).  Sharded: collective, rank 0 writes.
  void write(const std::string& filename);
  void read(const std::string& filename);
  // callback_from_filename (navier_io.rs:84-149): snapshot (on its interval when write_flow_intervall >= 0,
  // always when < 0 = None), then "time |div| Nu Nuv Re" to stdout and "time nu nuv re" appended to info_name
  void callback_from_filename(const std::string& flow_name, const std::string& info_name, bool suppress_io,
                              double write_flow_intervall);
  void callback();                       // Integrate::callback (navier.rs:476-480): data/flow{time:0>8.2}.h5, data/info.txt
  // Statistics (src/navier_stokes/statistics.rs:11-108, hooked into the callback at navier_io.rs:105-121): four
  // fields of the orthonormal `field` space kept on the device -- the running mean of temp.to_ortho() (`temp`), the
  // LAST velx / vely .to_ortho() (`ux`, `uy`: the reference assigns, it does not average) and the Nusselt field of
  // the last snapshot (`nusselt`, statistics.rs:248-271) -- plus avg_time, tot_time, num_save.
  void statistics_enable(double save_stat, double write_stat);   // navier.statistics = Some(Statistics::new(&navier, ..))
  bool statistics_enabled() const { return stats_ != nullptr; }
  // `navier.statistics = Some(stats)` / `= None` (navier_io.rs:105: the callback only acts on Some): the hook can be
  // taken off and put back without dropping the accumulated fields.  statistics_enable() leaves it attached.
  void statistics_attach(bool on) { stats_attached_ = on; }
  void statistics_update();                                       // Statistics::update(temp, velx, vely to_ortho, time)
  void statistics_write(const std::string& filename);             // groups temp, ux, uy, nusselt + tot_time, avg_time, num_save (u64), params
  void statistics_read(const std::string& filename);
  // name in {temp, ux, uy, nusselt}: coefficients in the `field` space (nx x ny, or (nx/2+1) x ny complex interleaved)
  void statistics_get(const std::string& name, double* host, size_t len);
  void statistics_scalars(double* avg_time, double* tot_time, long long* num_save) const;
  void set_write_intervall(double v) { write_intervall_ = v; }   // `write_intervall: Option<f64>`; < 0 = None
  double time() const { return time_; }
  double dt() const { return dt_; }
  double param(const std::string& key) const;
  const PoissonOp& poisson() const { return *pois_; }
  double last_update_ms() const { return last_ms_; }
  // per-launch profile: run `nsteps` with HIP events around every launch; returns a text table
  // "tag<TAB>launches<TAB>ms_total<TAB>algorithmic_bytes_per_launch<TAB>flops_per_launch" per line
  std::string profile(int nsteps);
  // launches whose tag contains `tag` are bracketed by HIP events inside update() (empty: none)
  void set_timed_tag(const std::string& tag) { timed_tag_ = tag; timed_ms_ = 0; timed_count_ = 0; }
  void get_timed(double* ms, long* count) const { *ms = timed_ms_; *count = timed_count_; }
  std::string describe_step() const;
  // diagnostics: one step with the first line-program launch whose tag contains `tag` instrumented
  // (Program::trace); text table of the per-op shader-clock durations over its workgroups
  std::string trace_launch(const std::string& tag);
  void grid(int axis, double* x, size_t len) const;
  void sync() { dev_sync(st_); }
  int nx() const { return nx_; }
  int ny() const { return ny_; }
  bool periodic() const { return periodic_; }
  int nranks() const { return comm_.size; }
  double exchange_bytes_per_step() const { return xchg_bytes_; }   // bytes this rank sends per step
  int exchanges_per_step() const { return xchg_count_; }

  Stream st_;

 private:
  struct Field;   // per-field bookkeeping
  void construct(int nx, int ny, double ra, double pr, double dt, double aspect, bool periodic);
  void release_device_objects();   // stream, events, graph, communicator: also on a throwing constructor
  Field& field(const std::string& name);
  void state_to_canonical(Field& f, Arr2& out, bool wait = true);
  void canonical_to_state(const Arr2& in, Field& f);
  void refresh_gy();

  // ---- pencil decomposition (P ranks): YX arrays are split by rows (y index), XY arrays by rows
  // (x index: physical points `xpart_`, spectral modes `kpart_`); a layout change is an exchange
  CommCb comm_;
  std::vector<int> ypart_, xpart_, kpart_;
  int yb_ = 0, ye_ = 0, nyl_ = 0, nxl_ = 0;
  DBuf sendbuf_, recvbuf_;
  double xchg_bytes_ = 0.0;
  int xchg_count_ = 0;
  double* yx(DBuf& b) const { return b.p + 2 * ldx_; }   // YX buffers carry two halo rows in front
  static std::vector<int> split(int n, int parts);
  static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
  int ylines(int rows) const { return clampi((ye_ < rows ? ye_ : rows) - yb_, 0, rows); }
  int xlines(int rows, bool spec) const {
    const std::vector<int>& p = spec ? kpart_ : xpart_;
    const int b = p[comm_.rank], e = p[comm_.rank + 1];
    return clampi((e < rows ? e : rows) - b, 0, rows);
  }
  int xb(bool spec) const { return (spec ? kpart_ : xpart_)[comm_.rank]; }
  void alltoallv(const double* send, const std::vector<int64_t>& sc, double* recv, const std::vector<int64_t>& rc) { alltoallv_on(st_, send, sc, recv, rc); }
  void alltoallv_on(Stream& s, const double* send, const std::vector<int64_t>& sc, double* recv, const std::vector<int64_t>& rc);
  // out = in^T across ranks.  to_xy: input YX (rows split by ypart_), output XY (rows split by
  // xpart_/kpart_); otherwise the reverse.  rows/cols are the GLOBAL logical sizes of the input.
  struct Xfer { const double* in; long ldi; double* out; long ldo; };
  void exchange_batch(const std::vector<Xfer>& xs, int rows, int cols, int elem, bool to_xy, bool spec) { exchange_batch_on(st_, xs, rows, cols, elem, to_xy, spec); }
  void exchange_batch_on(Stream& s, const std::vector<Xfer>& xs, int rows, int cols, int elem, bool to_xy, bool spec);
  // ---- overlap of the array transposes of a pencil-sharded step with the compute of the neighbouring fields (round 5):
  // the transposes T1 / T2 of ONE field go out as soon as that field's producer has run, on a second stream (pack, all-to-all,
  // unpack), while the main stream computes the next field; a consumer waits for the event of the exchange it reads.
  // RPDE_OVERLAP=0: the serial order of rounds 2-4 (every exchange on the main stream, T1 as one batch of six arrays).
  static constexpr int kMaxAsync = 8;
  bool overlap_ = false;
  Stream st2_;                 // the exchange stream
#ifndef RPDE_EMU
  hipEvent_t xdone_[kMaxAsync] = {}, xprod_ = nullptr;   // exchange `id` has landed; the producer of an exchange has run
#endif
  void after_exchange(unsigned wait_mask);   // the main stream waits for the exchanges of `wait_mask`
  // re-orders step_ for the overlap (the builders emit the serial order: S1 x3, T1 x6, S2 x5, T2 x3, S3 x3): S1(u), T1(u)*,
  // S1(v), T1(v)*, S1(T), T1(T)*, S2 u -> phys [waits T1(u)], v -> phys [T1(v)], conv_velx, T2(velx)*, conv_vely, T2(vely)*,
  // conv_temp [T1(T)], T2(temp)*, S3 velx [T2(velx)], S3 vely [T2(vely)], S3 temp [T2(temp)]  (* = on the exchange stream);
  // false (and the serial order stays) if the step does not have that shape
  bool apply_overlap_order();
  void exchange(const double* in, long ldi, double* out, long ldo, int rows, int cols, int elem,
                bool to_xy, bool spec) {
    exchange_batch({Xfer{in, ldi, out, ldo}}, rows, cols, elem, to_xy, spec);
  }
  // runs step_[i] (and, when sharded, the compatible exchanges that directly follow it in one
  // all-to-all); returns the index of the next launch
  size_t run_from(size_t i);
  size_t group_end(size_t i) const;   // one past the last launch that goes out together with step_[i]
  static constexpr int kMaxBatch = 6;
  // halo rows of up to three YX arrays in one exchange: `front` rows in front of the local rows (from rank - 1), `tail`
  // rows behind them (from rank + 1)
  void halo_rows(double* const* arr, int n, int front, int tail);
  void run_col_hholtz(ColHhArgs a, const ColHh1Tabs* x1 = nullptr, unsigned long long* site = nullptr, long long* trace = nullptr);
  void run_col_diff(ColDiffArgs a, unsigned long long* site);   // column scans, one rank or rows split over the ranks (colscan.h)
  ColHhDev colhh_vel_, colhh_temp_;   // Helmholtz-y tables of this rank's rows
  DBuf colsumm_, colsend_, colgath_, halo_s_, halo_r_;
  DBuf pdma_ws_;                      // "hc": workspace of the blocked PdmaPlus2 column solve (pdma.h)
  void scatter_rows_yx(const double* full, long ldf, DBuf& dst, int rows, int ncols);
  void scatter_rows_xy(const double* full, long ldf, DBuf& dst, int rows, int ncols, bool spec);
  void gather_rows(const double* local, long ld, int rows_global, const std::vector<int>& part,
                   double* full);

  int nx_, ny_, mx_, my_, kx_;   // kx_: x-modes of a spectral line (nx, or nx/2+1 complex)
  bool periodic_;
  bool hc_ = false;              // bc = "hc": the temperature is cheb_dirichlet_neumann along y (navier.rs:245-248 / 366-369)
  double ra_, pr_, nu_, ka_, dt_, time_ = 0.0, sx_, sy_;
  double last_ms_ = 0.0;
  long ldx_, ldy_;               // pitches (doubles) of YX and XY work arrays
  int ex_;                       // doubles per spectral x-entry (1 confined, 2 periodic)

  std::unique_ptr<Space2Ops> sp_vel_, sp_temp_, sp_ortho_, sp_pseu_;
  std::unique_ptr<HholtzAdiOp> hh_vel_, hh_temp_;
  std::unique_ptr<PoissonOp> pois_;

  // state + constants (YX layout: row = y index, contiguous x)
  DBuf U_, V_, T_, P_, GY_, GX_, TBC_, TBC2_, DIV_;   // GX_, GY_: d/dx p, d/dy p kept from the pressure update
  DBuf Y_[6];
  DBuf TO_;                      // "hc": the temperature in orthonormal-y, composite-x coefficients (ny rows), rebuilt every step
  // XY layout work arrays (row = x index, contiguous y)
  DBuf X_[9], BX_, BY_, PS_;
  DBuf red_;                     // reduction scratch (2 doubles)
  DBuf nanflag_;                 // device flag raised by the guarded stores of the step (int at offset 0)
  int* hflag_ = nullptr;         // pinned host landing pad of the flag
  bool dirty_ = false;           // a field was written from the host since the last update()
  double write_intervall_ = -1.0;   // navier.rs:79 `write_intervall: Option<f64>` (None)
  struct Stats;
  std::unique_ptr<Stats> stats_;   // `statistics: Option<Statistics<T, S>>` (navier.rs:88)
  bool stats_attached_ = true;     // false: the Statistics exist but `navier.statistics` is None
  int* flagp() const { return reinterpret_cast<int*>(nanflag_.p); }
  bool read_nanflag();
  DBuf postcut_x_, postcut_y_;   // forward-DCT scaling with the 2/3 dealiasing cut folded in
  DBuf UP_, VP_;                 // physical velocities of the step (XY), shared by the three conv programs
  DBuf colv1_, cols1_, colv2_, cols2_, coldv_, colds_;   // block carries of the column scans (colscan.h)
  DBuf coldtot_;                 // single-pass y-derivative (colscan1.h): super-block sums
  DBuf colagg_, colsync_;        // single-pass column scans (colscan1.h): super-block aggregates; the error flag
  std::vector<std::unique_ptr<DBuf>> col_sites_;   // one synchronisation area per launch site of a single-pass column scan (never reset: epochs)
  unsigned long long* new_col_site(size_t words);
  unsigned long long* gy_site_ = nullptr;          // refresh_gy's column scan
  int col1_W_ = 0, col1_NSB_ = 0, col1_tiles_ = 0;   // 0: the three-kernel form
  std::map<std::string, std::unique_ptr<Field>> fields_;

  // the step as a list of launches
  struct Launch {
    enum Type { kLine, kTranspose, kGemmPairNT, kGemmPairNN, kSetElem, kHalo, kColHholtz, kColDiff, kDctLine, kDctLine2, kConvLine, kRhsLine,
                kSten3Rows, kPdmaCols, kCorrLine, kPdmaLines, kDivLine, kRfftPair, kFourRhs, kProwLine, kPresLine, kPerRows } type;
    PerRowsArgs pr{};            // kPerRows  (periodic S5 / S8 / S9: element-wise along x, per_rows.h)
    RfftLineArgs rf{}, rf2{};    // kRfftPair (periodic S1: value and x-derivative of a spectral line, rfft_line.h)
    FourRhsArgs fr{};            // kFourRhs  (periodic S3)
    DivLineArgs dvl{};           // kDivLine
    ProwLineArgs prl{};          // kProwLine
    PresLineArgs psl{};          // kPresLine
    PdmaLinesArgs pl{};          // kPdmaLines ("hc", pencil-sharded: Helmholtz-y of the temperature on the y-lines of an x-pencil)
    CorrLineArgs crl{};          // kCorrLine
    Sten3RowsArgs s3{};          // kSten3Rows ("hc": temperature composite -> orthonormal along y, pdma.h)
    PdmaColsArgs pc{};           // kPdmaCols  ("hc": Helmholtz-y of the temperature, PdmaPlus2)
    RhsLineArgs rl{};            // kRhsLine
    ConvLineArgs cl{};           // kConvLine
    DctLineArgs dl{}, dl2{};     // kDctLine; kDctLine2: two transforms of the same lines in one launch
    GemmProblem gp[2];           // kGemmPair*
    ColHhArgs ch{};              // kColHholtz
    unsigned long long* site = nullptr;   // kColHholtz / kColDiff on one rank: the launch site's synchronisation area (colscan1.h)
    ColHh1Tabs ch1[kColMaxFields]{};   // kColHholtz on one rank: the tables of the single-pass form (colscan1.h)
    ColDiffArgs cd{};            // kColDiff
    bool to_xy = true, spec = false;
    Program pg;                  // kLine
    double* hal[3] = {nullptr, nullptr, nullptr};   // kHalo: arrays, rows in front / behind
    int nhal = 0, front = 0, tail = 0;
    const double* in = nullptr;  // transposes
    double* out = nullptr;
    long ldi = 0, ldo = 0;
    int rows = 0, cols = 0, elem = 1;
    const char* tag = "";
    int async_id = -1;           // kTranspose, overlap_: the exchange of this group runs on st2_ and signals xdone_[async_id]
    unsigned wait_mask = 0;      // any launch: before it, the main stream waits for these exchanges
    double bytes = 0.0;          // algorithmic HBM bytes of one launch (reads + writes)
    double flops = 0.0;          // floating point operations of one launch (GEMMs)
  };
  static int line_batch_kind(const Launch& l);
  std::string group_tag(size_t i, size_t j) const;
  std::string timed_tag_;
  double timed_ms_ = 0.0;
  long timed_count_ = 0;
  std::vector<Launch> step_;
#ifndef RPDE_EMU
  hipGraphExec_t graph_exec_ = nullptr;   // the whole step captured once (single GPU): replay removes
  bool graph_tried_ = false;              // the per-launch host cost that dominates small grids
  hipEvent_t ev0_ = nullptr, ev1_ = nullptr;   // brackets of update(), created once
  // The last four launches of a step are two independent chains behind the second eigen-transform: { C7 correction-y -> S8 correction-x }
  // writes the velocities from the pseudo-pressure, { S9 pressure update -> C10 d/dy pres } the pressure and its gradients.  With
  // RPDE_FORK the second chain runs on a stream of its own between two events (two parallel branches of the captured graph): on small
  // grids, where a launch does not fill the chip, the chains overlap (engine.cc run_step).
  Stream stf_;
  hipEvent_t evfork_ = nullptr, evjoin_ = nullptr;
  int fork_side_ = -1;                    // index of the first launch of the second chain (S9), -1: no fork
  int fork_main_ = -1;                    // index of the first launch of the first chain (C7)
#endif
  void run_step();                        // one pass over step_ (forked where fork_side_ >= 0)
  bool use_graph_ = true;
  void add_line(const ProgramBuilder& pb, const char* tag);
  void add_gemm_pair(bool nn, const GemmProblem& p0, const GemmProblem& p1, const char* tag);
  void add_transpose(const double* in, long ldi, double* out, long ldo, int rows, int cols, int elem,
                     bool to_xy, bool spec, const char* tag);
  void add_halo(std::initializer_list<double*> arrays, int front, int tail, const char* tag);
  // Helmholtz solve along y of the three fields on YX arrays / Chebyshev y-derivative of a YX array
  // (single GPU: column scans instead of transpose -> line program -> transpose)
  void add_col_hholtz(const double* const in[3], double* const out[3], int ncols, const char* tag);   // "hc": the velocities only
  void add_hc_to_ortho(int ncols);                                  // "hc": T_ -> TO_ (three-term stencil along y)
  void add_hc_hholtz(const double* in, double* out, int ncols);     // "hc": Helmholtz-y of the temperature (PdmaPlus2 along y)
  void add_hc_hholtz_sharded(const double* in, double* out, int rows_x, int elem, bool spec);   // the same through x-pencils
  void add_col_diff(const double* in, double* out, int m_in, const double* low, int ncols, double scale, const char* tag);
  void add_col_corr(const double* ps, int half, double* outa, double* outb, int ncols, const char* tag);
  ColHhDev colcorr_a_, colcorr_b_;   // column problems of the velocity correction (confined, one GPU)
  DBuf coldot_, colkap_;             // rank-one sums of the column scans
  int pseu_half_ = 0;                // > 0: the step leaves pseu in YX layout, parity blocks `pseu_half_` columns apart
  bool pseu_in_yx_ = false;          // the canonical array PS_ is out of date (state_to_canonical refreshes it)
  // structure of the time-independent lift arrays, found once at setup (analyse_lift; RPDE_LIFT_STRUCT=0: not used, A/B): a lift that
  // does not depend on x -- "rbc": linear in y -- has identical y-lines of its physical gradients (BX_ / BY_: pitch 0, every line reads
  // line 0 out of the L2) and one x-coefficient per row of its spectral arrays (TBC_: column 0; its Laplacian TBC2_: nothing)
  long lift_ldl_ = -1;               // pitch the convection term reads BX_ / BY_ with (-1: ldy_)
  int tbc_cols_ = -1, tbc2_cols_ = -1;   // leading coefficients of a row of TBC_ (TBC0_) / TBC2_ that can be non-zero (-1: not analysed)
  void analyse_lift();
  bool buoyancy_lift_ = true;
  int lnse_ = 0;                      // 1: Navier2DLnse, 2: Navier2DNonLin, 3: the adjoint step of Navier2DLnse
  DBuf TP_, ZX_, ZY_;                // lnse_ == 3: the physical adjoint temperature (XY), zero arrays in the XY and the YX layout
  DBuf NLC_[3];                      // lnse_ == 2: H^-1 of the constant right-hand sides of velx, vely, temp (state layout)
  DBuf LM_[8];                       // lnse_: U, V, then d/dx, d/dy of the mean velx, vely, temp -- physical, XY layout (pitch ldy_)
  DBuf TBC0_;                        // buoyancy_lift_ = false: a zero array in the lift's place (buoyancy term only)
  bool pseu_from_y4_ = false;   // periodic step with the real-view S6: the canonical pseu is the complex transpose of Y_[4]
  // whole-line backward transform (dct_line.h) when the shape is covered; otherwise false and the caller adds the line program
  bool add_dct_line(const DctLineArgs& a, const char* tag);
  bool add_dct_line2(const DctLineArgs& a0, const DctLineArgs& a1, const char* tag);
  bool add_rfft_pair(const RfftLineArgs& a0, const RfftLineArgs& a1, const char* tag);
  bool add_four_rhs(const FourRhsArgs& a, const char* tag);
  bool add_conv_line(const ConvLineArgs& c, const char* tag);
  bool add_rhs_line(RhsLineArgs a, int which, const char* tag);   // S3 as one kernel per field (rhs_line.h)
  bool add_corr_line(CorrLineArgs a, const char* tag);            // S8 as one kernel (corr_line.h)
  bool add_div_line(const DivLineArgs& a, const char* tag);       // S5 as one kernel (div_line.h)
  bool add_prow_line(ProwLineArgs a, const char* tag);            // S6 as one kernel (prow_line.h)
  bool add_per_rows(const PerRowsArgs& a, const char* tag, double arrays);   // periodic S5 / S8 / S9 as element-wise kernels (per_rows.h); arrays: array passes of the stage
  bool add_pres_line(const PresLineArgs& a, const char* tag);     // S9 as one kernel (pres_line.h)
  struct RhsTabs { DBuf t0, t1, t2, q1, p2, q2, r2; };            // chunk-major (16 per thread) tables of rhs_line, per field kind
  RhsTabs rhs_tabs_[2];                                           // 0: velocities (Dirichlet x, nu), 1: temperature (Neumann x, ka)
  RhsTabs prow_tabs_;                                             // prow_line: B2 rows of the pressure space's y axis (t0, t1, t2 only)
  RhsTabs corr_tabs_[2];                                          // corr_line: 0 the derivative branch (velx), 1 the plain one (vely)
  DBuf corr_w_, corr_h_;                                          // its rank-one term
  void build_confined();
  void build_periodic();
  void run_launch(const Launch& l);
};

}  // namespace rpde
