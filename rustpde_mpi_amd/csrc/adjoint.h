// Navier2DAdjoint (src/navier_stokes/steady_adjoint.rs): adjoint descent to steady states of the Rayleigh-Benard problem.
//
// One `update()` (steady_adjoint.rs:541-608) = a forward Navier-Stokes step with DT_NAVIER for the residual, the residual
// measured in the norm (1 - WEIGHT_LAPLACIAN D2)^-1 (three tensor Helmholtz solves, src/solver/hholtz.rs), and an explicit
// adjoint step (steady_adjoint_eq.rs:234-438).  First slice of SURVEY.md section 8f-4: the step runs on the device through
// the GENERIC operators of ops.h (canonical XY layout; every transform, stencil, derivative and solve is a launch of the
// same kernels the operator-level C ABI exposes) plus three element-wise kernels of its own (adjoint.cc); all fields stay in
// HBM, the host sees scalars.  It is NOT fused like Navier2DEngine's step (several hundred launches per update against 24): the
// reference solver is a research add-on with no headline benchmark; correctness first, a schedule of whole-line kernels
// is the follow-up.
#pragma once
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "engine.h"
#include "ops.h"

namespace rpde {

// What the two solvers of SURVEY section 8f-4 built so far share: the fields (canonical XY layout, HBM), the four function spaces,
// the lift, the forward-step solvers, and the pieces of the equations as launches of the generic operators.
class GenericFlow2D {
 public:
  GenericFlow2D(int nx, int ny, double ra, double pr, double dt, double aspect, const std::string& bc, bool periodic,
                double dt_helmholtz, std::initializer_list<const char*> extra_fields);
  virtual ~GenericFlow2D();
  double time() const { return time_; }
  int nx() const { return nx_; }
  int ny() const { return ny_; }
  double dt() const { return dt_; }
  void reset_time() { time_ = 0.0; }
  double param(const std::string& key) const;       // ra, pr, nu, ka
  double div_norm();
  void set_velocity(double amp, double m, double n);
  void set_temperature(double amp, double m, double n);
  void spectral_shape(const std::string& name, int* rows, int* cols, int* elem);
  void set_field_spectral(const std::string& name, const double* host, size_t len);
  void get_field_spectral(const std::string& name, double* host, size_t len);
  void set_field_physical(const std::string& name, const double* host, size_t len);
  void get_field_physical(const std::string& name, double* host, size_t len);
  void write(const std::string& filename);
  void read(const std::string& filename);
  void grid(int axis, double* x, size_t len) const;
  Stream& stream() { return st_; }

 protected:
  struct F { Space2Ops* sp; Arr2 vhat; bool read_only = false;
             bool constant = false; };   // the lift and the mean fields: their gradients are computed once per run, not once per term and step
  // gradients of the constant fields (physical: key d0, d1; orthonormal: key 100 + d0, d1), dropped whenever a field is set
  std::map<std::tuple<const F*, int, int>, Arr2> const_grad_;
  unsigned long const_gen_ = 1;          // counts the changes of the constant fields (drop_constant_gradients)
  void drop_constant_gradients();
  F& field(const std::string& name);
  void acc_to_ortho(F& f, double s, Arr2& out);
  void acc_gradient(F& f, int d0, int d1, double s, Arr2& out);
  void conv_term(const Arr2& u, F& f, int d0, int d1, double s, bool first);
  void conv_finish(Arr2& out);
  void zero(Arr2& a);
  void lincomb(Arr2& out, double a, const Arr2& x, double b, const Arr2& y);
  void backward(F& f, Arr2& phys);
  void div(Arr2& out);
  void solve_pres(const Arr2& div);
  void correct_velocity(double c);
  double norm(const Arr2& a);
  std::vector<std::string> snapshot_fields_{"velx", "vely", "temp", "pres", "tempbc"};   // write(): group names ux uy temp pres tempbc

  int nx_, ny_, ex_;
  bool periodic_, hc_ = false;
  Vec hc_profile() const;
  double ra_, pr_, nu_, ka_, dt_, sx_, sy_, time_ = 0.0;
  Stream st_;
  std::unique_ptr<Space2Ops> sp_vel_, sp_temp_, sp_ortho_, sp_pseu_;
  std::unique_ptr<HholtzAdiOp> hh_vel_, hh_temp_;
  std::unique_ptr<PoissonOp> pois_;
  std::map<std::string, F> f_;
  Arr2 rhs_, div_, t0_, t1_, old_[3], ux_, uy_, ta_, ph_, conv_, cv_, cp_, cvt_;
  DBuf red_;
};

class Navier2DAdjointEngine : public GenericFlow2D {
 public:
  static constexpr double kResTol = 1e-7;           // RES_TOL          steady_adjoint.rs:60
  static constexpr double kWeightLaplacian = 1e-1;  // WEIGHT_LAPLACIAN steady_adjoint.rs:62
  static constexpr double kDtNavier = 1e-3;         // DT_NAVIER        steady_adjoint.rs:64

  // Navier2DAdjoint::new_confined / new_periodic (steady_adjoint.rs:215-370, 372-531); bc = "rbc" (Navier2DLnse / Navier2DNonLin take "hc" too; here "hc": the reference
  // builds Hholtz -- a four-diagonal FdmaTensor -- on the three-term base cheb_dirichlet_neumann, which its Fdma cannot hold;
  // refused here)
  Navier2DAdjointEngine(int nx, int ny, double ra, double pr, double dt, double aspect, const std::string& bc, bool periodic);

  void update(int nsteps);                          // Integrate::update, steady_adjoint.rs:541-608
  bool exit();                                      // steady_adjoint.rs:624-638: NaN divergence, or mean residual < RES_TOL
  void norm_residual(double out[3]);                // DivNorm::norm_residual steady_adjoint_eq.rs:44-50


 private:
  std::unique_ptr<TensorHholtzOp> norm_vel_, norm_temp_;
  // The forward step for the residual (steady_adjoint.rs:547-585) IS Navier2D::update with DT_NAVIER and the buoyancy taken
  // without the lift: it runs on Navier2DEngine's fused schedule (16 launches) instead of a composition of generic operators;
  // u, v, T, p go in and u, v, T, p, pseu come back as device arrays (RPDE_ADJOINT_FUSED=0: the generic composition, A/B)
  std::unique_ptr<Navier2DEngine> fwd_;
  void forward_step_generic();
  void forward_step_fused();
};

// Navier2DLnse (src/navier_stokes_lnse/lnse.rs:24-63, 263-288; equations lnse_eq.rs): the Navier-Stokes equations linearised
// about mean fields U, V, T (MeanFields, meanfield.rs:20-56: orthonormal spaces; "rbc": U = V = 0, T = the conduction profile).
// One update() = Navier2D's step with the convection terms u . grad(U) + U . grad(u), the buoyancy temp.to_ortho() without a lift
// and no lift terms in the temperature equation.  Second slice of SURVEY section 8f-4, on the same generic operators.
class Navier2DLnseEngine : public GenericFlow2D {
 public:
  // mean_file: MeanFields::read_from_confined / _periodic (meanfield.rs:92-127, 194-231) -- "ux/v", "uy/v", "temp/v" (+ "tempbc/v")
  // of a snapshot if the file exists (the reference looks for "mean.h5"), the boundary condition's default mean otherwise
  // nonlinear: Navier2DNonLin (nonlin.rs, nonlin_eq.rs, nonlin_adj_eq.rs, nonlin_adj_grad.rs) -- the same fields and solvers; the
  // step keeps u . grad(u) + U . grad(U), the mean's diffusion and the mean temperature in the buoyancy; update_direct() records
  // the forward states (field_history, in HBM) that the adjoint step's convection terms read back, last in first out
  Navier2DLnseEngine(int nx, int ny, double ra, double pr, double dt, double aspect, const std::string& bc, bool periodic,
                     const std::string& mean_file, bool nonlinear = false);
  bool nonlinear() const { return nonlin_; }
  size_t history_len() const { return hist_.size(); }
  void clear_history() { dev_sync(st_); hist_.clear(); }
  void write(const std::string& filename);          // nonlin_io.rs:44-66: + the mean fields as ux_base, uy_base, temp_base
  void update(int nsteps);                          // Integrate::update, lnse.rs:263-288
  bool exit();                                      // lnse.rs:305-313: NaN divergence
  void set_mean_physical(const std::string& name, const double* host, size_t len);   // "velx" | "vely" | "temp": MeanFields::read's assignment + forward
  void get_mean_physical(const std::string& name, double* host, size_t len);

  // ---- adjoint-based sensitivity of the final energy (third slice of SURVEY 8f-4) ----
  void update_direct(int nsteps);                                     // lnse_adj_grad.rs:43-68 = update(); nonlin_adj_grad.rs:43-81: + history
  void update_adjoint(int nsteps);                                    // lnse_adj_grad.rs:71-99, equations lnse_adj_eq.rs
  // functions.rs:11-58 `energy`: 0.5 sum(b1 u^2 + b1 v^2 + b2 T^2) over the grid points of the physical fields; target (three
  // physical arrays of nx*ny doubles, or all null): the fields minus the target (lnse_adj_grad.rs:141-155)
  double energy(double beta1, double beta2, const double* tu = nullptr, const double* tv = nullptr, const double* tt = nullptr);
  // lnse_adj_grad.rs:105-202: forward loop to max_time, energy, adjoint initial condition beta (state - target), adjoint loop,
  // gradient = -(physical adjoint fields as the state holds them: those of the START of the last adjoint step, :185-191).
  // gu, gv, gt: nx*ny doubles each; filename: the reference's "data/grad_adjoint.h5" (groups ux, uy, temp), or null
  // save_intervall > 0: the snapshots of the two loops ("data/flow{time:0>8.2}.h5" / "data/adjoint{time:0>8.2}.h5" and the info files,
  // :122-130, :176-181; directory "data" of the working directory like the reference), <= 0: None
  double grad_adjoint(double max_time, double save_intervall, double beta1, double beta2, const double* tu, const double* tv,
                      const double* tt, double* gu, double* gv, double* gt, const char* filename, long* timesteps);
  // lnse_io.rs:73-126 / nonlin_io.rs:72-142: snapshot on the write interval (< 0: OUTPUT_INTERVALL = 1, lnse.rs:21), then unless
  // suppressed the line on stdout and in the info file (Navier2DLnse: time u2 v2 t2; Navier2DNonLin: time Nu Nuv Re u2 v2 t2)
  void callback_from_filename(const std::string& flow_name, const std::string& info_name, bool suppress_io, double write_flow_intervall);
  // out[7] = |div|, Nu, Nuv, Re (nonlin_io.rs:145-198 on state + mean; NaN for Navier2DLnse, which has none), <u^2>, <v^2>, <T^2>
  void diagnostics(double out[7]);
  // lnse_fd_grad.rs:31-157: one integration per perturbed grid point (eps = 1e-5).  points: npoints triples (field 0 / 1 / 2, i, j),
  // or null = every point of velx, vely, temp in the reference's order; entries not visited are 0
  // save_intervall > 0: the base run (and only it, :54) calls Integrate::callback on the interval -- data/flow{time:0>8.2}.h5, data/info.txt
  void grad_fd(double max_time, double beta1, double beta2, const int* points, long npoints, double* gu, double* gv, double* gt,
               const char* filename, double save_intervall = 0.0);
  long integrate(double max_time, double save_intervall = 0.0);       // src/lib.rs:187-219; save_intervall <= 0: None

 private:
  struct Hist { F velx, vely, temp; };              // one forward state: the spectral arrays (the physical ones = backward of them)
  bool exit_grad(double max_time, long timestep);                     // lnse_adj_grad.rs:204-225
  void conv_adj(F& f, int d0, int d1, bool mean_gradients, Arr2& out, Hist* nl);   // conv_*_adjoint of lnse_adj_eq.rs:16-94 / nonlin_adj_eq.rs:16-118
  double sumsq(const Arr2& a);
  void write_gradient(const char* filename, const double* gu, const double* gv, const double* gt);
  Arr2 tp_;                                         // physical temperature of the adjoint step
  bool nonlin_ = false;
  std::vector<std::unique_ptr<Hist>> hist_;
  Arr2 unl_, vnl_;                                  // physical velocities of the history entry of the current adjoint step
  void mean_diffusion(F& mean_f, double kappa);     // rhs += dt kappa (dxx + dyy) mean (nonlin_eq.rs:204-206, 221-223, 236-238)
  void conv_lin(F& mean_f, F& f, Arr2& out);        // conv_velx / vely / temp of lnse_eq.rs:59-110
  F& mean(const std::string& name) { return field("mean_" + name); }
  Arr2 um_, vm_;                                    // physical mean velocities (constant during a run)
  void refresh_mean();
  // update() of the LINEAR solver, bc = "rbc", on Navier2DEngine's fused schedule (16 launches; engine.h `lnse`) where the whole-line
  // convection kernel covers the y-lines; RPDE_LNSE_FUSED=0 or any other shape: the composition of generic operators below.  The
  // state goes in and comes back as device arrays per update(n) call, the mean arrays whenever a mean field was set.
  std::unique_ptr<Navier2DEngine> fwd_;
  unsigned long fwd_mean_gen_ = 0;                  // const_gen_ of the mean arrays fwd_ holds
  void update_fused(int nsteps);
  // update_adjoint() of the linear solver, confined, on the same schedule (engine.h lnse = 3); the physical arrays ux_ / uy_ / tp_ the
  // gradient is read from are refreshed by the generic backward transforms at the start of every step, as in the generic form
  std::unique_ptr<Navier2DEngine> adj_;
  unsigned long adj_mean_gen_ = 0;
  void push_mean(Navier2DEngine& e, bool negate_velocities);
};

}  // namespace rpde
