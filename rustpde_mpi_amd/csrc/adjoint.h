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
#include <map>
#include <memory>
#include <string>

#include "ops.h"

namespace rpde {

class Navier2DAdjointEngine {
 public:
  static constexpr double kResTol = 1e-7;           // RES_TOL          steady_adjoint.rs:60
  static constexpr double kWeightLaplacian = 1e-1;  // WEIGHT_LAPLACIAN steady_adjoint.rs:62
  static constexpr double kDtNavier = 1e-3;         // DT_NAVIER        steady_adjoint.rs:64

  // Navier2DAdjoint::new_confined / new_periodic (steady_adjoint.rs:215-370, 372-531); bc = "rbc" ("hc": the reference
  // builds Hholtz -- a four-diagonal FdmaTensor -- on the three-term base cheb_dirichlet_neumann, which its Fdma cannot hold;
  // refused here)
  Navier2DAdjointEngine(int nx, int ny, double ra, double pr, double dt, double aspect, const std::string& bc, bool periodic);
  ~Navier2DAdjointEngine();

  void update(int nsteps);                          // Integrate::update, steady_adjoint.rs:541-608
  bool exit();                                      // steady_adjoint.rs:624-638: NaN divergence, or mean residual < RES_TOL
  double time() const { return time_; }
  double dt() const { return dt_; }
  void reset_time() { time_ = 0.0; }
  double param(const std::string& key) const;       // ra, pr, nu, ka
  double div_norm();                                // DivNorm::div_norm      steady_adjoint_eq.rs:40-42
  void norm_residual(double out[3]);                // DivNorm::norm_residual steady_adjoint_eq.rs:44-50

  void set_velocity(double amp, double m, double n);      // steady_adjoint.rs:183-186
  void set_temperature(double amp, double m, double n);   // steady_adjoint.rs:190-192
  // fields: velx vely temp pres pseu velx_adj vely_adj temp_adj pres_adj tempbc(read only)
  void spectral_shape(const std::string& name, int* rows, int* cols, int* elem);
  void set_field_spectral(const std::string& name, const double* host, size_t len);
  void get_field_spectral(const std::string& name, double* host, size_t len);
  void set_field_physical(const std::string& name, const double* host, size_t len);
  void get_field_physical(const std::string& name, double* host, size_t len);
  // Navier2DAdjoint::write / read (steady_adjoint_io.rs:48-71, 22-33) in the reference's HDF5 layout (csrc/h5lite): write =
  // ux, uy, temp, pres, tempbc as Field2 groups (x, dx, y, dy, v, vhat) + time + params; read = vhat of ux, uy, temp (other
  // resolutions by truncation / zero padding like field/io.rs:151-176) + time -- also from a snapshot a Navier2D run wrote
  // (examples/navier_rbc_steady.rs starts from one)
  void write(const std::string& filename);
  void read(const std::string& filename);
  void grid(int axis, double* x, size_t len) const;
  Stream& stream() { return st_; }

 private:
  struct F { Space2Ops* sp; Arr2 vhat; };
  F& field(const std::string& name);
  // out (ortho shape) (+)= s * to_ortho(f) / gradient(f)
  void acc_to_ortho(F& f, double s, Arr2& out);
  void acc_gradient(F& f, int d0, int d1, double s, Arr2& out);
  // conv (+)= s * u * backward(gradient(f, deriv))           functions.rs:56-69
  void conv_term(const Arr2& u, F& f, int d0, int d1, double s, bool first);
  void conv_finish(Arr2& out);                       // forward + dealias (functions.rs:72-82) of conv_ into `out` (ortho shape)
  void zero(Arr2& a);
  void lincomb(Arr2& out, double a, const Arr2& x, double b, const Arr2& y);   // out = a x + b y (y may alias out)
  void backward(F& f, Arr2& phys);
  void div(Arr2& out);
  void solve_pres(const Arr2& div);
  void correct_velocity(double c);
  double norm(const Arr2& a);

  int nx_, ny_, ex_;
  bool periodic_;
  double ra_, pr_, nu_, ka_, dt_, sx_, sy_, time_ = 0.0;
  Stream st_;
  std::unique_ptr<Space2Ops> sp_vel_, sp_temp_, sp_ortho_, sp_pseu_;
  std::unique_ptr<HholtzAdiOp> hh_vel_, hh_temp_;
  std::unique_ptr<PoissonOp> pois_;
  std::unique_ptr<TensorHholtzOp> norm_vel_, norm_temp_;
  std::map<std::string, F> f_;
  // work arrays: orthonormal-space shape (rhs_, div_, t0_..t2_, old_[3]), physical shape (ux_, uy_, ta_, ph_, conv_)
  Arr2 rhs_, div_, t0_, t1_, old_[3], ux_, uy_, ta_, ph_, conv_, cv_, cp_;
  DBuf red_;
};

}  // namespace rpde
