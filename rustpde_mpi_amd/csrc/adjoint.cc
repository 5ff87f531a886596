// Navier2DAdjointEngine: see adjoint.h.  Reference: src/navier_stokes/steady_adjoint.rs, steady_adjoint_eq.rs.
#include "adjoint.h"

#include <cmath>
#include <cstdio>
#include <sys/stat.h>

#include "h5lite.h"

namespace rpde {

// ------------------------------------------------------------------------------------------------
// element-wise kernels on pitched 2-D arrays of doubles (complex arrays: twice the columns)
#ifndef RPDE_EMU
// out = a x + b y  (x, y may alias out)
__global__ __launch_bounds__(256) void adj_lincomb_kernel(double* out, long ldo, double a, const double* x, long ldx, double b,
                                                          const double* y, long ldy, int rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    double v = 0.0;
    if (x) v += a * x[(long)r * ldx + c];
    if (y) v += b * y[(long)r * ldy + c];
    out[(long)r * ldo + c] = v;
  }
}
// out = (acc ? out : 0) + s x y
__global__ __launch_bounds__(256) void adj_muladd_kernel(double* out, long ldo, double s, const double* x, long ldx, const double* y,
                                                         long ldy, int rows, int cols, int acc) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const double v = s * x[(long)r * ldx + c] * y[(long)r * ldy + c];
    double* o = out + (long)r * ldo + c;
    *o = acc ? *o + v : v;
  }
}
// the 2/3 rule (functions.rs:72-82): rows >= r0 and columns >= c0 (in doubles) become zero
__global__ __launch_bounds__(256) void adj_dealias_kernel(double* a, long ld, int rows, int cols, int r0, int c0) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    if (r >= r0 || c >= c0) a[(long)r * ld + c] = 0.0;
}
static dim3 ew_grid(int rows, int cols) { return dim3((cols + 255) / 256, rows < 2048 ? rows : 2048); }
static void ew_lincomb(double* out, long ldo, double a, const double* x, long ldx, double b, const double* y, long ldy, int rows, int cols, Stream& st) {
  hipLaunchKernelGGL(adj_lincomb_kernel, ew_grid(rows, cols), dim3(256), 0, st.s, out, ldo, a, x, ldx, b, y, ldy, rows, cols);
  RPDE_HIP(hipGetLastError());
}
static void ew_muladd(double* out, long ldo, double s, const double* x, long ldx, const double* y, long ldy, int rows, int cols, bool acc, Stream& st) {
  hipLaunchKernelGGL(adj_muladd_kernel, ew_grid(rows, cols), dim3(256), 0, st.s, out, ldo, s, x, ldx, y, ldy, rows, cols, acc ? 1 : 0);
  RPDE_HIP(hipGetLastError());
}
static void ew_dealias(double* a, long ld, int rows, int cols, int r0, int c0, Stream& st) {
  hipLaunchKernelGGL(adj_dealias_kernel, ew_grid(rows, cols), dim3(256), 0, st.s, a, ld, rows, cols, r0, c0);
  RPDE_HIP(hipGetLastError());
}
#else
static void ew_lincomb(double* out, long ldo, double a, const double* x, long ldx, double b, const double* y, long ldy, int rows, int cols, Stream&) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      double v = 0.0;
      if (x) v += a * x[(long)r * ldx + c];
      if (y) v += b * y[(long)r * ldy + c];
      out[(long)r * ldo + c] = v;
    }
}
static void ew_muladd(double* out, long ldo, double s, const double* x, long ldx, const double* y, long ldy, int rows, int cols, bool acc, Stream&) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const double v = s * x[(long)r * ldx + c] * y[(long)r * ldy + c];
      double* o = out + (long)r * ldo + c;
      *o = acc ? *o + v : v;
    }
}
static void ew_dealias(double* a, long ld, int rows, int cols, int r0, int c0, Stream&) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      if (r >= r0 || c >= c0) a[(long)r * ld + c] = 0.0;
}
#endif

static double get_nu(double ra, double pr, double h) { return std::sqrt(pr / (ra / std::pow(h, 3.0))); }
static double get_ka(double ra, double pr, double h) { return std::sqrt(1.0 / ((ra / std::pow(h, 3.0)) * pr)); }

// ------------------------------------------------------------------------------------------------
GenericFlow2D::GenericFlow2D(int nx, int ny, double ra, double pr, double dt, double aspect, const std::string& bc, bool periodic,
                             double dt_helmholtz, std::initializer_list<const char*> extra_fields)
    : nx_(nx), ny_(ny), ex_(periodic ? 2 : 1), periodic_(periodic), ra_(ra), pr_(pr), dt_(dt), sx_(aspect), sy_(1.0) {
  RPDE_REQUIRE(bc == "rbc" || bc == "hc", "Boundary condition type \"" + bc + "\" not recognized!");
  hc_ = bc == "hc";
  RPDE_REQUIRE(nx >= 8 && ny >= 8, "grid too small");
  RPDE_REQUIRE(!periodic || nx % 2 == 0, "fourier_r2c needs an even number of points");
  RPDE_REQUIRE(dt > 0 && ra > 0 && pr > 0 && aspect > 0, "ra, pr, dt, aspect must be positive");
#ifndef RPDE_EMU
  RPDE_HIP(hipStreamCreate(&st_.s));
#endif
  nu_ = get_nu(ra, pr, sy_ * 2.0);
  ka_ = get_ka(ra, pr, sy_ * 2.0);
  const BaseKind bx_vel = periodic ? kFourierR2c : kChebDirichlet;
  const BaseKind bx_tmp = periodic ? kFourierR2c : kChebNeumann;
  const BaseKind bx_ort = periodic ? kFourierR2c : kChebyshev;
  sp_vel_ = std::make_unique<Space2Ops>(make_base(bx_vel, nx), make_base(kChebDirichlet, ny));
  // "hc" (lnse.rs:115-119, 202-206; nonlin.rs:117-121, 208-212): Dirichlet at the bottom, Neumann at the top -- the three-term base
  sp_temp_ = std::make_unique<Space2Ops>(make_base(bx_tmp, nx), make_base(hc_ ? kChebDirichletNeumann : kChebDirichlet, ny));
  sp_ortho_ = std::make_unique<Space2Ops>(make_base(bx_ort, nx), make_base(kChebyshev, ny));
  sp_pseu_ = std::make_unique<Space2Ops>(make_base(bx_tmp, nx), make_base(kChebNeumann, ny));
  hh_vel_ = std::make_unique<HholtzAdiOp>(*sp_vel_, dt_helmholtz * nu_ / (sx_ * sx_), dt_helmholtz * nu_ / (sy_ * sy_));
  hh_temp_ = std::make_unique<HholtzAdiOp>(*sp_temp_, dt_helmholtz * ka_ / (sx_ * sx_), dt_helmholtz * ka_ / (sy_ * sy_));
  pois_ = std::make_unique<PoissonOp>(*sp_pseu_, 1.0 / (sx_ * sx_), 1.0 / (sy_ * sy_));

  auto mk = [&](const std::string& name, Space2Ops* sp, bool ro = false) {
    F f{sp, Arr2(sp->spec_rows(), sp->spec_cols(), ex_), ro, name == "tempbc" || name.rfind("mean_", 0) == 0};
    f_.emplace(name, std::move(f));
  };
  mk("velx", sp_vel_.get()); mk("vely", sp_vel_.get()); mk("temp", sp_temp_.get());
  mk("pres", sp_ortho_.get()); mk("pseu", sp_pseu_.get()); mk("tempbc", sp_ortho_.get(), true);
  for (const char* name : extra_fields) {
    const std::string n = name;
    Space2Ops* sp = n.rfind("mean_", 0) == 0 ? sp_ortho_.get() : n.rfind("velx", 0) == 0 || n.rfind("vely", 0) == 0 ? sp_vel_.get()
                    : n.rfind("temp", 0) == 0 ? sp_temp_.get() : sp_ortho_.get();
    mk(n, sp);
  }

  const int ro = sp_ortho_->ortho_rows(), co = sp_ortho_->ortho_cols();
  for (Arr2* a : {&rhs_, &div_, &t0_, &t1_, &old_[0], &old_[1], &old_[2], &cv_}) a->alloc(ro, co, ex_);
  for (Arr2* a : {&ux_, &uy_, &ta_, &ph_, &conv_, &cp_}) a->alloc(nx, ny, 1);
  red_.alloc(2);

  {  // the lift bc_rbc (boundary_conditions.rs:18-36 / 143-161): T = +0.5 (bottom) ... -0.5 (top), forward transformed
    const Vec y = base_coords(sp_ortho_->base(1));
    const double x1 = y.front(), x2 = y.back(), y1 = 0.5, y2 = -0.5;
    const double m = (y2 - y1) / (x2 - x1), n = (y1 * x2 - y2 * x1) / (x2 - x1);
    Vec prof((size_t)nx * ny);
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j < ny; ++j) prof[(size_t)i * ny + j] = m * y[j] + n;
    if (hc_) prof = hc_profile();   // bc_hc (boundary_conditions.rs:96-134 / 163-202)
    dev_upload2d(ph_.p(), ph_.ld, prof.data(), nx, ny);
    sp_ortho_->forward(ph_, field("tempbc").vhat, st_);
  }
  dev_sync(st_);
}

// per x a parabola in y with its vertex (value 0, slope 0) at the top wall y[n-1] and the value -0.5 cos(2 pi (x - x0) / L) at the
// bottom wall y[0]: the lift bc_hc (boundary_conditions.rs:96-134 / 163-202) and the default mean temperature of "hc"
// (MeanFields::new_hc_confined / _periodic, meanfield.rs:52-86, 154-188)
Vec GenericFlow2D::hc_profile() const {
  const Vec x = base_coords(sp_ortho_->base(0)), y = base_coords(sp_ortho_->base(1));
  const double x0 = x.front(), length = x.back() - x.front(), d = y.front() - y.back();
  Vec prof((size_t)nx_ * ny_);
  for (int i = 0; i < nx_; ++i) {
    const double a = -0.5 * std::cos(2.0 * M_PI * (x[i] - x0) / length) / (d * d);
    for (int j = 0; j < ny_; ++j) prof[(size_t)i * ny_ + j] = a * (y[j] - y.back()) * (y[j] - y.back());
  }
  return prof;
}

static const std::string& adjoint_bc(const std::string& bc) {
  // bc = "rbc" ("hc": the reference builds its four-diagonal tensor solver Hholtz on the three-term base cheb_dirichlet_neumann,
  // steady_adjoint.rs:312-318, which its Fdma cannot hold; refused here by name)
  RPDE_REQUIRE(bc != "hc", "bc = \"hc\" is not supported by Navier2DAdjoint (the reference builds its four-diagonal tensor solver Hholtz on "
                           "the three-term base cheb_dirichlet_neumann, steady_adjoint.rs:312-318, which Fdma cannot hold)");
  return bc;
}

Navier2DAdjointEngine::Navier2DAdjointEngine(int nx, int ny, double ra, double pr, double dt, double aspect,
                                             const std::string& bc, bool periodic)
    // the Helmholtz solvers of the FORWARD step run on DT_NAVIER (steady_adjoint.rs:273-295)
    : GenericFlow2D(nx, ny, ra, pr, dt, aspect, adjoint_bc(bc), periodic, kDtNavier, {"velx_adj", "vely_adj", "temp_adj", "pres_adj"}) {
  // smoother (1 - weight D2) (steady_adjoint.rs:300-322): velx and vely live in the same space -> one solver serves both
  norm_vel_ = std::make_unique<TensorHholtzOp>(*sp_vel_, kWeightLaplacian / (sx_ * sx_), kWeightLaplacian / (sy_ * sy_));
  norm_temp_ = std::make_unique<TensorHholtzOp>(*sp_temp_, kWeightLaplacian / (sx_ * sx_), kWeightLaplacian / (sy_ * sy_));
  dev_sync(st_);
  const char* e = std::getenv("RPDE_ADJOINT_FUSED");
  if (!e || std::atoi(e) != 0)
    fwd_ = std::make_unique<Navier2DEngine>(nx, ny, ra, pr, kDtNavier, aspect, bc, periodic, nullptr, /*buoyancy_lift=*/false);
}

GenericFlow2D::~GenericFlow2D() {
#ifndef RPDE_EMU
  if (st_.s) { (void)hipStreamSynchronize(st_.s); (void)hipStreamDestroy(st_.s); }
#endif
}

double GenericFlow2D::param(const std::string& key) const {
  if (key == "ra") return ra_;
  if (key == "pr") return pr_;
  if (key == "nu") return nu_;
  if (key == "ka") return ka_;
  fail("unknown parameter " + key);
}

GenericFlow2D::F& GenericFlow2D::field(const std::string& name) {
  auto it = f_.find(name);
  RPDE_REQUIRE(it != f_.end(), "unknown field " + name);
  return it->second;
}

void GenericFlow2D::spectral_shape(const std::string& name, int* rows, int* cols, int* elem) {
  F& f = field(name);
  *rows = f.vhat.rows; *cols = f.vhat.cols; *elem = f.vhat.elem;
}

void GenericFlow2D::set_field_spectral(const std::string& name, const double* host, size_t len) {
  F& f = field(name);
  RPDE_REQUIRE(!f.read_only, name + " is fixed by the boundary condition");
  RPDE_REQUIRE(len == (size_t)f.vhat.rows * f.vhat.cols * f.vhat.elem, "set_field: wrong length for the spectral shape of " + name);
  if (f.constant) drop_constant_gradients();
  dev_sync(st_);
  dev_upload2d(f.vhat.p(), f.vhat.ld, host, f.vhat.rows, (long)f.vhat.cols * f.vhat.elem);
}

void GenericFlow2D::get_field_spectral(const std::string& name, double* host, size_t len) {
  F& f = field(name);
  RPDE_REQUIRE(len == (size_t)f.vhat.rows * f.vhat.cols * f.vhat.elem, "get_field: wrong length for the spectral shape of " + name);
  dev_sync(st_);
  dev_download2d(host, f.vhat.p(), f.vhat.ld, f.vhat.rows, (long)f.vhat.cols * f.vhat.elem);
}

void GenericFlow2D::set_field_physical(const std::string& name, const double* host, size_t len) {
  F& f = field(name);
  RPDE_REQUIRE(!f.read_only, name + " is fixed by the boundary condition");
  RPDE_REQUIRE(len == (size_t)nx_ * ny_, "set_field: physical arrays are nx*ny doubles");
  if (f.constant) drop_constant_gradients();
  dev_sync(st_);
  dev_upload2d(ph_.p(), ph_.ld, host, nx_, ny_);
  f.sp->forward(ph_, f.vhat, st_);
  dev_sync(st_);
}

void GenericFlow2D::get_field_physical(const std::string& name, double* host, size_t len) {
  F& f = field(name);
  RPDE_REQUIRE(len == (size_t)nx_ * ny_, "get_field: physical arrays are nx*ny doubles");
  f.sp->backward(f.vhat, ph_, st_);
  dev_sync(st_);
  dev_download2d(host, ph_.p(), ph_.ld, nx_, ny_);
}

static void sincos_field(const Base& b0, const Base& b1, double sx, double sy, double amp, double m, double n, bool sin_cos, Vec& out) {
  // functions.rs:85-126 (coordinates normalised by x[last] - x[0])
  Vec x = base_coords(b0), y = base_coords(b1);
  for (double& v : x) v *= sx;
  for (double& v : y) v *= sy;
  const double x0 = x.front(), xl = x.back() - x.front(), y0 = y.front(), yl = y.back() - y.front();
  out.resize(x.size() * y.size());
  for (size_t i = 0; i < x.size(); ++i)
    for (size_t j = 0; j < y.size(); ++j) {
      const double xa = M_PI * m * ((x[i] - x0) / xl), ya = M_PI * n * ((y[j] - y0) / yl);
      out[i * y.size() + j] = sin_cos ? amp * std::sin(xa) * std::cos(ya) : amp * std::cos(xa) * std::sin(ya);
    }
}

void GenericFlow2D::set_velocity(double amp, double m, double n) {
  Vec v;
  sincos_field(sp_vel_->base(0), sp_vel_->base(1), sx_, sy_, amp, m, n, true, v);
  set_field_physical("velx", v.data(), v.size());
  sincos_field(sp_vel_->base(0), sp_vel_->base(1), sx_, sy_, -amp, m, n, false, v);
  set_field_physical("vely", v.data(), v.size());
}

void GenericFlow2D::set_temperature(double amp, double m, double n) {
  Vec v;
  sincos_field(sp_temp_->base(0), sp_temp_->base(1), sx_, sy_, -amp, m, n, false, v);
  set_field_physical("temp", v.data(), v.size());
}

// ------------------------------------------------------------------------------------------------
void GenericFlow2D::grid(int axis, double* x, size_t len) const {
  const Base& b = sp_vel_->base(axis);
  RPDE_REQUIRE((int)len == b.n, "grid: wrong length");
  const Vec c = base_coords(b);
  const double sc = axis == 0 ? sx_ : sy_;
  for (int i = 0; i < b.n; ++i) x[i] = c[i] * sc;
}

static const char* const kAdjSnapFields[5][2] = {{"velx", "ux"}, {"vely", "uy"}, {"temp", "temp"}, {"pres", "pres"}, {"tempbc", "tempbc"}};

void GenericFlow2D::write(const std::string& filename) {
  h5::Tree t;
  Vec x((size_t)nx_), y((size_t)ny_);
  grid(0, x.data(), x.size());
  grid(1, y.data(), y.size());
  for (const auto& fg : kAdjSnapFields) {
    const std::string g = fg[1];
    t[g + "/x"] = h5::Dataset{{(uint64_t)nx_}, x};       // field/io.rs:96-99: `dx` / `dy` are written from the coordinates
    t[g + "/dx"] = h5::Dataset{{(uint64_t)nx_}, x};
    t[g + "/y"] = h5::Dataset{{(uint64_t)ny_}, y};
    t[g + "/dy"] = h5::Dataset{{(uint64_t)ny_}, y};
    h5::Dataset v{{(uint64_t)nx_, (uint64_t)ny_}, Vec((size_t)nx_ * ny_)};
    get_field_physical(fg[0], v.data.data(), v.data.size());
    t[g + "/v"] = std::move(v);
    int r, c, e;
    spectral_shape(fg[0], &r, &c, &e);
    Vec vh((size_t)r * c * e);
    get_field_spectral(fg[0], vh.data(), vh.size());
    if (e == 1) {
      t[g + "/vhat"] = h5::Dataset{{(uint64_t)r, (uint64_t)c}, std::move(vh)};
    } else {   // read_write_hdf5.rs:171-188: complex arrays as two real datasets
      h5::Dataset re{{(uint64_t)r, (uint64_t)c}, Vec((size_t)r * c)}, im = re;
      for (size_t k = 0; k < (size_t)r * c; ++k) { re.data[k] = vh[2 * k]; im.data[k] = vh[2 * k + 1]; }
      t[g + "/vhat_re"] = std::move(re);
      t[g + "/vhat_im"] = std::move(im);
    }
  }
  t["time"] = h5::Dataset{{1}, {time_}};
  for (const char* k : {"ra", "pr", "nu", "ka"}) t[k] = h5::Dataset{{1}, {param(k)}};
  h5::update_file(filename, t);
}

void GenericFlow2D::read(const std::string& filename) {
  h5::Reader rd(filename);
  for (int k = 0; k < 3; ++k) {    // steady_adjoint_io.rs:23-25: ux, uy, temp
    const std::string name = kAdjSnapFields[k][0], g = kAdjSnapFields[k][1];
    int r, c, e;
    spectral_shape(name, &r, &c, &e);
    Vec neu((size_t)r * c * e, 0.0);
    uint64_t ro = 0, co = 0;
    auto place = [&](const h5::Dataset& d, int comp) {
      RPDE_REQUIRE(d.dims.size() == 2, "snapshot: " + g + "/vhat must be two-dimensional");
      ro = d.dims[0]; co = d.dims[1];
      const uint64_t rm = std::min<uint64_t>(ro, r), cm = std::min<uint64_t>(co, c);
      for (uint64_t i = 0; i < rm; ++i)
        for (uint64_t j = 0; j < cm; ++j) neu[(i * c + j) * e + comp] = d.data[i * co + j];
    };
    if (e == 1) place(rd.read(g + "/vhat"), 0);
    else { place(rd.read(g + "/vhat_re"), 0); place(rd.read(g + "/vhat_im"), 1); }
    if (((int)ro != r || (int)co != c) && periodic_) {   // field/io.rs:167-175: the unnormalised Fourier coefficients scale with the number of points
      const double norm = (double)(r - 1) / (double)(ro - 1);
      for (double& v : neu) v *= norm;
    }
    set_field_spectral(name, neu.data(), neu.size());
  }
  time_ = rd.read("time").data.at(0);
}

// ------------------------------------------------------------------------------------------------
void GenericFlow2D::zero(Arr2& a) { dev_zero(a.p(), a.bytes(), st_); }

void GenericFlow2D::lincomb(Arr2& out, double a, const Arr2& x, double b, const Arr2& y) {
  RPDE_REQUIRE(out.rows == x.rows && out.cols == x.cols && out.elem == x.elem && y.rows == x.rows && y.cols == x.cols && y.elem == x.elem,
               "lincomb: shapes differ");
  ew_lincomb(out.p(), out.ld, a, x.p(), x.ld, b, y.p(), y.ld, out.rows, out.cols * out.elem, st_);
}

void GenericFlow2D::acc_to_ortho(F& f, double s, Arr2& out) {
  f.sp->to_ortho(f.vhat, t0_, st_);
  lincomb(out, 1.0, out, s, t0_);
}

void GenericFlow2D::drop_constant_gradients() {
  ++const_gen_;
  if (const_grad_.empty()) return;
  dev_sync(st_);
  const_grad_.clear();
}

void GenericFlow2D::acc_gradient(F& f, int d0, int d1, double s, Arr2& out) {
  if (f.constant) {
    auto it = const_grad_.find({&f, 100 + d0, d1});
    if (it == const_grad_.end()) {
      it = const_grad_.emplace(std::make_tuple(&f, 100 + d0, d1), Arr2(out.rows, out.cols, out.elem)).first;
      f.sp->gradient(f.vhat, d0, d1, sx_, sy_, it->second, st_);
    }
    lincomb(out, 1.0, out, s, it->second);
    return;
  }
  f.sp->gradient(f.vhat, d0, d1, sx_, sy_, t0_, st_);
  lincomb(out, 1.0, out, s, t0_);
}

void GenericFlow2D::backward(F& f, Arr2& phys) { f.sp->backward(f.vhat, phys, st_); }

void GenericFlow2D::conv_term(const Arr2& u, F& f, int d0, int d1, double s, bool first) {
  const Arr2* g = &cp_;
  if (f.constant) {   // the lift, the mean fields: the physical gradient is the same in every term and step of a run
    auto it = const_grad_.find({&f, d0, d1});
    if (it == const_grad_.end()) {
      it = const_grad_.emplace(std::make_tuple(&f, d0, d1), Arr2(nx_, ny_, 1)).first;
      f.sp->gradient_backward(f.vhat, d0, d1, sx_, sy_, it->second, st_);
    }
    g = &it->second;
  } else {
    f.sp->gradient_backward(f.vhat, d0, d1, sx_, sy_, cp_, st_);   // backward_ortho(gradient(..)): one line program per axis
  }
  ew_muladd(conv_.p(), conv_.ld, s, u.p(), u.ld, g->p(), g->ld, nx_, ny_, !first, st_);
}

void GenericFlow2D::conv_finish(Arr2& out) {
  sp_ortho_->forward(conv_, out, st_);
  ew_dealias(out.p(), out.ld, out.rows, out.cols * out.elem, out.rows * 2 / 3, (out.cols * 2 / 3) * out.elem, st_);
}

void GenericFlow2D::div(Arr2& out) {   // steady_adjoint_eq.rs:19-24
  F &u = field("velx"), &v = field("vely");
  u.sp->gradient(u.vhat, 1, 0, sx_, sy_, out, st_);
  v.sp->gradient(v.vhat, 0, 1, sx_, sy_, t0_, st_);
  lincomb(out, 1.0, out, 1.0, t0_);
}

void GenericFlow2D::solve_pres(const Arr2& d) {   // steady_adjoint_eq.rs:226-231
  F& ps = field("pseu");
  pois_->solve(d, ps.vhat, st_);
  launch_set_element(ps.vhat.p(), 0, 0.0, st_);
  if (ex_ == 2) launch_set_element(ps.vhat.p(), 1, 0.0, st_);
}

void GenericFlow2D::correct_velocity(double c) {   // steady_adjoint_eq.rs:183-192
  F &ps = field("pseu"), &u = field("velx"), &v = field("vely");
  if (cvt_.rows != u.vhat.rows || cvt_.cols != u.vhat.cols) cvt_.alloc(u.vhat.rows, u.vhat.cols, ex_);   // once: no allocation inside the step
  ps.sp->gradient(ps.vhat, 1, 0, sx_, sy_, t0_, st_);
  u.sp->from_ortho(t0_, cvt_, st_);
  lincomb(u.vhat, 1.0, u.vhat, -c, cvt_);
  ps.sp->gradient(ps.vhat, 0, 1, sx_, sy_, t0_, st_);
  v.sp->from_ortho(t0_, cvt_, st_);
  lincomb(v.vhat, 1.0, v.vhat, -c, cvt_);
}

double GenericFlow2D::norm(const Arr2& a) {   // functions.rs:24-35
  launch_sumsq(a.p(), a.ld, a.rows, a.cols * a.elem, red_.p, st_);
  dev_sync(st_);
  double h[2];
  dev_download(h, red_.p, sizeof(h));
  if (h[1] > 0) return std::nan("");
  return std::sqrt(h[0]);
}

double GenericFlow2D::div_norm() {
  div(div_);
  return norm(div_);
}

void Navier2DAdjointEngine::norm_residual(double out[3]) {
  out[0] = norm(field("velx_adj").vhat);
  out[1] = norm(field("vely_adj").vhat);
  out[2] = norm(field("temp_adj").vhat);
}

bool Navier2DAdjointEngine::exit() {
  if (std::isnan(div_norm())) return true;
  double r[3];
  norm_residual(r);
  return (r[0] + r[1] + r[2]) / 3.0 < kResTol;
}

// The forward Navier-Stokes step of Navier2DAdjoint::update as a composition of the generic operators (rounds 5; kept as the A/B
// form, RPDE_ADJOINT_FUSED=0).  old_[0 .. 2] hold the orthonormal coefficients of the state before the step.
void Navier2DAdjointEngine::forward_step_generic() {
  F &velx = field("velx"), &vely = field("vely"), &temp = field("temp"), &pres = field("pres"), &pseu = field("pseu");
  F& tempbc = field("tempbc");
  const double dtn = kDtNavier;
  backward(velx, ux_);
  backward(vely, uy_);
  // solve_velx (steady_adjoint_eq.rs:134-144)
  lincomb(rhs_, 1.0, old_[0], 0.0, old_[0]);
  acc_gradient(pres, 1, 0, -dtn, rhs_);
  conv_term(ux_, velx, 1, 0, 1.0, true);
  conv_term(uy_, velx, 0, 1, 1.0, false);
  conv_finish(cv_);
  lincomb(rhs_, 1.0, rhs_, -dtn, cv_);
  hh_vel_->solve(rhs_, velx.vhat, st_);
  // solve_vely (steady_adjoint_eq.rs:147-160): buoyancy = temp.to_ortho() * dt, without the lift
  lincomb(rhs_, 1.0, old_[1], dtn, old_[2]);
  acc_gradient(pres, 0, 1, -dtn, rhs_);
  conv_term(ux_, vely, 1, 0, 1.0, true);
  conv_term(uy_, vely, 0, 1, 1.0, false);
  conv_finish(cv_);
  lincomb(rhs_, 1.0, rhs_, -dtn, cv_);
  hh_vel_->solve(rhs_, vely.vhat, st_);
  // projection (steady_adjoint.rs:563-567)
  div(div_);
  solve_pres(div_);
  correct_velocity(1.0);
  // update_pres (steady_adjoint_eq.rs:194-201): pres += -nu div + pseu.to_ortho() / dt
  lincomb(pres.vhat, 1.0, pres.vhat, -nu_, div_);
  acc_to_ortho(pseu, 1.0 / dtn, pres.vhat);
  // solve_temp (steady_adjoint_eq.rs:166-180)
  lincomb(rhs_, 1.0, old_[2], 0.0, old_[2]);
  acc_gradient(tempbc, 2, 0, dtn * ka_, rhs_);
  acc_gradient(tempbc, 0, 2, dtn * ka_, rhs_);
  conv_term(ux_, temp, 1, 0, 1.0, true);
  conv_term(uy_, temp, 0, 1, 1.0, false);
  conv_term(ux_, tempbc, 1, 0, 1.0, false);
  conv_term(uy_, tempbc, 0, 1, 1.0, false);
  conv_finish(cv_);
  lincomb(rhs_, 1.0, rhs_, -dtn, cv_);
  hh_temp_->solve(rhs_, temp.vhat, st_);
}

// The same step on Navier2DEngine's fused schedule: the state goes in as device arrays (canonical -> the engine's YX layout,
// d/dx p and d/dy p refreshed), one update(), and u, v, T, p and the pseudo-pressure come back.
void Navier2DAdjointEngine::forward_step_fused() {
  dev_sync(st_);                                 // the engine runs on its own stream
  for (const char* name : {"velx", "vely", "temp", "pres"}) fwd_->set_field_spectral_device(name, field(name).vhat);
  fwd_->update(1);
  for (const char* name : {"velx", "vely", "temp", "pres", "pseu"}) fwd_->get_field_spectral_device(name, field(name).vhat);
  fwd_->sync();
}

// ------------------------------------------------------------------------------------------------
void Navier2DAdjointEngine::update(int nsteps) {
  F &velx = field("velx"), &vely = field("vely"), &temp = field("temp"), &pres = field("pres"), &pseu = field("pseu");
  F &velx_adj = field("velx_adj"), &vely_adj = field("vely_adj"), &temp_adj = field("temp_adj"), &pres_adj = field("pres_adj");
  F& tempbc = field("tempbc");
  const double dtn = kDtNavier, dt = dt_;
  for (int step = 0; step < nsteps; ++step) {
    // ================= forward step for the residual (steady_adjoint.rs:547-585) =================
    velx.sp->to_ortho(velx.vhat, old_[0], st_);
    vely.sp->to_ortho(vely.vhat, old_[1], st_);
    temp.sp->to_ortho(temp.vhat, old_[2], st_);
    if (fwd_) forward_step_fused(); else forward_step_generic();
    // residual (steady_adjoint.rs:572-575) in the norm of solver_norm, sign flipped (577-584)
    velx.sp->to_ortho(velx.vhat, t1_, st_);
    lincomb(t1_, -1.0 / dtn, t1_, 1.0 / dtn, old_[0]);      // -(new - old) / dt
    norm_vel_->solve(t1_, velx_adj.vhat, st_);
    vely.sp->to_ortho(vely.vhat, t1_, st_);
    lincomb(t1_, -1.0 / dtn, t1_, 1.0 / dtn, old_[1]);
    norm_vel_->solve(t1_, vely_adj.vhat, st_);
    temp.sp->to_ortho(temp.vhat, t1_, st_);
    lincomb(t1_, -1.0 / dtn, t1_, 1.0 / dtn, old_[2]);
    norm_temp_->solve(t1_, temp_adj.vhat, st_);
    // ================= adjoint step (steady_adjoint.rs:588-607) =================
    backward(velx, ux_);
    backward(vely, uy_);
    backward(temp_adj, ta_);
    // solve_velx_adj (steady_adjoint_eq.rs:338-359)
    zero(rhs_);
    acc_to_ortho(velx, 1.0, rhs_);
    acc_gradient(pres_adj, 1, 0, -dt, rhs_);
    conv_term(ux_, velx_adj, 1, 0, 1.0, true);     // conv_velx_adjoint (steady_adjoint_eq.rs:236-262)
    conv_term(uy_, velx_adj, 0, 1, 1.0, false);
    conv_term(ux_, velx_adj, 1, 0, 1.0, false);
    conv_term(uy_, vely_adj, 1, 0, 1.0, false);
    conv_term(ta_, temp, 1, 0, -1.0, false);
    conv_term(ta_, tempbc, 1, 0, -1.0, false);
    conv_finish(cv_);
    lincomb(rhs_, 1.0, rhs_, dt, cv_);
    acc_gradient(velx_adj, 2, 0, dt * nu_, rhs_);
    acc_gradient(velx_adj, 0, 2, dt * nu_, rhs_);
    // (vely's right-hand side needs velx.to_ortho() no more, but conv_vely_adjoint reads velx_adj / vely_adj only: velx may change now)
    velx.sp->from_ortho(rhs_, velx.vhat, st_);
    // solve_vely_adj (steady_adjoint_eq.rs:362-388)
    zero(rhs_);
    acc_to_ortho(vely, 1.0, rhs_);
    acc_gradient(pres_adj, 0, 1, -dt, rhs_);
    conv_term(ux_, vely_adj, 1, 0, 1.0, true);     // conv_vely_adjoint (steady_adjoint_eq.rs:265-293)
    conv_term(uy_, vely_adj, 0, 1, 1.0, false);
    conv_term(ux_, velx_adj, 0, 1, 1.0, false);
    conv_term(uy_, vely_adj, 0, 1, 1.0, false);
    conv_term(ta_, temp, 0, 1, -1.0, false);
    conv_term(ta_, tempbc, 0, 1, -1.0, false);
    conv_finish(cv_);
    lincomb(rhs_, 1.0, rhs_, dt, cv_);
    acc_gradient(vely_adj, 2, 0, dt * nu_, rhs_);
    acc_gradient(vely_adj, 0, 2, dt * nu_, rhs_);
    vely.sp->from_ortho(rhs_, vely.vhat, st_);
    // projection (steady_adjoint.rs:598-602); update_pres_adj (steady_adjoint_eq.rs:203-210)
    div(div_);
    solve_pres(div_);
    correct_velocity(1.0);
    acc_to_ortho(pseu, 1.0 / dt, pres_adj.vhat);
    // solve_temp_adj (steady_adjoint_eq.rs:395-424)
    zero(rhs_);
    acc_to_ortho(temp, 1.0, rhs_);
    conv_term(ux_, temp_adj, 1, 0, 1.0, true);     // conv_temp_adjoint (steady_adjoint_eq.rs:296-314)
    conv_term(uy_, temp_adj, 0, 1, 1.0, false);
    conv_finish(cv_);
    lincomb(rhs_, 1.0, rhs_, dt, cv_);
    acc_to_ortho(vely_adj, dt, rhs_);
    acc_gradient(temp_adj, 2, 0, dt * ka_, rhs_);
    acc_gradient(temp_adj, 0, 2, dt * ka_, rhs_);
    temp.sp->from_ortho(rhs_, temp.vhat, st_);
    time_ += dt_;
  }
  dev_sync(st_);
}

// ================================================================================================
// Navier2DLnse (src/navier_stokes_lnse/lnse.rs, lnse_eq.rs, meanfield.rs)
Navier2DLnseEngine::Navier2DLnseEngine(int nx, int ny, double ra, double pr, double dt, double aspect, const std::string& bc,
                                       bool periodic, const std::string& mean_file, bool nonlinear)
    : GenericFlow2D(nx, ny, ra, pr, dt, aspect, bc, periodic, dt, {"mean_velx", "mean_vely", "mean_temp"}), nonlin_(nonlinear) {
  um_.alloc(nx, ny, 1); vm_.alloc(nx, ny, 1); tp_.alloc(nx, ny, 1);
  if (nonlin_) { unl_.alloc(nx, ny, 1); vnl_.alloc(nx, ny, 1); }
  bool from_file = false;
  if (!mean_file.empty()) {
    if (FILE* f = std::fopen(mean_file.c_str(), "rb")) { std::fclose(f); from_file = true; }
  }
  if (from_file) {
    // MeanFields::read (meanfield.rs:237-259): the PHYSICAL arrays of a snapshot, the lift added to the temperature
    h5::Reader rd(mean_file);
    const char* const grp[3][2] = {{"velx", "ux/v"}, {"vely", "uy/v"}, {"temp", "temp/v"}};
    for (const auto& g : grp) {
      h5::Dataset d = rd.read(g[1]);
      RPDE_REQUIRE(d.dims.size() == 2 && (int)d.dims[0] == nx && (int)d.dims[1] == ny, std::string("mean field ") + g[1] + ": shape differs from the grid");
      if (std::string(g[0]) == "temp") {
        try {
          const h5::Dataset b = rd.read("tempbc/v");
          if (b.data.size() == d.data.size()) for (size_t i = 0; i < d.data.size(); ++i) d.data[i] += b.data[i];
        } catch (const std::exception&) {}      // "if let Ok(x)": no lift in the file
      }
      set_mean_physical(g[0], d.data.data(), d.data.size());
    }
  } else {
    // MeanFields::new_rbc_confined / _periodic (meanfield.rs:28-49, 130-151): no mean flow, the conduction profile
    std::printf("File \"%s\" does not exist. Use \"%s\" meanfield.\n", mean_file.c_str(), bc.c_str());
    const Vec y = base_coords(sp_ortho_->base(1));
    const double height = y.back() - y.front();
    Vec prof((size_t)nx * ny);
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j < ny; ++j) prof[(size_t)i * ny + j] = -(y[j] - y.front()) / height + 0.5;
    if (hc_) prof = hc_profile();   // MeanFields::new_hc_confined / _periodic (meanfield.rs:52-86, 154-188)
    set_mean_physical("temp", prof.data(), prof.size());
  }
  refresh_mean();
  dev_sync(st_);
  const char* e = std::getenv("RPDE_LNSE_FUSED");
  if (!hc_ && (!e || std::atoi(e) != 0)) {
    try {
      fwd_ = std::make_unique<Navier2DEngine>(nx, ny, ra, pr, dt, aspect, bc, periodic, nullptr, /*buoyancy_lift=*/false, /*lnse=*/nonlin_ ? 2 : 1);
    } catch (const std::exception& ex) {
      fwd_.reset();   // a shape the fused schedule does not cover: the generic composition
      // (said aloud when the shape IS one the whole-line convection kernel covers: then something else refused, e.g. a switch)
      if (ny == 1025 || ny == 2049 || ny == 4097)
        std::fprintf(stderr, "Navier2DLnse: the fused schedule is not used (%s); composition of generic operators\n", ex.what());
    }
    if (fwd_ && !nonlin_) {
      try {
        adj_ = std::make_unique<Navier2DEngine>(nx, ny, ra, pr, dt, aspect, bc, periodic, nullptr, /*buoyancy_lift=*/false, /*lnse=*/3);
      } catch (const std::exception&) {
        adj_.reset();
      }
    }
  }
}

void Navier2DLnseEngine::set_mean_physical(const std::string& name, const double* host, size_t len) {
  RPDE_REQUIRE(name == "velx" || name == "vely" || name == "temp", "mean field: velx, vely or temp");
  set_field_physical("mean_" + name, host, len);
  refresh_mean();
}
void Navier2DLnseEngine::get_mean_physical(const std::string& name, double* host, size_t len) {
  RPDE_REQUIRE(name == "velx" || name == "vely" || name == "temp", "mean field: velx, vely or temp");
  get_field_physical("mean_" + name, host, len);
}
void Navier2DLnseEngine::refresh_mean() {
  backward(mean("velx"), um_);
  backward(mean("vely"), vm_);
}

void Navier2DLnseEngine::conv_lin(F& mean_f, F& f, Arr2& out) {   // lnse_eq.rs:59-110; nonlin_eq.rs:59-134
  conv_term(ux_, mean_f, 1, 0, 1.0, true);      // ux dM/dx + uy dM/dy
  conv_term(uy_, mean_f, 0, 1, 1.0, false);
  conv_term(um_, f, 1, 0, 1.0, false);          // U df/dx + V df/dy
  conv_term(vm_, f, 0, 1, 1.0, false);
  if (nonlin_) {
    conv_term(ux_, f, 1, 0, 1.0, false);        // ux df/dx + uy df/dy
    conv_term(uy_, f, 0, 1, 1.0, false);
    conv_term(um_, mean_f, 1, 0, 1.0, false);   // U dM/dx + V dM/dy
    conv_term(vm_, mean_f, 0, 1, 1.0, false);
  }
  conv_finish(out);
}

void Navier2DLnseEngine::mean_diffusion(F& mean_f, double kappa) {
  acc_gradient(mean_f, 2, 0, dt_ * kappa, rhs_);
  acc_gradient(mean_f, 0, 2, dt_ * kappa, rhs_);
}

void Navier2DLnseEngine::update_direct(int nsteps) {
  if (!nonlin_) { update(nsteps); return; }
  for (int step = 0; step < nsteps; ++step) {
    update(1);
    // nonlin_adj_grad.rs:65-77: clones of velx, vely, temp (after backward()) pushed onto field_history
    auto h = std::make_unique<Hist>();
    const char* const names[3] = {"velx", "vely", "temp"};
    F* dst[3] = {&h->velx, &h->vely, &h->temp};
    for (int k = 0; k < 3; ++k) {
      F& f = field(names[k]);
      dst[k]->sp = f.sp;
      dst[k]->vhat.alloc(f.vhat.rows, f.vhat.cols, ex_);
      lincomb(dst[k]->vhat, 1.0, f.vhat, 0.0, f.vhat);
    }
    hist_.push_back(std::move(h));
  }
  dev_sync(st_);
}

void Navier2DLnseEngine::write(const std::string& filename) {
  GenericFlow2D::write(filename);
  if (!nonlin_) return;
  h5::Tree t;
  Vec x((size_t)nx_), y((size_t)ny_);
  grid(0, x.data(), x.size());
  grid(1, y.data(), y.size());
  const char* const grp[3][2] = {{"mean_velx", "ux_base"}, {"mean_vely", "uy_base"}, {"mean_temp", "temp_base"}};
  for (const auto& fg : grp) {
    const std::string g = fg[1];
    t[g + "/x"] = h5::Dataset{{(uint64_t)nx_}, x};
    t[g + "/dx"] = h5::Dataset{{(uint64_t)nx_}, x};
    t[g + "/y"] = h5::Dataset{{(uint64_t)ny_}, y};
    t[g + "/dy"] = h5::Dataset{{(uint64_t)ny_}, y};
    h5::Dataset v{{(uint64_t)nx_, (uint64_t)ny_}, Vec((size_t)nx_ * ny_)};
    get_field_physical(fg[0], v.data.data(), v.data.size());
    t[g + "/v"] = std::move(v);
    int r, c, e;
    spectral_shape(fg[0], &r, &c, &e);
    Vec vh((size_t)r * c * e);
    get_field_spectral(fg[0], vh.data(), vh.size());
    if (e == 1) {
      t[g + "/vhat"] = h5::Dataset{{(uint64_t)r, (uint64_t)c}, std::move(vh)};
    } else {
      h5::Dataset re{{(uint64_t)r, (uint64_t)c}, Vec((size_t)r * c)}, im = re;
      for (size_t k = 0; k < (size_t)r * c; ++k) { re.data[k] = vh[2 * k]; im.data[k] = vh[2 * k + 1]; }
      t[g + "/vhat_re"] = std::move(re);
      t[g + "/vhat_im"] = std::move(im);
    }
  }
  h5::update_file(filename, t);
}

bool Navier2DLnseEngine::exit() { return std::isnan(div_norm()); }

// update() on Navier2DEngine's schedule (adjoint.h fwd_): the mean velocities and the six physical mean gradients once per change of a
// mean field, the state per call
void Navier2DLnseEngine::push_mean(Navier2DEngine& e, bool negate_velocities) {
  refresh_mean();
  if (negate_velocities) {   // the adjoint term carries -(U d/dx + V d/dy) f (engine.h lnse = 3)
    lincomb(cp_, -1.0, um_, 0.0, um_); dev_sync(st_); e.set_lnse_mean_device(0, cp_);
    lincomb(cp_, -1.0, vm_, 0.0, vm_); dev_sync(st_); e.set_lnse_mean_device(1, cp_);
  } else {
    e.set_lnse_mean_device(0, um_);
    e.set_lnse_mean_device(1, vm_);
  }
  const char* const names[3] = {"velx", "vely", "temp"};
  for (int f = 0; f < 3; ++f)
    for (int d = 0; d < 2; ++d) {
      F& m = mean(names[f]);
      m.sp->gradient_backward(m.vhat, d == 0 ? 1 : 0, d == 0 ? 0 : 1, sx_, sy_, cp_, st_);
      dev_sync(st_);
      e.set_lnse_mean_device(2 + 2 * f + d, cp_);
    }
}

void Navier2DLnseEngine::update_fused(int nsteps) {
  if (fwd_mean_gen_ != const_gen_) {
    push_mean(*fwd_, false);
    const char* const names[3] = {"velx", "vely", "temp"};
    if (nonlin_) {
      // what the mean fields add to the right-hand sides of Navier2DNonLin (nonlin_eq.rs:193-240, nonlin.rs:266): dt nu lap(U);
      // dt nu lap(V) + dt mean.temp.to_ortho() (the buoyancy of the mean temperature); dt ka lap(T).  The Helmholtz solves are linear:
      // H^-1 of these constant rows is added behind the solves of the fused step (engine.h, fields nl_velx / nl_vely / nl_temp)
      const char* const dst[3] = {"nl_velx", "nl_vely", "nl_temp"};
      for (int f = 0; f < 3; ++f) {
        F& st = field(names[f]);
        zero(rhs_);
        mean_diffusion(mean(names[f]), f == 2 ? ka_ : nu_);
        if (f == 1) lincomb(rhs_, 1.0, rhs_, dt_, mean("temp").vhat);
        Arr2 c(st.vhat.rows, st.vhat.cols, st.vhat.elem);
        (f == 2 ? *hh_temp_ : *hh_vel_).solve(rhs_, c, st_);
        dev_sync(st_);
        fwd_->set_field_spectral_device(dst[f], c);
      }
    }
    fwd_mean_gen_ = const_gen_;
  }
  dev_sync(st_);
  for (const char* name : {"velx", "vely", "temp", "pres"}) fwd_->set_field_spectral_device(name, field(name).vhat);
  fwd_->update(nsteps);
  for (const char* name : {"velx", "vely", "temp", "pres", "pseu"}) fwd_->get_field_spectral_device(name, field(name).vhat);
  fwd_->sync();
  time_ += nsteps * dt_;
}

void Navier2DLnseEngine::update(int nsteps) {
  if (fwd_ && nsteps > 0) { update_fused(nsteps); return; }
  F &velx = field("velx"), &vely = field("vely"), &temp = field("temp"), &pres = field("pres"), &pseu = field("pseu");
  const double dt = dt_;
  for (int step = 0; step < nsteps; ++step) {
    temp.sp->to_ortho(temp.vhat, old_[2], st_);            // buoyancy: temp.to_ortho(), no lift (lnse.rs:265)
    if (nonlin_) lincomb(old_[2], 1.0, old_[2], 1.0, mean("temp").vhat);   // + mean.temp.to_ortho() (nonlin.rs:266)
    backward(velx, ux_);
    backward(vely, uy_);
    // solve_velx (lnse_eq.rs:179-190; nonlin_eq.rs:193-208)
    zero(rhs_);
    acc_to_ortho(velx, 1.0, rhs_);
    acc_gradient(pres, 1, 0, -dt, rhs_);
    conv_lin(mean("velx"), velx, cv_);
    lincomb(rhs_, 1.0, rhs_, -dt, cv_);
    if (nonlin_) mean_diffusion(mean("velx"), nu_);
    hh_vel_->solve(rhs_, velx.vhat, st_);
    // solve_vely (lnse_eq.rs:193-207)
    zero(rhs_);
    acc_to_ortho(vely, 1.0, rhs_);
    acc_gradient(pres, 0, 1, -dt, rhs_);
    lincomb(rhs_, 1.0, rhs_, dt, old_[2]);
    conv_lin(mean("vely"), vely, cv_);
    lincomb(rhs_, 1.0, rhs_, -dt, cv_);
    if (nonlin_) mean_diffusion(mean("vely"), nu_);
    hh_vel_->solve(rhs_, vely.vhat, st_);
    // projection (lnse.rs:277-281); update_pres (lnse_eq.rs:140-146)
    div(div_);
    solve_pres(div_);
    correct_velocity(1.0);
    lincomb(pres.vhat, 1.0, pres.vhat, -nu_, div_);
    acc_to_ortho(pseu, 1.0 / dt, pres.vhat);
    // solve_temp (lnse_eq.rs:212-221)
    zero(rhs_);
    acc_to_ortho(temp, 1.0, rhs_);
    conv_lin(mean("temp"), temp, cv_);
    lincomb(rhs_, 1.0, rhs_, -dt, cv_);
    if (nonlin_) mean_diffusion(mean("temp"), ka_);
    hh_temp_->solve(rhs_, temp.vhat, st_);
    time_ += dt_;
  }
  dev_sync(st_);
}

// ================================================================================================
// adjoint LNSE step and the gradient of the final energy (lnse_adj_eq.rs, lnse_adj_grad.rs, lnse_fd_grad.rs, functions.rs)

// + U d/dx f* + V d/dy f*  - u* d_j U - v* d_j V - T* d_j Tm   (lnse_adj_eq.rs:16-94; the temperature: no mean-gradient terms)
// nl (Navier2DNonLin, nonlin_adj_eq.rs:16-118): the same terms once more with the forward state of this time level in the mean's place
void Navier2DLnseEngine::conv_adj(F& f, int d0, int d1, bool mean_gradients, Arr2& out, Hist* nl) {
  conv_term(um_, f, 1, 0, 1.0, true);
  conv_term(vm_, f, 0, 1, 1.0, false);
  if (mean_gradients) {
    conv_term(ux_, mean("velx"), d0, d1, -1.0, false);
    conv_term(uy_, mean("vely"), d0, d1, -1.0, false);
    conv_term(tp_, mean("temp"), d0, d1, -1.0, false);
  }
  if (nl) {
    conv_term(unl_, f, 1, 0, 1.0, false);
    conv_term(vnl_, f, 0, 1, 1.0, false);
    if (mean_gradients) {
      conv_term(ux_, nl->velx, d0, d1, -1.0, false);
      conv_term(uy_, nl->vely, d0, d1, -1.0, false);
      conv_term(tp_, nl->temp, d0, d1, -1.0, false);
    }
  }
  conv_finish(out);
}

void Navier2DLnseEngine::update_adjoint(int nsteps) {
  F &velx = field("velx"), &vely = field("vely"), &temp = field("temp"), &pres = field("pres"), &pseu = field("pseu");
  const double dt = dt_;
  if (adj_ && nsteps > 0) {   // the linear solver, confined: the whole adjoint step on the fused schedule (adjoint.h adj_)
    if (adj_mean_gen_ != const_gen_) { push_mean(*adj_, true); adj_mean_gen_ = const_gen_; }
    // the physical arrays of the START of the last step are what grad_adjoint returns (lnse_adj_grad.rs:185-191): the state of the
    // start of step n is the state after n - 1 fused steps
    if (nsteps > 1) {
      dev_sync(st_);
      for (const char* name : {"velx", "vely", "temp", "pres"}) adj_->set_field_spectral_device(name, field(name).vhat);
      adj_->update(nsteps - 1);
      for (const char* name : {"velx", "vely", "temp", "pres", "pseu"}) adj_->get_field_spectral_device(name, field(name).vhat);
      adj_->sync();
    }
    backward(velx, ux_);
    backward(vely, uy_);
    backward(temp, tp_);
    dev_sync(st_);
    for (const char* name : {"velx", "vely", "temp", "pres"}) adj_->set_field_spectral_device(name, field(name).vhat);
    adj_->update(1);
    for (const char* name : {"velx", "vely", "temp", "pres", "pseu"}) adj_->get_field_spectral_device(name, field(name).vhat);
    adj_->sync();
    time_ += nsteps * dt_;
    return;
  }
  for (int step = 0; step < nsteps; ++step) {
    std::unique_ptr<Hist> nlh;
    if (nonlin_) {                                         // nonlin_adj_grad.rs:190-193: the last forward state, removed from the history
      RPDE_REQUIRE(!hist_.empty(), "update_adjoint: the field history is empty (Navier2DNonLin: one update_direct() per adjoint step)");
      nlh = std::move(hist_.back());
      hist_.pop_back();
      backward(nlh->velx, unl_);
      backward(nlh->vely, vnl_);
    }
    Hist* nl = nlh.get();
    vely.sp->to_ortho(vely.vhat, old_[2], st_);            // adjoint buoyancy: vely.to_ortho() (lnse_adj_grad.rs:73)
    backward(velx, ux_);
    backward(vely, uy_);
    backward(temp, tp_);
    // solve_velx_adj (lnse_adj_eq.rs:217-238)
    zero(rhs_);
    acc_to_ortho(velx, 1.0, rhs_);
    acc_gradient(pres, 1, 0, -dt, rhs_);
    conv_adj(velx, 1, 0, true, cv_, nl);
    lincomb(rhs_, 1.0, rhs_, dt, cv_);
    hh_vel_->solve(rhs_, velx.vhat, st_);
    // solve_vely_adj (lnse_adj_eq.rs:241-262)
    zero(rhs_);
    acc_to_ortho(vely, 1.0, rhs_);
    acc_gradient(pres, 0, 1, -dt, rhs_);
    conv_adj(vely, 0, 1, true, cv_, nl);
    lincomb(rhs_, 1.0, rhs_, dt, cv_);
    hh_vel_->solve(rhs_, vely.vhat, st_);
    // projection (lnse_adj_grad.rs:87-91)
    div(div_);
    solve_pres(div_);
    correct_velocity(1.0);
    lincomb(pres.vhat, 1.0, pres.vhat, -nu_, div_);
    acc_to_ortho(pseu, 1.0 / dt, pres.vhat);
    // solve_temp_adj (lnse_adj_eq.rs:269-294)
    zero(rhs_);
    acc_to_ortho(temp, 1.0, rhs_);
    conv_adj(temp, 0, 0, false, cv_, nl);
    lincomb(rhs_, 1.0, rhs_, dt, cv_);
    lincomb(rhs_, 1.0, rhs_, dt, old_[2]);
    hh_temp_->solve(rhs_, temp.vhat, st_);
    time_ += dt_;
    if (nl) dev_sync(st_);                                 // the history entry goes out of scope
  }
  dev_sync(st_);
}

double Navier2DLnseEngine::sumsq(const Arr2& a) {
  launch_sumsq(a.p(), a.ld, a.rows, a.cols * a.elem, red_.p, st_);
  dev_sync(st_);
  double h[2];
  dev_download(h, red_.p, sizeof(h));
  return h[1] > 0 ? std::nan("") : h[0];
}

double Navier2DLnseEngine::energy(double beta1, double beta2, const double* tu, const double* tv, const double* tt) {
  RPDE_REQUIRE((tu && tv && tt) || (!tu && !tv && !tt), "energy: a target is three arrays (velx, vely, temp) or none");
  backward(field("velx"), ux_);
  backward(field("vely"), uy_);
  backward(field("temp"), tp_);
  double s[3];
  Arr2* phys[3] = {&ux_, &uy_, &tp_};
  const double* tg[3] = {tu, tv, tt};
  for (int k = 0; k < 3; ++k) {
    if (tg[k]) {
      dev_sync(st_);
      dev_upload2d(ph_.p(), ph_.ld, tg[k], nx_, ny_);
      lincomb(ta_, 1.0, *phys[k], -1.0, ph_);
      s[k] = sumsq(ta_);
    } else {
      s[k] = sumsq(*phys[k]);
    }
  }
  return 0.5 * (beta1 * s[0] + beta1 * s[1] + beta2 * s[2]);
}

bool Navier2DLnseEngine::exit_grad(double max_time, long timestep) {
  if (time_ + dt_ * 1e-4 >= max_time) return true;
  if (timestep >= 10000000L) return true;
  if (std::isnan(div_norm())) { std::printf("Divergence is nan\n"); return true; }
  return false;
}

long Navier2DLnseEngine::integrate(double max_time, double save_intervall) {
  long timestep = 0;
  char fname[64];
  for (;;) {
    update(1);
    ++timestep;
    if (save_intervall > 0.0) {                // src/lib.rs:196-204: Integrate::callback (lnse.rs:298-302 / nonlin.rs) on the save interval
      const double r = std::fmod(time_, save_intervall);
      if (r < dt_ / 2.0 || r > save_intervall - dt_ / 2.0) {
        std::snprintf(fname, sizeof fname, "data/flow%08.2f.h5", time_);
        callback_from_filename(fname, "data/info.txt", false, -1.0);
      }
    }
    if (time_ + dt_ * 1e-4 >= max_time) break;
    if (timestep >= 10000000L) break;
    if (exit()) break;
  }
  return timestep;
}

void Navier2DLnseEngine::write_gradient(const char* filename, const double* gu, const double* gv, const double* gt) {
  // Field2::write_unwrap(filename, "ux" | "uy" | "temp") of the gradient fields (lnse_adj_grad.rs:192-200): v, vhat = forward(v)
  h5::Tree t;
  Vec x((size_t)nx_), y((size_t)ny_);
  grid(0, x.data(), x.size());
  grid(1, y.data(), y.size());
  const char* const grp[3] = {"ux", "uy", "temp"};
  const char* const fld[3] = {"velx", "vely", "temp"};
  const double* g[3] = {gu, gv, gt};
  for (int k = 0; k < 3; ++k) {
    const std::string gname = grp[k];
    F& f = field(fld[k]);
    t[gname + "/x"] = h5::Dataset{{(uint64_t)nx_}, x};
    t[gname + "/dx"] = h5::Dataset{{(uint64_t)nx_}, x};
    t[gname + "/y"] = h5::Dataset{{(uint64_t)ny_}, y};
    t[gname + "/dy"] = h5::Dataset{{(uint64_t)ny_}, y};
    t[gname + "/v"] = h5::Dataset{{(uint64_t)nx_, (uint64_t)ny_}, Vec(g[k], g[k] + (size_t)nx_ * ny_)};
    Arr2 vh(f.vhat.rows, f.vhat.cols, ex_);
    dev_sync(st_);
    dev_upload2d(ph_.p(), ph_.ld, g[k], nx_, ny_);
    f.sp->forward(ph_, vh, st_);
    dev_sync(st_);
    const int r = vh.rows, c = vh.cols, e = vh.elem;
    Vec h((size_t)r * c * e);
    dev_download2d(h.data(), vh.p(), vh.ld, r, (long)c * e);
    if (e == 1) {
      t[gname + "/vhat"] = h5::Dataset{{(uint64_t)r, (uint64_t)c}, std::move(h)};
    } else {
      h5::Dataset re{{(uint64_t)r, (uint64_t)c}, Vec((size_t)r * c)}, im = re;
      for (size_t q = 0; q < (size_t)r * c; ++q) { re.data[q] = h[2 * q]; im.data[q] = h[2 * q + 1]; }
      t[gname + "/vhat_re"] = std::move(re);
      t[gname + "/vhat_im"] = std::move(im);
    }
  }
  h5::update_file(filename, t);
}

void Navier2DLnseEngine::diagnostics(double out[7]) {
  for (int k = 0; k < 7; ++k) out[k] = std::nan("");
  out[0] = div_norm();
  Space2Ops& so = *sp_ortho_;
  const Vec x0 = base_coords(so.base(0)), y0 = base_coords(so.base(1));
  Vec wx = base_dx(so.base(0), x0), wy = base_dx(so.base(1), y0);
  const double lx = std::fabs(x0.back() - x0.front()), ly = std::fabs(y0.back() - y0.front());
  for (double& w : wx) w /= lx;
  for (double& w : wy) w /= ly;
  DBuf dwx, dwy, part((size_t)4 * nx_), out4(4);
  dwx.upload(wx); dwy.upload(wy);
  double h[4];
  // <f^2> (field/average.rs:26-59 of the squared physical field): the volume sum of launch_diag_reduce with T = uy = f
  const char* const names[3] = {"velx", "vely", "temp"};
  for (int k = 0; k < 3; ++k) {
    backward(field(names[k]), ta_);
    launch_diag_reduce(ta_.p(), ta_.p(), ta_.p(), ta_.p(), ta_.ld, nx_, ny_, dwx.p, dwy.p, 0.0, 0.0, 1.0, 0.0, part.p, out4.p, st_);
    dev_sync(st_);
    dev_download(h, out4.p, sizeof(h));
    out[4 + k] = h[2];
  }
  if (!nonlin_) return;
  // eval_nu / eval_nuvol / eval_re (functions.rs:60-143) on state.to_ortho() + mean.vhat (nonlin_io.rs:145-198)
  Arr2 tot(so.ortho_rows(), so.ortho_cols(), ex_), g(so.ortho_rows(), so.ortho_cols(), ex_);
  Arr2 tphys(nx_, ny_, 1), dtdz(nx_, ny_, 1), ux(nx_, ny_, 1), uy(nx_, ny_, 1);
  auto total = [&](const char* name, Arr2& phys) {
    F& f = field(name);
    f.sp->to_ortho(f.vhat, tot, st_);
    lincomb(tot, 1.0, tot, 1.0, mean(name).vhat);
    so.backward(tot, phys, st_);
  };
  total("velx", ux);
  total("vely", uy);
  total("temp", tphys);                        // tot = total temperature
  so.gradient(tot, 0, 1, 1.0, 1.0, g, st_);    // unscaled d/dy (scale = None)
  so.backward(g, dtdz, st_);
  launch_diag_reduce(tphys.p(), dtdz.p(), ux.p(), uy.p(), tphys.ld, nx_, ny_, dwx.p, dwy.p, -2.0 / sy_,
                     (1.0 / (sy_ * -1.0)) * 2.0 * sy_, (1.0 / ka_) * 2.0 * sy_, 2.0 * sy_ / nu_, part.p, out4.p, st_);
  dev_sync(st_);
  dev_download(h, out4.p, sizeof(h));
  out[1] = (h[1] + h[0]) / 2.0;
  out[2] = h[2];
  out[3] = h[3];
}

void Navier2DLnseEngine::callback_from_filename(const std::string& flow_name, const std::string& info_name, bool suppress_io,
                                                double write_flow_intervall) {
  (void)::mkdir("data", 0777);                 // std::fs::create_dir_all("data")
  const double out_intervall = write_flow_intervall >= 0.0 ? write_flow_intervall : 1.0;   // OUTPUT_INTERVALL (lnse.rs:21)
  if (std::fmod(time_ + dt_ / 2.0, out_intervall) < dt_) {
    try {
      write(flow_name);
      // write() of the reference refreshes the physical arrays the state holds (lnse_io.rs:44-47) -- the arrays grad_adjoint returns
      backward(field("velx"), ux_);
      backward(field("vely"), uy_);
      backward(field("temp"), tp_);
    } catch (const std::exception& ex) {
      std::fprintf(stderr, "Error while writing file \"%s\". Error: %s\n", flow_name.c_str(), ex.what());
    }
  }
  if (suppress_io) return;
  double d[7];
  diagnostics(d);
  char tbuf[64];
  std::snprintf(tbuf, sizeof tbuf, "%5.3f", time_);
  FILE* fp = std::fopen(info_name.c_str(), "a");
  if (nonlin_) {
    std::printf("time = %s |div| = %s Nu = %s Nuv = %s Re = %s u2 = %s v2 = %s t2 = %s\n", tbuf, rust_exp(d[0], 2).c_str(),
                rust_exp(d[1], 3).c_str(), rust_exp(d[2], 3).c_str(), rust_exp(d[3], 3).c_str(), rust_exp(d[4], 3).c_str(),
                rust_exp(d[5], 3).c_str(), rust_exp(d[6], 3).c_str());
    if (fp) std::fprintf(fp, "%s %s %s %s %s %s %s\n", rust_display(time_).c_str(), rust_display(d[1]).c_str(), rust_display(d[2]).c_str(),
                         rust_display(d[3]).c_str(), rust_display(d[4]).c_str(), rust_display(d[5]).c_str(), rust_display(d[6]).c_str());
  } else {
    std::printf("time = %s      |div| = %s     u2 = %s     v2 = %s    t2 = %s\n", tbuf, rust_exp(d[0], 2).c_str(), rust_exp(d[4], 3).c_str(),
                rust_exp(d[5], 3).c_str(), rust_exp(d[6], 3).c_str());
    if (fp) std::fprintf(fp, "%s %s %s %s\n", rust_display(time_).c_str(), rust_display(d[4]).c_str(), rust_display(d[5]).c_str(),
                         rust_display(d[6]).c_str());
  }
  std::fflush(stdout);
  if (fp) std::fclose(fp);
  else std::fprintf(stderr, "Couldn't write to file: %s\n", info_name.c_str());
}

double Navier2DLnseEngine::grad_adjoint(double max_time, double save_intervall, double beta1, double beta2, const double* tu,
                                        const double* tv, const double* tt, double* gu, double* gv, double* gt, const char* filename,
                                        long* timesteps) {
  RPDE_REQUIRE(gu && gv && gt, "grad_adjoint: null output");
  RPDE_REQUIRE((tu && tv && tt) || (!tu && !tv && !tt), "grad_adjoint: a target is three arrays (velx, vely, temp) or none");
  long timestep = 0;
  char fname[64];
  for (;;) {                                   // forward loop (:119-136)
    update_direct(1);
    ++timestep;
    if (save_intervall > 0.0) {                // :122-130 (Navier2DLnse suppresses the info line here, Navier2DNonLin does not)
      const double r = std::fmod(time_, save_intervall);
      if (r < dt_ / 2.0 || r > save_intervall - dt_ / 2.0) {
        std::snprintf(fname, sizeof fname, "data/flow%08.2f.h5", time_);
        callback_from_filename(fname, "data/info.txt", !nonlin_, -1.0);
      }
    }
    if (exit_grad(max_time, timestep)) break;
  }
  const double fun_val = energy(beta1, beta2, tu, tv, tt);   // :139-155
  // initial condition of the adjoint fields (:160-169): (state - from_ortho(target.vhat)) * beta
  F* fl[3] = {&field("velx"), &field("vely"), &field("temp")};
  const double* tg[3] = {tu, tv, tt};
  const double be[3] = {beta1, beta1, beta2};
  for (int k = 0; k < 3; ++k) {
    F& f = *fl[k];
    if (tg[k]) {
      Arr2 tmp(f.vhat.rows, f.vhat.cols, ex_);
      dev_sync(st_);
      dev_upload2d(ph_.p(), ph_.ld, tg[k], nx_, ny_);
      sp_ortho_->forward(ph_, t1_, st_);       // MeanFields live in the orthonormal space (meanfield.rs:11-18)
      f.sp->from_ortho(t1_, tmp, st_);
      lincomb(f.vhat, 1.0, f.vhat, -1.0, tmp);
      dev_sync(st_);
    }
    lincomb(f.vhat, be[k], f.vhat, 0.0, f.vhat);
  }
  reset_time();
  for (;;) {                                   // adjoint loop (:172-187); the step counter keeps running
    update_adjoint(1);
    ++timestep;
    if (save_intervall > 0.0 && std::fmod(time_ + dt_ / 2.0, save_intervall) < dt_) {   // :176-181
      std::snprintf(fname, sizeof fname, "data/adjoint%08.2f.h5", time_);
      callback_from_filename(fname, "data/info_adjoint.txt", false, -1.0);
    }
    if (exit_grad(max_time, timestep)) break;
  }
  // :189-196: the gradient is fac * self.velx.v, the physical arrays of the backward() at the START of the last adjoint step
  const double fac = -1.0;                     // MAXIMIZE = false (:16)
  Arr2* phys[3] = {&ux_, &uy_, &tp_};
  double* out[3] = {gu, gv, gt};
  dev_sync(st_);
  for (int k = 0; k < 3; ++k) {
    dev_download2d(out[k], phys[k]->p(), phys[k]->ld, nx_, ny_);
    for (size_t q = 0; q < (size_t)nx_ * ny_; ++q) out[k][q] *= fac;
  }
  if (filename && *filename) write_gradient(filename, gu, gv, gt);
  if (timesteps) *timesteps = timestep;
  return fun_val;
}

void Navier2DLnseEngine::grad_fd(double max_time, double beta1, double beta2, const int* points, long npoints, double* gu, double* gv,
                                 double* gt, const char* filename, double save_intervall) {
  RPDE_REQUIRE(gu && gv && gt, "grad_fd: null output");
  const double eps = 1e-5;                     // lnse_fd_grad.rs:38
  const char* const fld[3] = {"velx", "vely", "temp"};
  const size_t np = (size_t)nx_ * ny_;
  // base state (:41-46): the reference saves v and vhat; the engine keeps spectral fields only, v = backward(vhat)
  Vec base_v[3];
  Arr2 base_h[3];
  for (int k = 0; k < 3; ++k) {
    F& f = field(fld[k]);
    base_v[k].resize(np);
    get_field_physical(fld[k], base_v[k].data(), np);
    base_h[k].alloc(f.vhat.rows, f.vhat.cols, ex_);
    lincomb(base_h[k], 1.0, f.vhat, 0.0, f.vhat);
  }
  auto reset = [&]() {
    reset_time();
    for (int k = 0; k < 3; ++k) { F& f = field(fld[k]); lincomb(f.vhat, 1.0, base_h[k], 0.0, base_h[k]); }
    zero(field("pres").vhat);
    zero(field("pseu").vhat);
  };
  reset();
  integrate(max_time, save_intervall);         // the base run alone writes its snapshot series (lnse_fd_grad.rs:54)
  const double e_base = energy(beta1, beta2);
  double* out[3] = {gu, gv, gt};
  for (int k = 0; k < 3; ++k) std::fill(out[k], out[k] + np, 0.0);
  const long total = points ? npoints : (long)(3 * np);
  for (long q = 0; q < total; ++q) {
    int k, i, j;
    if (points) { k = points[3 * q]; i = points[3 * q + 1]; j = points[3 * q + 2]; }
    else { k = (int)(q / (long)np); i = (int)((q % (long)np) / ny_); j = (int)(q % ny_); }
    RPDE_REQUIRE(k >= 0 && k < 3 && i >= 0 && i < nx_ && j >= 0 && j < ny_, "grad_fd: point outside the grid");
    reset();
    F& f = field(fld[k]);
    dev_sync(st_);
    dev_upload2d(ph_.p(), ph_.ld, base_v[k].data(), nx_, ny_);
    launch_set_element(ph_.p(), (long)i * ph_.ld + j, base_v[k][(size_t)i * ny_ + j] + eps, st_);
    f.sp->forward(ph_, f.vhat, st_);
    integrate(max_time);
    out[k][(size_t)i * ny_ + j] = 1.0 / eps * (energy(beta1, beta2) - e_base);
  }
  dev_sync(st_);
  if (filename && *filename) write_gradient(filename, gu, gv, gt);
}

}  // namespace rpde
