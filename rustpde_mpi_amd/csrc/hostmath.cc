#include "hostmath.h"

#include <dlfcn.h>
#include <glob.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "platform.h"

namespace rpde {

static const long double kPiL = 3.141592653589793238462643383279502884L;

Base make_base(BaseKind kind, int n) {
  Base b{kind, n, n};
  if (kind == kChebDirichlet || kind == kChebNeumann || kind == kChebDirichletNeumann) b.m = n - 2;
  if (kind == kFourierR2c) {
    RPDE_REQUIRE(n >= 2, "fourier_r2c needs at least two points");   // odd lengths: realfft's r2c / c2r take them, so do the Bluestein lines
    b.m = n / 2 + 1;
  }
  RPDE_REQUIRE(n >= 5 || kind == kFourierR2c, "Chebyshev bases need n >= 5");
  return b;
}

Vec base_coords(const Base& b) {
  Vec x(b.n);
  for (int j = 0; j < b.n; ++j)
    x[j] = b.is_cheb() ? -std::cos(M_PI * (double)j / (double)(b.n - 1))
                       : 2.0 * M_PI * (double)j / (double)b.n;
  return x;
}

Vec base_dx(const Base& b, const Vec& x) {
  const int n = b.n;
  Vec dx(n);
  if (!b.is_cheb()) {
    std::fill(dx.begin(), dx.end(), x[2] - x[1]);
    return dx;
  }
  for (int i = 0; i < n; ++i) {
    const double l = (i == 0) ? x[0] : (x[i] + x[i - 1]) / 2.0;
    const double r = (i == n - 1) ? x[n - 1] : (x[i + 1] + x[i]) / 2.0;
    dx[i] = r - l;
  }
  return dx;
}

Vec stencil_low(const Base& b) {
  RPDE_REQUIRE(b.is_composite(), "stencil of a non-composite base");
  Vec low(b.m);
  for (int k = 0; k < b.m; ++k) {
    if (b.kind == kChebDirichlet) low[k] = -1.0;
    else if (b.kind == kChebNeumann) { const double r = (double)k / ((double)k + 2.0); low[k] = -(r * r); }
    else {   // phi_k = T_k + a_k T_{k+1} + b_k T_{k+2}, phi_k(-1) = 0, phi_k'(+1) = 0:  b_k = a_k - 1 (see stencil_low1)
      const double kk = (double)k;
      low[k] = -(kk * kk + (kk + 1.0) * (kk + 1.0)) / ((kk + 1.0) * (kk + 1.0) + (kk + 2.0) * (kk + 2.0));
    }
  }
  return low;
}

Vec stencil_low1(const Base& b) {
  RPDE_REQUIRE(b.is_composite(), "stencil of a non-composite base");
  Vec low1(b.m, 0.0);
  if (b.kind == kChebDirichletNeumann)
    // Dirichlet at x = -1 (T_k(-1) = (-1)^k), Neumann at x = +1 (T_k'(1) = k^2):
    // 1 - a_k + b_k = 0 and k^2 + a_k (k+1)^2 + b_k (k+2)^2 = 0  =>  a_k = 4 (k+1) / ((k+1)^2 + (k+2)^2).
    // The wall assignment is the one the lift bc_hc needs (boundary_conditions.rs:96-134: the bottom temperature
    // at y[0] = -1, T = T' = 0 at y[n-1] = +1).
    for (int k = 0; k < b.m; ++k) {
      const double kk = (double)k;
      low1[k] = 4.0 * (kk + 1.0) / ((kk + 1.0) * (kk + 1.0) + (kk + 2.0) * (kk + 2.0));
    }
  return low1;
}

Vec cheb_fwd_post(int n) {
  Vec f(n);
  for (int k = 0; k < n; ++k) f[k] = ((k & 1) ? -1.0 : 1.0) / (double)(n - 1);
  f[0] *= 0.5; f[n - 1] *= 0.5;
  return f;
}
Vec cheb_bwd_pre(int n) {
  Vec f(n);
  for (int k = 0; k < n; ++k) f[k] = ((k & 1) ? -1.0 : 1.0) * 0.5;
  f[0] *= 2.0; f[n - 1] *= 2.0;
  return f;
}

Vec fft_twiddles(int nfft) {
  Vec t(2 * (size_t)nfft);
  for (int k = 0; k < nfft; ++k) {
    const long double a = 2.0L * kPiL * (long double)k / (long double)nfft;
    t[2 * k] = (double)cosl(a);
    t[2 * k + 1] = (double)(-sinl(a));
  }
  return t;
}
Vec dct_split_twiddles(int N) {
  Vec t(2 * (size_t)(N + 1));
  for (int k = 0; k <= N; ++k) {
    const long double a = kPiL * (long double)k / (long double)N;
    t[2 * k] = (double)cosl(a);
    t[2 * k + 1] = (double)sinl(a);
  }
  return t;
}
Vec rfft_split_twiddles(int nx) {
  const int M = nx / 2;
  Vec t(2 * (size_t)(M + 1));
  for (int k = 0; k <= M; ++k) {
    const long double a = 2.0L * kPiL * (long double)k / (long double)nx;
    t[2 * k] = (double)cosl(a);
    t[2 * k + 1] = (double)sinl(a);
  }
  return t;
}
Vec dct_direct_costab(int N) {
  Vec t(2 * (size_t)N);
  for (int m = 0; m < 2 * N; ++m) t[m] = (double)cosl(kPiL * (long double)m / (long double)N);
  return t;
}

// ---------------------------------------------------------------------------------------------
// Bluestein (chirp-z) tables: a transform of ANY length as a circular convolution of power-of-two length M.
// The reference's transforms accept every n (rustdct / rustfft under funspace: mixed radix, Rader, Bluestein); its own
// benches run Chebyshev n = 128, 264, 512, 1024 (benches/benchmark_navier.rs:6-7, benchmark_transform.rs:6), i.e.
// N = n - 1 = 127, 263 (primes), 511 = 7 * 73, 1023 = 3 * 11 * 31 -- lengths no small-radix plan covers.
//   2 j k = j^2 + k^2 - (k - j)^2  =>  sum_j x_j w^(2 j k) = w^(k^2) sum_j (x_j w^(j^2)) w^(-(k - j)^2)
int bluestein_len(int lags) {
  int m = 2;
  while (m < lags) m *= 2;
  return m;
}
namespace {
using CplxL = std::pair<long double, long double>;
// in-place radix-2 FFT (forward sign) in long double: setup only, M <= 16384
void fft_long(std::vector<CplxL>& a) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t k = 0; k < len / 2; ++k) {
      const long double ang = -2.0L * kPiL * (long double)k / (long double)len;
      const long double wr = cosl(ang), wi = sinl(ang);
      for (size_t i = k; i < n; i += len) {
        const CplxL u = a[i], v = a[i + len / 2];
        const long double tr = v.first * wr - v.second * wi, ti = v.first * wi + v.second * wr;
        a[i] = {u.first + tr, u.second + ti};
        a[i + len / 2] = {u.first - tr, u.second - ti};
      }
    }
  }
}
// (cos, sin)(pi * (m^2 mod period) / den): the argument is reduced in integers, so the chirp is exact to the last bit of
// cosl / sinl however large m^2 gets
CplxL chirp(long m, long period, long den) {
  const long r = (long)(((long long)m * (long long)m) % (long long)period);
  const long double a = kPiL * (long double)r / (long double)den;
  return {cosl(a), sinl(a)};
}
// FFT_M of the wrapped filter b~[m mod M] = (cos, sgn * sin)(pi m^2 / den), m in [lo, hi], scaled by 1 / M
void filter_spectrum(Vec& out, size_t off, int M, long lo, long hi, long period, long den, int sgn) {
  RPDE_REQUIRE(hi - lo + 1 <= M, "bluestein: the convolution length does not hold the lags");
  std::vector<CplxL> b((size_t)M, CplxL{0.0L, 0.0L});
  for (long m = lo; m <= hi; ++m) {
    const CplxL c = chirp(m, period, den);
    b[(size_t)(((m % M) + M) % M)] = {c.first, sgn * c.second};
  }
  fft_long(b);
  for (int i = 0; i < M; ++i) {
    out[off + 2 * (size_t)i] = (double)(b[(size_t)i].first / (long double)M);
    out[off + 2 * (size_t)i + 1] = (double)(b[(size_t)i].second / (long double)M);
  }
}
}  // namespace

// DCT-I of N + 1 points: w = exp(i pi / (2 N)), chirp (cos, sin)(pi j^2 / (2 N)) for j <= N [2 (N + 1) doubles], then the
// spectrum of the filter w^(-m^2), |m| <= N [2 M doubles]
Vec bluestein_dct_tables(int N, int M) {
  RPDE_REQUIRE(N >= 1 && M >= 2 * N + 1, "bluestein_dct_tables: M >= 2 N + 1");
  Vec t(2 * (size_t)(N + 1) + 2 * (size_t)M);
  for (int j = 0; j <= N; ++j) {
    const CplxL c = chirp(j, 4L * N, 2L * N);
    t[2 * (size_t)j] = (double)c.first;
    t[2 * (size_t)j + 1] = (double)c.second;
  }
  filter_spectrum(t, 2 * (size_t)(N + 1), M, -N, N, 4L * N, 2L * N, -1);
  return t;
}
// real FFT of nx points (K = nx / 2): chirp (cos, sin)(pi j^2 / nx), j < nx [2 nx doubles]; the forward filter
// exp(+i pi m^2 / nx), -(nx - 1) <= m <= K [2 M]; the backward filter exp(-i pi m^2 / nx), -K <= m <= nx - 1 [2 M]
Vec bluestein_rfft_tables(int nx, int M) {
  const int K = nx / 2;
  RPDE_REQUIRE(nx >= 2 && M >= nx + K, "bluestein_rfft_tables: M >= nx + nx / 2");
  Vec t(2 * (size_t)nx + 4 * (size_t)M);
  for (int j = 0; j < nx; ++j) {
    const CplxL c = chirp(j, 2L * nx, nx);
    t[2 * (size_t)j] = (double)c.first;
    t[2 * (size_t)j + 1] = (double)c.second;
  }
  filter_spectrum(t, 2 * (size_t)nx, M, -(long)(nx - 1), K, 2L * nx, nx, +1);
  filter_spectrum(t, 2 * (size_t)nx + 2 * (size_t)M, M, -(long)K, nx - 1, 2L * nx, nx, -1);
  return t;
}

// ---------------------------------------------------------------------------------------------
// three-term stencil: seven-diagonal matrices and their PdmaPlus2 factorisation
static double sten_at(const Vec& low1, const Vec& low2, int m, int t, int j) {   // S[j + t, j]
  if (j < 0 || j >= m) return 0.0;
  return t == 0 ? 1.0 : t == 1 ? low1[j] : t == 2 ? low2[j] : 0.0;
}
Bands7 hholtz7_mat_a(const Base& b) {
  const int m = b.m;
  const Mv3Tables p = pinv_tables(b);
  const Vec l1 = stencil_low1(b), l2 = stencil_low(b);
  Bands7 a;
  for (auto& v : a.d) v.assign(m, 0.0);
  for (int r = 0; r < m; ++r)
    for (int o = -2; o <= 4; ++o) {
      double s = 0.0;   // sum_q pinv[r, r+q] S[r+q, r+o], q = 0, 2, 4; S[i, j] != 0 for 0 <= i - j <= 2
      if (o <= 0) s += p.t0[r] * sten_at(l1, l2, m, 0 - o, r + o);
      if (o <= 2 && o >= 0) s += p.t1[r] * sten_at(l1, l2, m, 2 - o, r + o);
      if (o >= 2) s += p.t2[r] * sten_at(l1, l2, m, 4 - o, r + o);
      a.d[o + 2][r] = s;
    }
  return a;
}
Bands7 hholtz7_mat_b(const Base& b) {
  const int m = b.m;
  const Vec l1 = stencil_low1(b), l2 = stencil_low(b);
  Bands7 x;
  for (auto& v : x.d) v.assign(m, 0.0);
  for (int r = 0; r < m; ++r)
    for (int o = 0; o <= 2; ++o) x.d[o + 2][r] = sten_at(l1, l2, m, 2 - o, r + o);   // peye[r, r+2] = 1
  return x;
}
Bands7 bands7_axpy(const Bands7& a, double c, const Bands7& b) {
  Bands7 r;
  for (int o = 0; o < 7; ++o) {
    r.d[o].resize(a.d[o].size());
    for (size_t i = 0; i < a.d[o].size(); ++i) r.d[o][i] = a.d[o][i] + c * b.d[o][i];
  }
  return r;
}
Bands7 from_ortho7(const Base& b) {
  const int m = b.m;
  const Vec l1 = stencil_low1(b), l2 = stencil_low(b);
  Bands7 x;
  for (auto& v : x.d) v.assign(m, 0.0);
  for (int r = 0; r < m; ++r)
    for (int o = -2; o <= 2; ++o) {
      if (r + o < 0 || r + o >= m) continue;
      double s = 0.0;   // sum_i S[i, r] S[i, r + o]
      for (int t = 0; t <= 2; ++t) s += sten_at(l1, l2, m, t, r) * sten_at(l1, l2, m, t - o, r + o);
      x.d[o + 2][r] = s;
    }
  return x;
}
PdmaTables pdma_factor(const Bands7& mt) {
  const int n = (int)mt.d[2].size();
  PdmaTables t;
  t.n = n;
  for (Vec* v : {&t.l2, &t.ka, &t.imu, &t.al, &t.be, &t.ga, &t.de}) v->assign((size_t)n + 4, 0.0);
  auto at = [&](int o, int r) { return (r >= 0 && r < n && r + o >= 0 && r + o < n) ? mt.d[o + 2][r] : 0.0; };
  Vec mu((size_t)n + 4, 1.0);
  auto prev = [](const Vec& v, int i) { return i >= 0 ? v[i] : 0.0; };
  for (int i = 0; i < n; ++i) {   // pdma_plus2.rs:66-103 with vanishing out-of-range terms (its rows 0, 1, n-4 .. n-1)
    const double w2 = at(-2, i);                       // the reference's l2[i-2] = a[i, i-2]
    t.l2[i] = w2;
    t.ka[i] = at(-1, i) - prev(t.al, i - 2) * w2;
    mu[i] = at(0, i) - prev(t.be, i - 2) * w2 - prev(t.al, i - 1) * t.ka[i];
    t.al[i] = (at(1, i) - prev(t.ga, i - 2) * w2 - prev(t.be, i - 1) * t.ka[i]) / mu[i];
    t.be[i] = (at(2, i) - prev(t.de, i - 2) * w2 - prev(t.ga, i - 1) * t.ka[i]) / mu[i];
    t.ga[i] = (at(3, i) - prev(t.de, i - 1) * t.ka[i]) / mu[i];
    t.de[i] = at(4, i) / mu[i];
    t.imu[i] = 1.0 / mu[i];
  }
  return t;
}

PdmaBlockTables pdma_block_tables(const PdmaTables& t, int BR) {
  const int n = t.n, NB = (n + BR - 1) / BR;
  PdmaBlockTables o;
  o.NB = NB;
  for (Vec* v : {&o.phi1, &o.phi2, &o.psi1, &o.psi2, &o.psi3, &o.psi4}) v->assign((size_t)n + 4, 0.0);
  o.fm.assign((size_t)4 * NB, 0.0);
  o.bm.assign((size_t)16 * NB, 0.0);
  for (int b = 0; b < NB; ++b) {
    const int j0 = b * BR, j1 = std::min(n, j0 + BR);
    // forward: z_j = (- l2_j z_{j-2} - ka_j z_{j-1}) imu_j with (z_{j0-1}, z_{j0-2}) = (1, 0) and (0, 1)
    for (int s = 0; s < 2; ++s) {
      Vec& phi = s == 0 ? o.phi1 : o.phi2;
      double z1 = s == 0 ? 1.0 : 0.0, z2 = s == 0 ? 0.0 : 1.0;
      for (int j = j0; j < j1; ++j) {
        const double z = (-t.l2[j] * z2 - t.ka[j] * z1) * t.imu[j];
        phi[j] = z;
        z2 = z1; z1 = z;
      }
      o.fm[4 * b + 0 + s] = z1;     // z_{j1-1} (a one-row block: z_{j1-2} = z_{j0-1} = the inflow itself)
      o.fm[4 * b + 2 + s] = z2;
    }
    // backward: x_i = - al_i x_{i+1} - be_i x_{i+2} - ga_i x_{i+3} - de_i x_{i+4} with x_{j1 + k} = delta_{k s}
    for (int s = 0; s < 4; ++s) {
      Vec& psi = s == 0 ? o.psi1 : s == 1 ? o.psi2 : s == 2 ? o.psi3 : o.psi4;
      double x[4] = {s == 0 ? 1.0 : 0.0, s == 1 ? 1.0 : 0.0, s == 2 ? 1.0 : 0.0, s == 3 ? 1.0 : 0.0};   // x_{i+1} .. x_{i+4}
      for (int i = j1 - 1; i >= j0; --i) {
        const double v = -t.al[i] * x[0] - t.be[i] * x[1] - t.ga[i] * x[2] - t.de[i] * x[3];
        psi[i] = v;
        x[3] = x[2]; x[2] = x[1]; x[1] = x[0]; x[0] = v;
      }
      for (int k = 0; k < 4; ++k) o.bm[16 * b + 4 * k + s] = x[k];   // x_{j0 + k} (short blocks: the inflows shine through)
    }
  }
  return o;
}

FromOrthoTables from_ortho_tables(const Base& b) {
  RPDE_REQUIRE(b.is_two_term(), "from_ortho tables of the stride-2 form need a two-term stencil");
  const int m = b.m;
  const Vec low = stencil_low(b);
  Vec main(m), off(std::max(0, m - 2));
  for (int k = 0; k < m; ++k) main[k] = 1.0 + low[k] * low[k];
  for (int k = 0; k + 2 < m; ++k) off[k] = 1.0 * low[k];
  Vec w(m, 0.0), den(m, 0.0);
  for (int i = 0; i < m; ++i) {
    den[i] = (i >= 2) ? main[i] - off[i - 2] * w[i - 2] : main[i];
    if (i < m - 2) w[i] = off[i] / den[i];
  }
  FromOrthoTables t;
  t.t0.assign(m, 1.0);
  t.t1 = low;
  t.t2.assign(m, 0.0);
  t.p_up.resize(m); t.q_up.assign(m, 0.0); t.q_dn.assign(m, 0.0);
  for (int i = 0; i < m; ++i) {
    t.p_up[i] = 1.0 / den[i];
    if (i >= 2) t.q_up[i] = -off[i - 2] / den[i];
    if (i < m - 2) t.q_dn[i] = -w[i];
  }
  return t;
}

Mv3Tables pinv_tables(const Base& b) {
  RPDE_REQUIRE(b.is_cheb(), "pinv of a Fourier base");
  const int n = b.n, m = n - 2;
  Mv3Tables t;
  t.t0.resize(m); t.t1.assign(m, 0.0); t.t2.assign(m, 0.0);
  for (int r = 0; r < m; ++r) {
    const double i = (double)(r + 2);
    t.t0[r] = (r == 0) ? 0.25 : 1.0 / (4.0 * i * (i - 1.0));
    if (r + 2 <= n - 3) t.t1[r] = -1.0 / (2.0 * (i * i - 1.0));
    if (r + 2 <= n - 5) t.t2[r] = 1.0 / (4.0 * i * (i + 1.0));
  }
  return t;
}

Bands hholtz_mat_a(const Base& b) {
  RPDE_REQUIRE(b.is_two_term(), "four-diagonal Helmholtz matrices need a two-term stencil (see hholtz7_mat_a)");
  const int m = b.m;
  const Mv3Tables p = pinv_tables(b);
  const Vec low = stencil_low(b);
  auto dia_at = [&](int k) { return k < m ? 1.0 : 0.0; };
  auto low_at = [&](int k) { return k < m ? low[k] : 0.0; };
  Bands a{Vec(m, 0.0), Vec(m), Vec(m), Vec(m)};
  for (int r = 0; r < m; ++r) {
    if (r >= 2) a.low[r] = p.t0[r] * low[r - 2];
    a.dia[r] = p.t0[r] * 1.0 + p.t1[r] * low[r];
    a.up1[r] = p.t1[r] * dia_at(r + 2) + p.t2[r] * low_at(r + 2);
    a.up2[r] = p.t2[r] * dia_at(r + 4);
  }
  return a;
}

Bands hholtz_mat_b(const Base& b) {
  RPDE_REQUIRE(b.is_two_term(), "four-diagonal Helmholtz matrices need a two-term stencil (see hholtz7_mat_b)");
  const int m = b.m;
  const Vec low = stencil_low(b);
  Bands x{Vec(m, 0.0), low, Vec(m, 0.0), Vec(m, 0.0)};
  for (int r = 0; r + 2 < m; ++r) x.up1[r] = 1.0;
  return x;
}

Bands bands_axpy(const Bands& a, double c, const Bands& b) {
  const size_t m = a.dia.size();
  Bands r{Vec(m), Vec(m), Vec(m), Vec(m)};
  for (size_t i = 0; i < m; ++i) {
    r.low[i] = a.low[i] + c * b.low[i];
    r.dia[i] = a.dia[i] + c * b.dia[i];
    r.up1[i] = a.up1[i] + c * b.up1[i];
    r.up2[i] = a.up2[i] + c * b.up2[i];
  }
  return r;
}

void fdma_sweep(Bands& mtx) {
  // reference indexing: low_ref[i-2] = low[i]; see src/solver/fdma.rs:73-82
  const int n = (int)mtx.dia.size();
  for (int i = 2; i < n; ++i) {
    mtx.low[i] = mtx.low[i] / mtx.dia[i - 2];
    mtx.dia[i] = mtx.dia[i] - mtx.low[i] * mtx.up1[i - 2];
    if (i < n - 2) mtx.up1[i] = mtx.up1[i] - mtx.low[i] * mtx.up2[i - 2];
  }
}

FdmaTables fdma_tables(const Bands& s) {
  const int n = (int)s.dia.size();
  FdmaTables t{Vec(n, 0.0), Vec(n), Vec(n, 0.0), Vec(n, 0.0)};
  for (int i = 0; i < n; ++i) {
    if (i >= 2) t.q1[i] = -s.low[i];
    t.p2[i] = 1.0 / s.dia[i];
    if (i < n - 2) t.q2[i] = -s.up1[i] / s.dia[i];
    if (i < n - 4) t.r2[i] = -s.up2[i] / s.dia[i];
  }
  return t;
}

// transfer of the rows [j0, j1) of one parity taken as one block: forward factor, backward 2 x 2 matrix, and the backward
// end state per unit of forward inflow (zero backward inflow).  An empty chain keeps its inflow.
static void colhh_block_transfer(const ColHhHost& h, int j0, int j1, int par, std::vector<long double>& ya, double* m1,
                                 double* m2, double* g) {
  // forward chain (ascending): response to a unit inflow y_{j0+par-2} = 1
  long double y = 1.0L;
  for (int j = j0 + par; j < j1; j += 2) { y = (long double)h.q1[j] * y; ya[j] = y; }
  *m1 = (double)y;
  // backward chain (descending): runs from the inflow states (1,0) and (0,1); third column: zero inflow state,
  // driven by the forward response above (what a unit of forward inflow leaves at the block's lower end)
  long double x1[3] = {1.0L, 0.0L, 0.0L}, x2[3] = {0.0L, 1.0L, 0.0L};
  int jt = j1 - 1;
  if ((jt & 1) != par) --jt;                // highest row of this parity in the block
  for (int j = jt; j >= j0; j -= 2) {
    for (int c = 0; c < 3; ++c) {
      long double nw = (long double)h.q2[j] * x1[c] + (long double)h.r2[j] * x2[c];
      if (c == 2) nw += (long double)h.p2[j] * ya[j];
      x2[c] = x1[c]; x1[c] = nw;
    }
  }
  m2[0] = (double)x1[0]; m2[1] = (double)x1[1]; m2[2] = (double)x2[0]; m2[3] = (double)x2[1];
  g[0] = (double)x1[2]; g[1] = (double)x2[2];
}

ColHhHost build_colhh_tables(const Mv3Tables& pv, const FdmaTables& f, int BR, int row0, int jend, const std::vector<int>* ranks) {
  RPDE_REQUIRE(BR >= 2 && BR % 2 == 0, "column-scan block must hold an even number of rows");
  RPDE_REQUIRE(row0 >= 0 && row0 % 2 == 0, "column-scan blocks start at an even row");
  ColHhHost h;
  const int n = (int)f.p2.size();
  if (jend < 0 || jend > n) jend = n;
  h.n = n; h.BR = BR; h.NB = (jend > row0) ? (jend - row0 + BR - 1) / BR : 0;
  const size_t np = (size_t)n + 2 * BR + 8;   // zero padding: rows past the system come out as zeros
  auto padded = [&](const Vec& v) { Vec o(np, 0.0); std::copy(v.begin(), v.begin() + std::min<size_t>(v.size(), n), o.begin()); return o; };
  h.t0 = padded(pv.t0); h.t1 = padded(pv.t1); h.t2 = padded(pv.t2);
  h.q1 = padded(f.q1); h.p2 = padded(f.p2); h.q2 = padded(f.q2); h.r2 = padded(f.r2);
  h.m1.assign((size_t)std::max(h.NB, 1) * 2, 1.0);
  h.m2.assign((size_t)std::max(h.NB, 1) * 8, 0.0);
  h.g.assign((size_t)std::max(h.NB, 1) * 4, 0.0);
  std::vector<long double> ya(np, 0.0L);
  for (int b = 0; b < h.NB; ++b) {
    const int j0 = row0 + b * BR, j1 = std::min(j0 + BR, jend);
    for (int par = 0; par < 2; ++par)
      colhh_block_transfer(h, j0, j1, par, ya, &h.m1[(size_t)b * 2 + par], &h.m2[((size_t)b * 2 + par) * 4],
                           &h.g[((size_t)b * 2 + par) * 2]);
  }
  if (ranks) {   // every rank's rows as one block: [nranks][14] = m1[2], m2[2][4], g[2][2]
    const int P = (int)ranks->size() - 1;
    h.rk.assign((size_t)P * 14, 0.0);
    for (int r = 0; r < P; ++r) {
      const int j0 = std::min((*ranks)[r], n), j1 = std::min((*ranks)[r + 1], n);
      RPDE_REQUIRE(j0 % 2 == 0 || j0 == n, "column scans: a rank's first row must be even");
      for (int par = 0; par < 2; ++par)
        colhh_block_transfer(h, j0, j1, par, ya, &h.rk[(size_t)r * 14 + par], &h.rk[(size_t)r * 14 + 2 + par * 4],
                             &h.rk[(size_t)r * 14 + 10 + par * 2]);
    }
  }
  return h;
}

ColHh1Host build_colhh1_tables(const ColHhHost& h, int W) {
  RPDE_REQUIRE(W >= 1 && h.BR > 0, "colhh1: blocks per workgroup");
  ColHh1Host o;
  o.W = W; o.NSB = (h.NB + W - 1) / W;
  const size_t np = h.q1.size();
  const int n = h.n, BR = h.BR;
  o.F.assign(np, 0.0); o.H0.assign(np, 0.0); o.H1.assign(np, 0.0);
  std::vector<long double> ya(np, 0.0L);
  for (int b = 0; b < h.NB; ++b) {
    const int j0 = b * BR, j1 = std::min(j0 + BR, n);
    for (int par = 0; par < 2; ++par) {
      long double y = 1.0L;                     // forward response to a unit inflow, as in colhh_block_transfer
      for (int j = j0 + par; j < j1; j += 2) { y = (long double)h.q1[j] * y; ya[j] = y; }
      long double x1[3] = {1.0L, 0.0L, 0.0L}, x2[3] = {0.0L, 1.0L, 0.0L};
      int jt = j1 - 1;
      if ((jt & 1) != par) --jt;
      for (int j = jt; j >= j0; j -= 2) {
        for (int c = 0; c < 3; ++c) {
          long double nw = (long double)h.q2[j] * x1[c] + (long double)h.r2[j] * x2[c];
          if (c == 2) nw += (long double)h.p2[j] * ya[j];
          x2[c] = x1[c]; x1[c] = nw;
        }
        o.H0[j] = (double)x1[0]; o.H1[j] = (double)x1[1]; o.F[j] = (double)x1[2];
      }
    }
  }
  o.m1w.assign((size_t)std::max(o.NSB, 1) * 2, 1.0);
  o.m2w.assign((size_t)std::max(o.NSB, 1) * 8, 0.0);
  o.gw.assign((size_t)std::max(o.NSB, 1) * 4, 0.0);
  for (int q = 0; q < o.NSB; ++q) {
    const int j0 = q * W * BR, j1 = std::min(j0 + W * BR, n);
    for (int par = 0; par < 2; ++par)
      colhh_block_transfer(h, j0, j1, par, ya, &o.m1w[(size_t)q * 2 + par], &o.m2w[((size_t)q * 2 + par) * 4], &o.gw[((size_t)q * 2 + par) * 2]);
  }
  return o;
}

ColCorrHost build_colcorr_tables(const Base& bd, const Base& bn, double dscale, int BR, int row0, int jend, const std::vector<int>* ranks) {
  RPDE_REQUIRE(bd.is_composite() && bn.is_composite() && bd.m == bn.m && bd.n == bn.n, "colcorr: two composite bases of one size");
  const int m = bd.m, n = bd.n;
  const Vec lowd = stencil_low(bd), lown = stencil_low(bn);
  for (int k = 0; k < m; ++k) RPDE_REQUIRE(lowd[k] == -1.0, "colcorr: the target base must carry the Dirichlet stencil");
  const FromOrthoTables fo = from_ortho_tables(bd);
  FdmaTables f{fo.q_up, Vec(m, 1.0), fo.q_dn, Vec(m, 0.0)};
  auto ln = [&](int k) { return (k >= 0 && k < m) ? lown[k] : 0.0; };
  ColCorrHost out;
  {   // a: rhs_k = c_k + lowd_k c_{k+2}, c = S_N ps
    Mv3Tables t{Vec(m), Vec(m), Vec(m)};
    for (int k = 0; k < m; ++k) {
      t.t0[k] = fo.p_up[k] * ln(k - 2);
      t.t1[k] = fo.p_up[k] * (1.0 + lowd[k] * ln(k));
      t.t2[k] = fo.p_up[k] * lowd[k];
    }
    out.a = build_colhh_tables(t, f, BR, row0, jend, ranks);
  }
  {   // b: rhs_k = dscale 2 (k + 1) c_{k+1} (k >= 1), rhs_0 = dscale (c_1 - d_2 / 2)
    Mv3Tables t{Vec(m), Vec(m), Vec(m, 0.0)};
    for (int k = 0; k < m; ++k) {
      const double sk = (k == 0) ? dscale : dscale * 2.0 * (double)(k + 1);
      t.t0[k] = fo.p_up[k] * sk * ln(k - 1);
      t.t1[k] = fo.p_up[k] * sk;
    }
    out.b = build_colhh_tables(t, f, BR, row0, jend, ranks);
    const size_t np = out.b.t0.size();
    // kappa = -dscale / 2 * d_2,  d_2 = sum_{odd j >= 3} 2 j c_j,  c_j = ps_j + lown_{j-2} ps_{j-2}
    out.b.w.assign(np, 0.0);
    for (int j = 1; j < m; j += 2) {
      double wj = (j >= 3) ? 2.0 * j : 0.0;
      if (j + 2 <= n - 1) wj += 2.0 * (j + 2) * lown[j];
      out.b.w[j] = -0.5 * dscale * wj;
    }
    // response to a unit right-hand side in row 0 (even chain): g_0 = p_0, g_k = q_k g_{k-2}; x_k = g_k + qd_k x_{k+2}
    out.b.h.assign(np, 0.0);
    std::vector<long double> g(m, 0.0L);
    long double y = 0.0L;
    for (int k = 0; k < m; k += 2) { y = (k == 0) ? (long double)fo.p_up[0] : (long double)fo.q_up[k] * y; g[k] = y; }
    long double x = 0.0L;
    for (int k = ((m - 1) & ~1); k >= 0; k -= 2) { x = g[k] + (long double)fo.q_dn[k] * x; out.b.h[k] = (double)x; }
  }
  return out;
}

// ------------------------------------------------------------------------------------- LAPACK
namespace {
using dgeev_t = void (*)(const char*, const char*, const int*, double*, const int*, double*,
                         double*, double*, const int*, double*, const int*, double*, const int*,
                         int*, size_t, size_t);
using dgetrf_t = void (*)(const int*, const int*, double*, const int*, int*, int*);
using dgetri_t = void (*)(const int*, double*, const int*, const int*, double*, const int*, int*);
using dgemm_t = void (*)(const char*, const char*, const int*, const int*, const int*,
                         const double*, const double*, const int*, const double*, const int*,
                         const double*, double*, const int*, size_t, size_t);

struct Lapack {
  void* h = nullptr;
  std::string path;
  dgeev_t dgeev = nullptr;
  dgetrf_t dgetrf = nullptr;
  dgetri_t dgetri = nullptr;
  dgemm_t dgemm = nullptr;
};

void* sym2(void* h, const char* a, const char* b) {
  void* p = dlsym(h, a);
  return p ? p : dlsym(h, b);
}

bool try_open(Lapack& L, const std::string& path) {
  void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) return false;
  Lapack t;
  t.h = h;
  t.path = path;
  t.dgeev = (dgeev_t)sym2(h, "scipy_dgeev_", "dgeev_");
  t.dgetrf = (dgetrf_t)sym2(h, "scipy_dgetrf_", "dgetrf_");
  t.dgetri = (dgetri_t)sym2(h, "scipy_dgetri_", "dgetri_");
  t.dgemm = (dgemm_t)sym2(h, "scipy_dgemm_", "dgemm_");
  if (t.dgeev && t.dgetrf && t.dgetri && t.dgemm) { L = t; return true; }
  dlclose(h);
  return false;
}

Lapack& lapack() {
  static Lapack L;
  if (L.h) return L;
  // search order: RPDE_LAPACK_LIB (explicit), then the system's OpenBLAS / LAPACK by soname (what a
  // Rust host links anyway: Cargo.toml:38-53 openblas-system / intel-mkl), and only then the
  // OpenBLAS that ships inside SciPy wheels (the one this Python-only image happens to have)
  std::vector<std::string> cand;
  if (const char* e = std::getenv("RPDE_LAPACK_LIB")) cand.push_back(e);
  for (const char* s : {"libopenblas.so.0", "libopenblas.so", "liblapack.so.3", "liblapack.so", "libmkl_rt.so.2", "libmkl_rt.so"})
    cand.push_back(s);
  const char* pats[] = {
      "/usr/local/lib/python3*/dist-packages/scipy.libs/libscipy_openblas*.so",
      "/usr/lib/python3*/dist-packages/scipy.libs/libscipy_openblas*.so",
      "/usr/local/lib/python3*/site-packages/scipy.libs/libscipy_openblas*.so",
      "/opt/conda/lib/python3*/site-packages/scipy.libs/libscipy_openblas*.so",
  };
  for (const char* p : pats) {
    glob_t g;
    if (glob(p, 0, nullptr, &g) == 0)
      for (size_t i = 0; i < g.gl_pathc; ++i) cand.push_back(g.gl_pathv[i]);
    globfree(&g);
  }
  for (const auto& c : cand)
    if (try_open(L, c)) return L;
  fail("rustpde_hip: no LAPACK (dgeev/dgetrf/dgetri/dgemm) found for the Poisson eigen-"
       "decomposition; set RPDE_LAPACK_LIB to an OpenBLAS/LAPACK shared library");
}

// column-major helpers
void invert_cm(Lapack& L, Vec& a, int n) {
  std::vector<int> ipiv(n);
  int info = 0;
  L.dgetrf(&n, &n, a.data(), &n, ipiv.data(), &info);
  RPDE_REQUIRE(info == 0, "dgetrf failed");
  int lwork = -1;
  double wq = 0;
  L.dgetri(&n, a.data(), &n, ipiv.data(), &wq, &lwork, &info);
  lwork = (int)wq + 1;
  Vec work(lwork);
  L.dgetri(&n, a.data(), &n, ipiv.data(), work.data(), &lwork, &info);
  RPDE_REQUIRE(info == 0, "dgetri failed");
}
Vec matmul_cm(Lapack& L, const Vec& a, const Vec& b, int n) {
  Vec c((size_t)n * n);
  const double one = 1.0, zero = 0.0;
  L.dgemm("N", "N", &n, &n, &n, &one, a.data(), &n, b.data(), &n, &zero, c.data(), &n, 1, 1);
  return c;
}
}  // namespace

std::string lapack_library_path() { return lapack().path; }

// RPDE_EIG_CACHE=<directory> (optional; the GPU tests and the evidence scripts set it): the decomposition of a pencil is kept
// in a file named after the pencil's bytes and read back by later engines of the same operator -- a 4097-point axis costs two
// dgeev of 2048 x 2048 (tens of seconds on the host), and a test run builds that engine dozens of times.  What is stored is
// what LAPACK returned the first time; nothing on the device side changes.
namespace {
std::string eig_cache_file(const Bands& a, const Bands& c) {
  const char* dir = std::getenv("RPDE_EIG_CACHE");
  if (!dir || !*dir) return std::string();
  unsigned long long h = 1469598103934665603ull;   // FNV-1a over the eight bands
  auto mix = [&](const Vec& v) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(v.data());
    for (size_t i = 0; i < v.size() * sizeof(double); ++i) { h ^= p[i]; h *= 1099511628211ull; }
  };
  mix(a.low); mix(a.dia); mix(a.up1); mix(a.up2); mix(c.low); mix(c.dia); mix(c.up1); mix(c.up2);
  char name[96];
  snprintf(name, sizeof name, "/eigx_%d_%016llx.bin", (int)a.dia.size(), h);
  return std::string(dir) + name;
}
bool eig_cache_load(const std::string& path, int m, EigenX& out) {
  FILE* f = path.empty() ? nullptr : std::fopen(path.c_str(), "rb");
  if (!f) return false;
  long long hdr[3] = {0, 0, 0};
  bool ok = std::fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == m && hdr[1] == (m + 1) / 2 && hdr[2] == m / 2;
  if (ok) {
    out.me = (int)hdr[1]; out.mo = (int)hdr[2];
    const size_t nm = (size_t)out.me * out.me + (size_t)out.mo * out.mo;
    out.lam.resize(m); out.fwd.resize(nm); out.bwd.resize(nm);
    ok = std::fread(out.lam.data(), sizeof(double), m, f) == (size_t)m && std::fread(out.fwd.data(), sizeof(double), nm, f) == nm &&
         std::fread(out.bwd.data(), sizeof(double), nm, f) == nm && std::fgetc(f) == EOF;
  }
  std::fclose(f);
  return ok;
}
void eig_cache_store(const std::string& path, const EigenX& e) {
  if (path.empty()) return;
  const std::string tmp = path + ".tmp" + std::to_string((long long)getpid());
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return;                                   // an unwritable directory only costs the time the cache would have saved
  const long long hdr[3] = {(long long)e.lam.size(), e.me, e.mo};
  bool ok = std::fwrite(hdr, sizeof hdr, 1, f) == 1 && std::fwrite(e.lam.data(), sizeof(double), e.lam.size(), f) == e.lam.size() &&
            std::fwrite(e.fwd.data(), sizeof(double), e.fwd.size(), f) == e.fwd.size() &&
            std::fwrite(e.bwd.data(), sizeof(double), e.bwd.size(), f) == e.bwd.size();
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) std::remove(tmp.c_str());
}
}  // namespace

EigenX eigen_decomposition_parity(const Bands& a, const Bands& c) {
  const std::string cache = eig_cache_file(a, c);
  {
    EigenX hit;
    if (eig_cache_load(cache, (int)a.dia.size(), hit)) return hit;
  }
  Lapack& L = lapack();
  const int m = (int)a.dia.size();
  EigenX out;
  out.me = (m + 1) / 2;
  out.mo = m / 2;
  out.lam.resize(m);
  out.fwd.resize((size_t)out.me * out.me + (size_t)out.mo * out.mo);
  out.bwd.resize(out.fwd.size());
  size_t moff = 0;
  int loff = 0;
  for (int par = 0; par < 2; ++par) {
    const int mb = par == 0 ? out.me : out.mo;
    auto dense_cm = [&](const Bands& bd) {
      Vec d((size_t)mb * mb, 0.0);
      for (int r = 0; r < mb; ++r) {
        const int R = par + 2 * r;
        auto put = [&](int cc, double v) { if (cc >= 0 && cc < mb) d[(size_t)cc * mb + r] = v; };
        put(r - 1, bd.low[R]);
        put(r, bd.dia[R]);
        put(r + 1, bd.up1[R]);
        put(r + 2, bd.up2[R]);
      }
      return d;
    };
    Vec cinv = dense_cm(c);
    invert_cm(L, cinv, mb);
    Vec x = matmul_cm(L, cinv, dense_cm(a), mb);
    Vec wr(mb), wi(mb), vr((size_t)mb * mb);
    int info = 0, lwork = -1, one = 1;
    double wq = 0;
    L.dgeev("N", "V", &mb, x.data(), &mb, wr.data(), wi.data(), nullptr, &one, vr.data(), &mb, &wq,
            &lwork, &info, 1, 1);
    lwork = (int)wq + 1;
    Vec work(lwork);
    L.dgeev("N", "V", &mb, x.data(), &mb, wr.data(), wi.data(), nullptr, &one, vr.data(), &mb,
            work.data(), &lwork, &info, 1, 1);
    RPDE_REQUIRE(info == 0, "dgeev failed");
    {  // the x operator has a real spectrum; the reference keeps the real parts
       // (src/solver/utils.rs:82-87) -- a complex pair here would mean a broken operator, not round-off
      double wmax = 0.0, imax = 0.0;
      for (int k = 0; k < mb; ++k) { wmax = std::max(wmax, std::fabs(wr[k])); imax = std::max(imax, std::fabs(wi[k])); }
      RPDE_REQUIRE(imax <= 1e-8 * std::max(wmax, 1.0), "Poisson eigen-decomposition: complex eigenvalues");
    }
    std::vector<int> perm(mb);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int i, int j) { return wr[i] > wr[j]; });
    Vec q((size_t)mb * mb);
    for (int k = 0; k < mb; ++k)
      std::copy(vr.begin() + (size_t)perm[k] * mb, vr.begin() + (size_t)(perm[k] + 1) * mb,
                q.begin() + (size_t)k * mb);
    Vec qinv = q;
    invert_cm(L, qinv, mb);
    Vec f = matmul_cm(L, qinv, cinv, mb);
    for (int k = 0; k < mb; ++k) out.lam[loff + k] = wr[perm[k]];
    for (int i = 0; i < mb; ++i)
      for (int k = 0; k < mb; ++k) {
        out.bwd[moff + (size_t)i * mb + k] = q[(size_t)k * mb + i];   // Q(i,k)
        out.fwd[moff + (size_t)k * mb + i] = f[(size_t)i * mb + k];   // F(k,i)
      }
    moff += (size_t)mb * mb;
    loff += mb;
  }
  eig_cache_store(cache, out);
  return out;
}

Vec eigen_spectrum_parity(const Bands& a, const Bands& c) {
  Lapack& L = lapack();
  const int m = (int)a.dia.size();
  const int me = (m + 1) / 2, mo = m / 2;
  Vec lam(m);
  int loff = 0;
  for (int par = 0; par < 2; ++par) {
    const int mb = par == 0 ? me : mo;
    auto dense_cm = [&](const Bands& bd) {
      Vec d((size_t)mb * mb, 0.0);
      for (int r = 0; r < mb; ++r) {
        const int R = par + 2 * r;
        auto put = [&](int cc, double v) { if (cc >= 0 && cc < mb) d[(size_t)cc * mb + r] = v; };
        put(r - 1, bd.low[R]); put(r, bd.dia[R]); put(r + 1, bd.up1[R]); put(r + 2, bd.up2[R]);
      }
      return d;
    };
    Vec cinv = dense_cm(c);
    invert_cm(L, cinv, mb);
    Vec x = matmul_cm(L, cinv, dense_cm(a), mb);
    Vec wr(mb), wi(mb);
    int info = 0, lwork = -1, one = 1;
    double wq = 0;
    L.dgeev("N", "N", &mb, x.data(), &mb, wr.data(), wi.data(), nullptr, &one, nullptr, &one, &wq, &lwork, &info, 1, 1);
    lwork = (int)wq + 1;
    Vec work(lwork);
    L.dgeev("N", "N", &mb, x.data(), &mb, wr.data(), wi.data(), nullptr, &one, nullptr, &one, work.data(), &lwork, &info, 1, 1);
    RPDE_REQUIRE(info == 0, "dgeev failed");
    double wmax = 0.0, imax = 0.0;
    for (int k = 0; k < mb; ++k) { wmax = std::max(wmax, std::fabs(wr[k])); imax = std::max(imax, std::fabs(wi[k])); }
    RPDE_REQUIRE(imax <= 1e-8 * std::max(wmax, 1.0), "Poisson eigen-decomposition: complex eigenvalues");
    std::stable_sort(wr.begin(), wr.end(), [](double x0, double x1) { return x0 > x1; });
    std::copy(wr.begin(), wr.end(), lam.begin() + loff);
    loff += mb;
  }
  return lam;
}

namespace {
// LU with partial pivoting of an n x n band matrix (kl sub-, ku super-diagonals), row storage with room for the fill:
// row i holds columns i - kl .. i + ku + kl at ab[i * w + (j - i + kl)], w = 2 kl + ku + 1.  Rows are swapped physically
// (a pivot row reaches at most kl rows down).  A pivot that vanishes against the matrix norm is replaced by `tiny`:
// the matrix is singular ON PURPOSE (inverse iteration at an eigenvalue), the solve then returns the null direction.
struct BandLU {
  int n = 0, kl = 0, ku = 0, w = 0;
  Vec ab;                  // factorised in place: U in the row, multipliers in mult
  Vec mult;                // mult[i * kl + (r - i - 1)]: multiplier of row r (after the swap of step i)
  std::vector<int> piv;    // row swapped with row i in step i
  double& at(int i, int j) { return ab[(size_t)i * w + (j - i + kl)]; }
  void factor(double tiny) {
    mult.assign((size_t)n * std::max(kl, 1), 0.0);
    piv.assign(n, 0);
    for (int i = 0; i < n; ++i) {
      int p = i;
      double best = std::fabs(at(i, i));
      for (int r = i + 1; r <= std::min(n - 1, i + kl); ++r) {
        const double v = std::fabs(ab[(size_t)r * w + (i - r + kl)]);
        if (v > best) { best = v; p = r; }
      }
      piv[i] = p;
      const int jmax = std::min(n - 1, i + ku + kl);
      if (p != i)
        for (int j = i; j <= jmax; ++j) std::swap(at(i, j), ab[(size_t)p * w + (j - p + kl)]);
      if (std::fabs(at(i, i)) < tiny) at(i, i) = at(i, i) < 0.0 ? -tiny : tiny;
      const double d = at(i, i);
      for (int r = i + 1; r <= std::min(n - 1, i + kl); ++r) {
        double& e = ab[(size_t)r * w + (i - r + kl)];
        const double f = e / d;
        mult[(size_t)i * kl + (r - i - 1)] = f;
        e = 0.0;
        if (f != 0.0)
          for (int j = i + 1; j <= jmax; ++j) ab[(size_t)r * w + (j - r + kl)] -= f * at(i, j);
      }
    }
  }
  void solve(Vec& x) {     // in place
    for (int i = 0; i < n; ++i) {
      if (piv[i] != i) std::swap(x[i], x[piv[i]]);
      for (int r = i + 1; r <= std::min(n - 1, i + kl); ++r) x[r] -= mult[(size_t)i * kl + (r - i - 1)] * x[i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = x[i];
      const int jmax = std::min(n - 1, i + ku + kl);
      for (int j = i + 1; j <= jmax; ++j) v -= at(i, j) * x[j];
      x[i] = v / at(i, i);
    }
  }
};
}  // namespace

EigenX eigenbasis_from_spectrum(const Bands& a, const Bands& c, const Vec& lam) {
  const int m = (int)a.dia.size();
  RPDE_REQUIRE((int)lam.size() == m, "eigenbasis_from_spectrum: one eigenvalue per coefficient");
  EigenX out;
  out.me = (m + 1) / 2;
  out.mo = m / 2;
  out.lam = lam;
  out.fwd.assign((size_t)out.me * out.me + (size_t)out.mo * out.mo, 0.0);
  out.bwd.assign(out.fwd.size(), 0.0);
  size_t moff = 0;
  int loff = 0;
  for (int par = 0; par < 2; ++par) {
    const int mb = par == 0 ? out.me : out.mo;
    // entries of row r of a parity block: columns r - 1, r, r + 1, r + 2
    auto ent = [&](const Bands& bd, int r, int off) {
      const int R = par + 2 * r;
      return off == -1 ? bd.low[R] : off == 0 ? bd.dia[R] : off == 1 ? bd.up1[R] : bd.up2[R];
    };
    double norm = 0.0;
    for (int r = 0; r < mb; ++r)
      for (int off = -1; off <= 2; ++off) norm = std::max(norm, std::fabs(ent(a, r, off)));
    auto band_mul = [&](const Bands& bd, const Vec& x, Vec& y) {   // y = (block of bd) x
      for (int r = 0; r < mb; ++r) {
        double v = 0.0;
        for (int off = -1; off <= 2; ++off) { const int jc = r + off; if (jc >= 0 && jc < mb) v += ent(bd, r, off) * x[jc]; }
        y[r] = v;
      }
    };
    Vec q(mb), f(mb), aq(mb), cq(mb);
    for (int k = 0; k < mb; ++k) {
      // Rayleigh-quotient iteration on the banded pencil: dgeev works on the dense inv(C) A, whose norm grows like n^4, and
      // returns the small eigenvalues with an ABSOLUTE error of eps * n^4 (1e-7 relative at n = 1025); against the pencil
      // itself the generalised Rayleigh quotient rho = (f A q) / (f C q) of a left / right pair converges in two or three
      // rounds to the eigenvalue the operator has (residual |A q - rho C q| at round-off level)
      double l = lam[loff + k];
      for (int round = 0; round < 6; ++round) {
        double mnorm = norm;
        for (int r = 0; r < mb; ++r)
          for (int off = -1; off <= 2; ++off) mnorm = std::max(mnorm, std::fabs(l * ent(c, r, off)));
        const double tiny = 2.220446049250313e-16 * mnorm;
        BandLU R_, L_;                                // M = A - l C (kl 1, ku 2) and its transpose (kl 2, ku 1)
        R_.n = L_.n = mb;
        R_.kl = 1; R_.ku = 2; R_.w = 2 * R_.kl + R_.ku + 1; R_.ab.assign((size_t)mb * R_.w, 0.0);
        L_.kl = 2; L_.ku = 1; L_.w = 2 * L_.kl + L_.ku + 1; L_.ab.assign((size_t)mb * L_.w, 0.0);
        for (int r = 0; r < mb; ++r)
          for (int off = -1; off <= 2; ++off) {
            const int jc = r + off;
            if (jc < 0 || jc >= mb) continue;
            const double v = ent(a, r, off) - l * ent(c, r, off);
            R_.at(r, jc) = v;
            L_.at(jc, r) = v;
          }
        R_.factor(tiny);
        L_.factor(tiny);
        auto iterate = [&](BandLU& lu, Vec& x, int its) {
          if (round == 0)
            for (int i = 0; i < mb; ++i) x[i] = 1.0 / (1.0 + (double)((i * 7 + 3) % 11));   // a fixed start with every component
          for (int it = 0; it < its; ++it) {
            lu.solve(x);
            double mx = 0.0;
            for (double v : x) mx = std::max(mx, std::fabs(v));
            RPDE_REQUIRE(mx > 0.0 && mx == mx && mx < 1e300, "inverse iteration failed");
            for (double& v : x) v /= mx;
          }
        };
        iterate(R_, q, round == 0 ? 3 : 1);
        iterate(L_, f, round == 0 ? 3 : 1);
        band_mul(a, q, aq);
        band_mul(c, q, cq);
        double faq = 0.0, fcq = 0.0;
        for (int r = 0; r < mb; ++r) { faq += f[r] * aq[r]; fcq += f[r] * cq[r]; }
        RPDE_REQUIRE(fcq != 0.0 && fcq == fcq, "left and right eigenvector are orthogonal in the C inner product");
        const double rho = faq / fcq;
        const bool done = std::fabs(rho - l) <= 8.0 * 2.220446049250313e-16 * std::max(std::fabs(l), 1e-3 * norm / std::max(1.0, (double)mb));
        // a refined value must stay the eigenvalue it started from: well inside the gap to its neighbours
        l = rho;
        if (done || round == 5) break;
      }
      out.lam[loff + k] = l;
      {  // q: unit 2-norm, its largest component positive
        double s2 = 0.0, big = 0.0;
        int ib = 0;
        for (int i = 0; i < mb; ++i) { s2 += q[i] * q[i]; if (std::fabs(q[i]) > big) { big = std::fabs(q[i]); ib = i; } }
        const double sc = (q[ib] < 0.0 ? -1.0 : 1.0) / std::sqrt(s2);
        for (double& v : q) v *= sc;
      }
      {  // f: f C q = 1
        band_mul(c, q, cq);
        double dot = 0.0;
        for (int r = 0; r < mb; ++r) dot += f[r] * cq[r];
        RPDE_REQUIRE(dot != 0.0 && dot == dot, "left and right eigenvector are orthogonal in the C inner product");
        for (double& v : f) v /= dot;
      }
      for (int i = 0; i < mb; ++i) {
        out.bwd[moff + (size_t)i * mb + k] = q[i];   // Q(i, k)
        out.fwd[moff + (size_t)k * mb + i] = f[i];   // F(k, i)
      }
    }
    moff += (size_t)mb * mb;
    loff += mb;
  }
  return out;
}

}  // namespace rpde
