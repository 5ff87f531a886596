// extern "C" boundary (include/rustpde_hip.h).  Exceptions never cross it.
#include "../../include/rustpde_hip.h"

#include <cmath>
#include <cstring>
#include <string>

#include "engine.h"
#include "adjoint.h"
#include "dct_line.h"
#include "h5lite.h"
#include "rccl_transport.h"

using namespace rpde;

static thread_local std::string g_err;

struct rpde_navier2d { Navier2DEngine* e; int device; };
struct rpde_space2 { Space2Ops* sp; Stream st; int device; };
struct rpde_hholtz_adi { HholtzAdiOp* op; rpde_space2* s; };
struct rpde_poisson { PoissonOp* op; rpde_space2* s; };
struct rpde_hholtz { TensorHholtzOp* op; rpde_space2* s; };
struct rpde_adjoint2d { Navier2DAdjointEngine* e; int device; };
struct rpde_lnse2d { Navier2DLnseEngine* e; int device; };

static void select_device(int device) {
#ifndef RPDE_EMU
  RPDE_HIP(hipSetDevice(device));
#else
  (void)device;
#endif
}

#define RPDE_TRY(...)                                      \
  try {                                                    \
    __VA_ARGS__;                                           \
    return 0;                                              \
  } catch (const std::exception& ex) {                     \
    g_err = ex.what();                                     \
    return 1;                                              \
  } catch (...) {                                          \
    g_err = "unknown error";                               \
    return 2;                                              \
  }

#define RPDE_CHECK_HANDLE(h) RPDE_REQUIRE((h) != nullptr, "null handle")

extern "C" {

const char* rpde_last_error(void) { return g_err.c_str(); }
const char* rpde_version(void) {
#ifdef RPDE_EMU
  return "rustpde_hip 0.1 (host emulation build: tests only)";
#else
  return "rustpde_hip 0.1 (HIP gfx950)";
#endif
}
int rpde_is_device_build(void) {
#ifdef RPDE_EMU
  return 0;
#else
  return 1;
#endif
}
int rpde_device_count(int* count) {
  RPDE_TRY({
    RPDE_REQUIRE(count, "null pointer");
#ifdef RPDE_EMU
    *count = 0;
#else
    RPDE_HIP(hipGetDeviceCount(count));
#endif
  })
}

int rpde_device_memory(int device, size_t* slab_bytes, size_t* used_bytes) {
  RPDE_TRY({
    RPDE_REQUIRE(slab_bytes && used_bytes, "null pointer");
#ifdef RPDE_EMU
    (void)device; *slab_bytes = 0; *used_bytes = 0;
#else
    *slab_bytes = DevArena::get().slab_bytes(device);
    *used_bytes = DevArena::get().used_bytes(device);
#endif
  })
}
int rpde_device_trim(int device, size_t* released_bytes) {
  RPDE_TRY({
    size_t r = 0;
#ifndef RPDE_EMU
    if (dev_arena_on()) r = DevArena::get().trim(device);
#else
    (void)device;
#endif
    if (released_bytes) *released_bytes = r;
  })
}
int rpde_arena_check(long* violations) {
  RPDE_TRY({
    RPDE_REQUIRE(violations, "null pointer");
#ifdef RPDE_EMU
    *violations = 0;
#else
    *violations = dev_arena_on() ? DevArena::get().check() : 0;
#endif
  })
}
int rpde_arena_selftest(void) {
  RPDE_TRY({
    using A = ArenaT<ArenaHostBackend>;
    int& cur = ArenaHostBackend::current();
    const int saved = cur;
    struct Restore { int& c; int v; ~Restore() { c = v; } } restore{cur, saved};
    {
      A a(false);
      cur = 0;
      void* p0 = a.alloc(1000);
      void* q0 = a.alloc(5 << 20);
      RPDE_REQUIRE(p0 && q0 && a.device_of(p0) == 0 && a.device_of(q0) == 0, "device 0 blocks");
      RPDE_REQUIRE(a.slab_bytes(0) == A::kFirstSlabBytes && a.slab_bytes(1) == 0, "one small first slab on device 0");
      a.free(q0);                                      // 5 MB free on a device-0 slab ...
      cur = 1;
      void* p1 = a.alloc(5 << 20);                     // ... which an allocation for device 1 must not take
      RPDE_REQUIRE(p1 && a.device_of(p1) == 1, "a device-1 block lies in a device-1 slab");
      RPDE_REQUIRE(a.slab_bytes(1) == A::kFirstSlabBytes && a.slab_bytes(0) == A::kFirstSlabBytes, "device 1 got its own slab");
      RPDE_REQUIRE(a.used_bytes(0) == 4096 && a.used_bytes(1) == (size_t(5) << 20), "per-device accounting");
      a.free(p0);                                      // freed while ANOTHER device is current: back to its own slab
      RPDE_REQUIRE(a.used_bytes(0) == 0 && a.used_bytes(1) == (size_t(5) << 20), "free finds the owning slab");
      RPDE_REQUIRE(a.trim(1) == 0, "a slab with a live block stays");
      RPDE_REQUIRE(a.trim(0) == A::kFirstSlabBytes && a.slab_bytes(0) == 0, "trim(0) releases device 0 only");
      cur = 0;
      void* r0 = a.alloc(size_t(300) << 20);           // larger than the first slab: a slab of its own size
      RPDE_REQUIRE(r0 && a.device_of(r0) == 0 && a.slab_bytes(0) == (size_t(300) << 20), "oversized request");
      void* s0 = a.alloc(1 << 20);                     // second slab of the device: the large size
      RPDE_REQUIRE(s0 && a.slab_bytes(0) == (size_t(300) << 20) + A::kSlabBytes, "second slab");
      a.free(r0); a.free(s0); a.free(p1);
      RPDE_REQUIRE(a.trim(-1) == (size_t(300) << 20) + A::kSlabBytes + A::kFirstSlabBytes && a.slab_bytes(-1) == 0, "trim(-1)");
    }
    {
      A g(true);                                       // guards: a write behind a block is counted, a write inside is not
      cur = 0;
      char* p = static_cast<char*>(g.alloc(1000));
      char* q = static_cast<char*>(g.alloc(8192));
      RPDE_REQUIRE(p && q, "guarded blocks");
      std::memset(p, 0, 1000); std::memset(q, 0, 8192);
      RPDE_REQUIRE(g.check() == 0, "clean guards");
      p[1000] = 0;                                     // first byte behind the 1000 requested
      RPDE_REQUIRE(g.check() == 1, "overrun by one byte");
      q[8192 + 4095] = 1;                              // last byte of the guard granule
      RPDE_REQUIRE(g.check() == 2, "overrun into the guard granule");
      g.free(p);
      RPDE_REQUIRE(g.check() == 2, "free() counts the block it releases, check() the live one");
      g.free(q);
      RPDE_REQUIRE(g.check() == 1 && g.check() == 0, "counted once");
      g.trim(-1);
    }
  })
}

static int create_engine(int nx, int ny, double ra, double pr, double dt, double aspect,
                         const char* bc, int device, bool periodic, rpde_navier2d** out) {
  RPDE_TRY({
    RPDE_REQUIRE(out && bc, "null pointer");
    select_device(device);
    auto* h = new rpde_navier2d{nullptr, device};
    try {
      h->e = new Navier2DEngine(nx, ny, ra, pr, dt, aspect, bc, periodic);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  })
}
int rpde_navier2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect,
                                  const char* bc, int device, rpde_navier2d** out) {
  return create_engine(nx, ny, ra, pr, dt, aspect, bc, device, false, out);
}
int rpde_navier2d_create_confined_with_spectrum(int nx, int ny, double ra, double pr, double dt, double aspect,
                                                const char* bc, int device, const double* lam, size_t m,
                                                rpde_navier2d** out) {
  if (!lam) { g_err = "null pointer"; return 1; }
  const Vec spectrum(lam, lam + m);
  struct Pending { explicit Pending(const Vec* v) { set_pending_x_spectrum(v); } ~Pending() { set_pending_x_spectrum(nullptr); } } guard(&spectrum);
  return create_engine(nx, ny, ra, pr, dt, aspect, bc, device, false, out);
}
int rpde_navier2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect,
                                  const char* bc, int device, rpde_navier2d** out) {
  return create_engine(nx, ny, ra, pr, dt, aspect, bc, device, true, out);
}
int rpde_navier2d_create_sharded(int periodic, int nx, int ny, double ra, double pr, double dt,
                                 double aspect, const char* bc, int device, int rank, int nranks,
                                 rpde_alltoallv_fn alltoallv, void* user, rpde_navier2d** out) {
  RPDE_TRY({
    RPDE_REQUIRE(out && bc, "null pointer");
    select_device(device);
    CommCb cb;
    cb.rank = rank; cb.size = nranks; cb.fn = alltoallv; cb.user = user;
    auto* h = new rpde_navier2d{nullptr, device};
    try {
      h->e = new Navier2DEngine(nx, ny, ra, pr, dt, aspect, bc, periodic != 0, &cb);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  })
}
int rpde_rccl_unique_id(char* id128) {
  RPDE_TRY({ RPDE_REQUIRE(id128, "null pointer"); rccl_unique_id(id128); })
}
int rpde_rccl_alltoallv_once(const char* id128, int rank, int nranks, int device, const double* send,
                             const int64_t* sendcounts, double* recv, const int64_t* recvcounts) {
  RPDE_TRY({
    RPDE_REQUIRE(id128 && sendcounts && recvcounts, "null pointer");
    select_device(device);
    RcclComm* c = rccl_comm_create(rank, nranks, id128);
    try {
      Stream st;
#ifndef RPDE_EMU
      RPDE_HIP(hipStreamCreate(&st.s));
#endif
      rccl_alltoallv(c, send, sendcounts, recv, recvcounts, st);
      dev_sync(st);
#ifndef RPDE_EMU
      (void)hipStreamDestroy(st.s);
#endif
    } catch (...) {
      rccl_comm_destroy(c);
      throw;
    }
    rccl_comm_destroy(c);
  })
}
int rpde_navier2d_create_sharded_rccl(int periodic, int nx, int ny, double ra, double pr, double dt,
                                      double aspect, const char* bc, int device, int rank, int nranks,
                                      const char* id128, rpde_navier2d** out) {
  RPDE_TRY({
    RPDE_REQUIRE(out && bc && id128, "null pointer");
    select_device(device);
    CommCb cb;
    cb.rank = rank; cb.size = nranks;
    cb.rccl = rccl_comm_create(rank, nranks, id128);   // collective: every rank is in this call
    auto* h = new rpde_navier2d{nullptr, device};
    try {
      h->e = new Navier2DEngine(nx, ny, ra, pr, dt, aspect, bc, periodic != 0, &cb);
    } catch (...) {
      rccl_comm_destroy(cb.rccl);
      delete h;
      throw;
    }
    *out = h;
  })
}
int rpde_navier2d_comm_stats(rpde_navier2d* h, double* bytes_per_step, int* exchanges_per_step) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(bytes_per_step && exchanges_per_step, "null pointer");
    *bytes_per_step = h->e->exchange_bytes_per_step();
    *exchanges_per_step = h->e->exchanges_per_step();
  })
}
int rpde_navier2d_destroy(rpde_navier2d* h) {
  RPDE_TRY({
    if (h) { select_device(h->device); delete h->e; delete h; dev_trim(); }
  })
}
// ---- Navier2DAdjoint (adjoint.h)
static int create_adjoint(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc, int device,
                          bool periodic, rpde_adjoint2d** out) {
  RPDE_TRY({
    RPDE_REQUIRE(out && bc, "null pointer");
    select_device(device);
    auto* h = new rpde_adjoint2d{nullptr, device};
    try {
      h->e = new Navier2DAdjointEngine(nx, ny, ra, pr, dt, aspect, bc, periodic);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  })
}
int rpde_adjoint2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc, int device,
                                   rpde_adjoint2d** out) {
  return create_adjoint(nx, ny, ra, pr, dt, aspect, bc, device, false, out);
}
int rpde_adjoint2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc, int device,
                                   rpde_adjoint2d** out) {
  return create_adjoint(nx, ny, ra, pr, dt, aspect, bc, device, true, out);
}
int rpde_adjoint2d_destroy(rpde_adjoint2d* h) {
  RPDE_TRY({ if (h) { select_device(h->device); delete h->e; delete h; dev_trim(); } })
}
int rpde_adjoint2d_set_velocity(rpde_adjoint2d* h, double amp, double m, double n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->set_velocity(amp, m, n); })
}
int rpde_adjoint2d_set_temperature(rpde_adjoint2d* h, double amp, double m, double n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->set_temperature(amp, m, n); })
}
int rpde_adjoint2d_reset_time(rpde_adjoint2d* h) { RPDE_TRY({ RPDE_CHECK_HANDLE(h); h->e->reset_time(); }) }
int rpde_adjoint2d_spectral_shape(rpde_adjoint2d* h, const char* name, int* rows, int* cols, int* is_complex) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && rows && cols && is_complex, "null pointer");
    int e = 1;
    h->e->spectral_shape(name, rows, cols, &e);
    *is_complex = e == 2;
  })
}
int rpde_adjoint2d_set_field(rpde_adjoint2d* h, const char* name, int space, const double* data, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && data, "null pointer");
    select_device(h->device);
    if (space == RPDE_PHYSICAL) h->e->set_field_physical(name, data, len);
    else if (space == RPDE_SPECTRAL) h->e->set_field_spectral(name, data, len);
    else fail("space must be RPDE_PHYSICAL or RPDE_SPECTRAL");
  })
}
int rpde_adjoint2d_get_field(rpde_adjoint2d* h, const char* name, int space, double* data, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && data, "null pointer");
    select_device(h->device);
    if (space == RPDE_PHYSICAL) h->e->get_field_physical(name, data, len);
    else if (space == RPDE_SPECTRAL) h->e->get_field_spectral(name, data, len);
    else fail("space must be RPDE_PHYSICAL or RPDE_SPECTRAL");
  })
}
int rpde_adjoint2d_update(rpde_adjoint2d* h, int nsteps) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(nsteps >= 0, "negative step count"); select_device(h->device); h->e->update(nsteps); })
}
int rpde_adjoint2d_time(rpde_adjoint2d* h, double* time) { RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(time, "null pointer"); *time = h->e->time(); }) }
int rpde_adjoint2d_dt(rpde_adjoint2d* h, double* dt) { RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(dt, "null pointer"); *dt = h->e->dt(); }) }
int rpde_adjoint2d_param(rpde_adjoint2d* h, const char* key, double* value) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(key && value, "null pointer"); *value = h->e->param(key); })
}
int rpde_adjoint2d_exit(rpde_adjoint2d* h, int* stop) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(stop, "null pointer"); select_device(h->device); *stop = h->e->exit() ? 1 : 0; })
}
int rpde_adjoint2d_div_norm(rpde_adjoint2d* h, double* norm) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(norm, "null pointer"); select_device(h->device); *norm = h->e->div_norm(); })
}
int rpde_adjoint2d_write(rpde_adjoint2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->write(filename); })
}
int rpde_adjoint2d_read(rpde_adjoint2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->read(filename); })
}
int rpde_adjoint2d_norm_residual(rpde_adjoint2d* h, double* res3) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(res3, "null pointer"); select_device(h->device); h->e->norm_residual(res3); })
}

// ---- Navier2DLnse (adjoint.h): the same entry points as the adjoint solver over the shared base
static int create_lnse(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc, const char* mean_file,
                       int device, bool periodic, rpde_lnse2d** out, bool nonlinear = false) {
  RPDE_TRY({
    RPDE_REQUIRE(out && bc, "null pointer");
    select_device(device);
    auto* h = new rpde_lnse2d{nullptr, device};
    try {
      h->e = new Navier2DLnseEngine(nx, ny, ra, pr, dt, aspect, bc, periodic, mean_file ? mean_file : "mean.h5", nonlinear);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
  })
}
int rpde_lnse2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                const char* mean_file, int device, rpde_lnse2d** out) {
  return create_lnse(nx, ny, ra, pr, dt, aspect, bc, mean_file, device, false, out);
}
int rpde_lnse2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                const char* mean_file, int device, rpde_lnse2d** out) {
  return create_lnse(nx, ny, ra, pr, dt, aspect, bc, mean_file, device, true, out);
}
int rpde_nonlin2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                  const char* mean_file, int device, rpde_lnse2d** out) {
  return create_lnse(nx, ny, ra, pr, dt, aspect, bc, mean_file, device, false, out, true);
}
int rpde_nonlin2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                  const char* mean_file, int device, rpde_lnse2d** out) {
  return create_lnse(nx, ny, ra, pr, dt, aspect, bc, mean_file, device, true, out, true);
}
int rpde_lnse2d_update_direct(rpde_lnse2d* h, int nsteps) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(nsteps >= 0, "negative step count"); select_device(h->device); h->e->update_direct(nsteps); })
}
int rpde_lnse2d_history_len(rpde_lnse2d* h, long* n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(n, "null pointer"); *n = (long)h->e->history_len(); })
}
int rpde_lnse2d_clear_history(rpde_lnse2d* h) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->clear_history(); dev_trim(); })
}
int rpde_lnse2d_destroy(rpde_lnse2d* h) {
  RPDE_TRY({ if (h) { select_device(h->device); delete h->e; delete h; dev_trim(); } })
}
int rpde_lnse2d_set_velocity(rpde_lnse2d* h, double amp, double m, double n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->set_velocity(amp, m, n); })
}
int rpde_lnse2d_set_temperature(rpde_lnse2d* h, double amp, double m, double n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->set_temperature(amp, m, n); })
}
int rpde_lnse2d_reset_time(rpde_lnse2d* h) { RPDE_TRY({ RPDE_CHECK_HANDLE(h); h->e->reset_time(); }) }
int rpde_lnse2d_spectral_shape(rpde_lnse2d* h, const char* name, int* rows, int* cols, int* is_complex) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && rows && cols && is_complex, "null pointer");
    int e = 1;
    h->e->spectral_shape(name, rows, cols, &e);
    *is_complex = e == 2;
  })
}
int rpde_lnse2d_set_field(rpde_lnse2d* h, const char* name, int space, const double* data, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && data, "null pointer");
    select_device(h->device);
    if (space == RPDE_PHYSICAL) h->e->set_field_physical(name, data, len);
    else if (space == RPDE_SPECTRAL) h->e->set_field_spectral(name, data, len);
    else fail("space must be RPDE_PHYSICAL or RPDE_SPECTRAL");
  })
}
int rpde_lnse2d_get_field(rpde_lnse2d* h, const char* name, int space, double* data, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && data, "null pointer");
    select_device(h->device);
    if (space == RPDE_PHYSICAL) h->e->get_field_physical(name, data, len);
    else if (space == RPDE_SPECTRAL) h->e->get_field_spectral(name, data, len);
    else fail("space must be RPDE_PHYSICAL or RPDE_SPECTRAL");
  })
}
int rpde_lnse2d_set_mean(rpde_lnse2d* h, const char* name, const double* data, size_t len) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(name && data, "null pointer"); select_device(h->device); h->e->set_mean_physical(name, data, len); })
}
int rpde_lnse2d_get_mean(rpde_lnse2d* h, const char* name, double* data, size_t len) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(name && data, "null pointer"); select_device(h->device); h->e->get_mean_physical(name, data, len); })
}
int rpde_lnse2d_update(rpde_lnse2d* h, int nsteps) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(nsteps >= 0, "negative step count"); select_device(h->device); h->e->update(nsteps); })
}
int rpde_lnse2d_time(rpde_lnse2d* h, double* time) { RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(time, "null pointer"); *time = h->e->time(); }) }
int rpde_lnse2d_dt(rpde_lnse2d* h, double* dt) { RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(dt, "null pointer"); *dt = h->e->dt(); }) }
int rpde_lnse2d_param(rpde_lnse2d* h, const char* key, double* value) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(key && value, "null pointer"); *value = h->e->param(key); })
}
int rpde_lnse2d_exit(rpde_lnse2d* h, int* stop) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(stop, "null pointer"); select_device(h->device); *stop = h->e->exit() ? 1 : 0; })
}
int rpde_lnse2d_div_norm(rpde_lnse2d* h, double* norm) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(norm, "null pointer"); select_device(h->device); *norm = h->e->div_norm(); })
}
int rpde_lnse2d_write(rpde_lnse2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->write(filename); })
}
int rpde_lnse2d_read(rpde_lnse2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->read(filename); })
}

int rpde_lnse2d_update_adjoint(rpde_lnse2d* h, int nsteps) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(nsteps >= 0, "negative step count"); select_device(h->device); h->e->update_adjoint(nsteps); })
}
int rpde_lnse2d_integrate(rpde_lnse2d* h, double max_time, long* timesteps) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); const long n = h->e->integrate(max_time); if (timesteps) *timesteps = n; })
}
int rpde_lnse2d_energy(rpde_lnse2d* h, double beta1, double beta2, const double* target_velx, const double* target_vely,
                       const double* target_temp, size_t len, double* energy) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(energy, "null pointer");
    RPDE_REQUIRE(!target_velx || len == (size_t)h->e->nx() * h->e->ny(), "energy: target arrays are nx*ny doubles");
    select_device(h->device);
    *energy = h->e->energy(beta1, beta2, target_velx, target_vely, target_temp);
  })
}
int rpde_lnse2d_grad_adjoint(rpde_lnse2d* h, double max_time, double save_intervall, double beta1, double beta2, const double* target_velx,
                             const double* target_vely, const double* target_temp, size_t len, const char* filename, double* fun_val,
                             double* grad_velx, double* grad_vely, double* grad_temp, long* timesteps) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(fun_val && grad_velx && grad_vely && grad_temp, "null pointer");
    RPDE_REQUIRE(len == (size_t)h->e->nx() * h->e->ny(), "grad_adjoint: physical arrays are nx*ny doubles");
    select_device(h->device);
    *fun_val = h->e->grad_adjoint(max_time, save_intervall, beta1, beta2, target_velx, target_vely, target_temp, grad_velx, grad_vely, grad_temp, filename, timesteps);
  })
}
int rpde_lnse2d_callback_from_filename(rpde_lnse2d* h, const char* flow_name, const char* info_name, int suppress_io,
                                       double write_flow_intervall) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(flow_name && info_name, "null pointer"); select_device(h->device);
    h->e->callback_from_filename(flow_name, info_name, suppress_io != 0, write_flow_intervall);
  })
}
int rpde_lnse2d_diagnostics(rpde_lnse2d* h, double* out7) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(out7, "null pointer"); select_device(h->device); h->e->diagnostics(out7); })
}
int rpde_lnse2d_grad_fd(rpde_lnse2d* h, double max_time, double beta1, double beta2, const int* points, long npoints, size_t len,
                        const char* filename, double* grad_velx, double* grad_vely, double* grad_temp) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(grad_velx && grad_vely && grad_temp, "null pointer");
    RPDE_REQUIRE(len == (size_t)h->e->nx() * h->e->ny(), "grad_fd: physical arrays are nx*ny doubles");
    RPDE_REQUIRE(!points || npoints >= 0, "grad_fd: negative point count");
    select_device(h->device);
    h->e->grad_fd(max_time, beta1, beta2, points, npoints, grad_velx, grad_vely, grad_temp, filename);
  })
}
int rpde_lnse2d_grad_fd_save(rpde_lnse2d* h, double max_time, double save_intervall, double beta1, double beta2, const int* points,
                             long npoints, size_t len, const char* filename, double* grad_velx, double* grad_vely, double* grad_temp) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(grad_velx && grad_vely && grad_temp, "null pointer");
    RPDE_REQUIRE(len == (size_t)h->e->nx() * h->e->ny(), "grad_fd: physical arrays are nx*ny doubles");
    RPDE_REQUIRE(!points || npoints >= 0, "grad_fd: negative point count");
    select_device(h->device);
    h->e->grad_fd(max_time, beta1, beta2, points, npoints, grad_velx, grad_vely, grad_temp, filename, save_intervall);
  })
}
// functions.rs:30-58
int rpde_l2_norm(size_t len, const double* a1, const double* a2, const double* b1, const double* b2, const double* c1, const double* c2,
                 double beta1, double beta2, double* out) {
  RPDE_TRY({
    RPDE_REQUIRE(a1 && a2 && b1 && b2 && c1 && c2 && out, "null pointer");
    double s = 0.0;
    for (size_t i = 0; i < len; ++i) s += beta1 * a1[i] * a2[i] + beta1 * b1[i] * b2[i] + beta2 * c1[i] * c2[i];
    *out = 0.5 * s;
  })
}
// opt_routines.rs:16-56 (host arrays in, host arrays out, like the reference; the gradients are projected in place)
int rpde_steepest_descent_energy_constrained(size_t len, const double* velx_0, const double* vely_0, const double* temp_0, double* grad_velx,
                                             double* grad_vely, double* grad_temp, double* velx_new, double* vely_new, double* temp_new,
                                             double beta1, double beta2, double alpha) {
  RPDE_TRY({
    RPDE_REQUIRE(velx_0 && vely_0 && temp_0 && grad_velx && grad_vely && grad_temp && velx_new && vely_new && temp_new, "null pointer");
    RPDE_REQUIRE(alpha <= 2.0 * M_PI, "alpha must be less than 2 pi");
    auto l2 = [&](const double* a1, const double* a2, const double* b1, const double* b2, const double* c1, const double* c2) {
      double s = 0.0;
      for (size_t i = 0; i < len; ++i) s += beta1 * a1[i] * a2[i] + beta1 * b1[i] * b2[i] + beta2 * c1[i] * c2[i];
      return 0.5 * s;
    };
    const double n = (double)len;
    const double e0 = l2(velx_0, velx_0, vely_0, vely_0, temp_0, temp_0) / n;
    double eg = l2(grad_velx, velx_0, grad_vely, vely_0, grad_temp, temp_0) / n;
    const double ee = eg / e0;
    for (size_t i = 0; i < len; ++i) { grad_velx[i] -= ee * velx_0[i]; grad_vely[i] -= ee * vely_0[i]; grad_temp[i] -= ee * temp_0[i]; }
    eg = l2(grad_velx, grad_velx, grad_vely, grad_vely, grad_temp, grad_temp) / n;
    const double ee2 = std::sqrt(e0 / eg), ca = std::cos(alpha), sa = std::sin(alpha);
    for (size_t i = 0; i < len; ++i) {
      velx_new[i] = velx_0[i] * ca + grad_velx[i] * ee2 * sa;
      vely_new[i] = vely_0[i] * ca + grad_vely[i] * ee2 * sa;
      temp_new[i] = temp_0[i] * ca + grad_temp[i] * ee2 * sa;
    }
  })
}

int rpde_navier2d_set_velocity(rpde_navier2d* h, double amp, double m, double n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->set_velocity(amp, m, n); })
}
int rpde_navier2d_set_temperature(rpde_navier2d* h, double amp, double m, double n) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->set_temperature(amp, m, n); })
}
int rpde_navier2d_init_random(rpde_navier2d* h, double amp, uint64_t seed) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->init_random(amp, seed); })
}
int rpde_navier2d_reset_time(rpde_navier2d* h) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); h->e->reset_time(); })
}
int rpde_navier2d_spectral_shape(rpde_navier2d* h, const char* name, int* rows, int* cols, int* is_complex) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && rows && cols && is_complex, "null pointer");
    int e = 1;
    h->e->spectral_shape(name, rows, cols, &e);
    *is_complex = e == 2;
  })
}
int rpde_navier2d_set_field(rpde_navier2d* h, const char* name, int space, const double* data, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && data, "null pointer");
    select_device(h->device);
    if (space == RPDE_PHYSICAL) h->e->set_field_physical(name, data, len);
    else if (space == RPDE_SPECTRAL) h->e->set_field_spectral(name, data, len);
    else fail("space must be RPDE_PHYSICAL or RPDE_SPECTRAL");
  })
}
int rpde_navier2d_get_field(rpde_navier2d* h, const char* name, int space, double* data, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(name && data, "null pointer");
    select_device(h->device);
    if (space == RPDE_PHYSICAL) h->e->get_field_physical(name, data, len);
    else if (space == RPDE_SPECTRAL) h->e->get_field_spectral(name, data, len);
    else fail("space must be RPDE_PHYSICAL or RPDE_SPECTRAL");
  })
}
int rpde_navier2d_get_grid(rpde_navier2d* h, int axis, double* x, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(x && (axis == 0 || axis == 1), "bad argument");
    h->e->grid(axis, x, len);
  })
}
int rpde_navier2d_update(rpde_navier2d* h, int nsteps) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->update(nsteps); })
}
int rpde_navier2d_last_update_ms(rpde_navier2d* h, double* ms) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(ms, "null pointer"); *ms = h->e->last_update_ms(); })
}
int rpde_navier2d_profile(rpde_navier2d* h, int nsteps, char* buf, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(buf && len > 0, "null pointer"); select_device(h->device);
    const std::string s = h->e->profile(nsteps);
    RPDE_REQUIRE(s.size() + 1 <= len, "profile buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
  })
}
int rpde_navier2d_trace_launch(rpde_navier2d* h, const char* tag, char* buf, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(buf && len > 0, "null pointer"); select_device(h->device);
    const std::string s = h->e->trace_launch(tag ? tag : "");
    RPDE_REQUIRE(s.size() + 1 <= len, "trace buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
  })
}

int rpde_navier2d_describe_step(rpde_navier2d* h, char* buf, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(buf && len > 0, "null pointer");
    const std::string s = h->e->describe_step();
    RPDE_REQUIRE(s.size() + 1 <= len, "schedule buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
  })
}
int rpde_navier2d_set_timed_tag(rpde_navier2d* h, const char* tag) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(tag, "null pointer"); h->e->set_timed_tag(tag); })
}
int rpde_navier2d_get_timed(rpde_navier2d* h, double* ms_total, long* launches) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(ms_total && launches, "null pointer"); h->e->get_timed(ms_total, launches); })
}
int rpde_navier2d_time(rpde_navier2d* h, double* t) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(t, "null pointer"); *t = h->e->time(); })
}
int rpde_navier2d_dt(rpde_navier2d* h, double* dt) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(dt, "null pointer"); *dt = h->e->dt(); })
}
int rpde_navier2d_param(rpde_navier2d* h, const char* key, double* value) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(key && value, "null pointer"); *value = h->e->param(key); })
}
int rpde_navier2d_exit(rpde_navier2d* h, int* flag) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(flag, "null pointer");
    select_device(h->device);
    *flag = h->e->exit() ? 1 : 0;
  })
}
int rpde_navier2d_div_norm(rpde_navier2d* h, double* value) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(value, "null pointer");
    select_device(h->device);
    *value = h->e->div_norm();
  })
}
int rpde_navier2d_diagnostics(rpde_navier2d* h, double* nu, double* nuvol, double* re) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(nu && nuvol && re, "null pointer");
    select_device(h->device);
    h->e->diagnostics(nu, nuvol, re);
  })
}
int rpde_navier2d_write(rpde_navier2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->write(filename); })
}
int rpde_navier2d_read(rpde_navier2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->read(filename); })
}
int rpde_navier2d_set_write_intervall(rpde_navier2d* h, double dt_save) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); h->e->set_write_intervall(dt_save); })
}
int rpde_navier2d_callback(rpde_navier2d* h) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->callback(); })
}
int rpde_navier2d_callback_from_filename(rpde_navier2d* h, const char* flow_name, const char* info_name, int suppress_io,
                                         double write_flow_intervall) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(flow_name && info_name, "null pointer");
    select_device(h->device);
    h->e->callback_from_filename(flow_name, info_name, suppress_io != 0, write_flow_intervall);
  })
}
int rpde_navier2d_statistics_enable(rpde_navier2d* h, double save_stat, double write_stat) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(save_stat > 0.0 && write_stat > 0.0, "statistics: save_stat and write_stat are positive time intervals");
    select_device(h->device);
    h->e->statistics_enable(save_stat, write_stat);
  })
}
int rpde_navier2d_statistics_attach(rpde_navier2d* h, int on) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    RPDE_REQUIRE(on == 0 || h->e->statistics_enabled(), "statistics_attach: statistics are not enabled");
    h->e->statistics_attach(on != 0);
  })
}
int rpde_navier2d_statistics_update(rpde_navier2d* h) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); select_device(h->device); h->e->statistics_update(); })
}
int rpde_navier2d_statistics_write(rpde_navier2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->statistics_write(filename); })
}
int rpde_navier2d_statistics_read(rpde_navier2d* h, const char* filename) {
  RPDE_TRY({ RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(filename, "null pointer"); select_device(h->device); h->e->statistics_read(filename); })
}
int rpde_navier2d_statistics_get(rpde_navier2d* h, const char* name, double* out, size_t len) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(name && out, "null pointer");
    select_device(h->device);
    h->e->statistics_get(name, out, len);
  })
}
int rpde_navier2d_statistics_scalars(rpde_navier2d* h, double* avg_time, double* tot_time, long long* num_save) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); RPDE_REQUIRE(avg_time && tot_time && num_save, "null pointer");
    h->e->statistics_scalars(avg_time, tot_time, num_save);
  })
}
// h5lite access for hosts without libhdf5 (tests, Python mirror): datasets of rank 1 / 2, f64
int rpde_h5_shape(const char* filename, const char* path, int* rank, uint64_t* dims2) {
  RPDE_TRY({
    RPDE_REQUIRE(filename && path && rank && dims2, "null pointer");
    h5::Reader r(filename);
    const std::vector<uint64_t> d = r.shape(path);
    RPDE_REQUIRE(d.size() <= 2, "rank > 2");
    *rank = (int)d.size();
    for (size_t i = 0; i < d.size(); ++i) dims2[i] = d[i];
  })
}
int rpde_h5_read(const char* filename, const char* path, double* out, size_t len) {
  RPDE_TRY({
    RPDE_REQUIRE(filename && path && out, "null pointer");
    h5::Reader r(filename);
    const h5::Dataset d = r.read(path);
    RPDE_REQUIRE(d.data.size() == len, "h5 read: wrong buffer length for " + std::string(path));
    std::copy(d.data.begin(), d.data.end(), out);
  })
}
int rpde_h5_write(const char* filename, const char* path, int rank, const uint64_t* dims, const double* data) {
  RPDE_TRY({
    RPDE_REQUIRE(filename && path && dims && data && (rank == 1 || rank == 2), "bad argument");
    h5::Dataset d;
    size_t n = 1;
    for (int i = 0; i < rank; ++i) { d.dims.push_back(dims[i]); n *= dims[i]; }
    d.data.assign(data, data + n);
    h5::Tree t;
    t[path] = std::move(d);
    h5::update_file(filename, t);
  })
}
int rpde_h5_list(const char* filename, char* buf, size_t len) {
  RPDE_TRY({
    RPDE_REQUIRE(filename && buf && len > 0, "bad argument");
    h5::Reader r(filename);
    std::string s;
    for (const std::string& p : r.paths()) s += p + "\n";
    RPDE_REQUIRE(s.size() < len, "buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
  })
}

int rpde_navier2d_integrate(rpde_navier2d* h, double max_time, int exit_check_every, long* steps) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h);
    select_device(h->device);
    // src/lib.rs:187-219 without the I/O callback; the NaN guard (which costs two gradients in
    // the reference, every step) is evaluated every `exit_check_every` steps (>= 1)
    const long kMaxTimestep = 10000000;
    const int every = exit_check_every < 1 ? 1 : exit_check_every;
    const double eps_dt = h->e->dt() * 1e-4;
    long n = 0;
    for (;;) {
      h->e->update(1);
      ++n;
      if (h->e->time() + eps_dt >= max_time) break;
      if (n >= kMaxTimestep) break;
      if (n % every == 0 && h->e->exit()) break;
    }
    if (steps) *steps = n;
  })
}

// ------------------------------------------------------------------------------------------
int rpde_space2_create(int kind0, int n0, int kind1, int n1, int device, rpde_space2** out) {
  RPDE_TRY({
    RPDE_REQUIRE(out, "null pointer");
    RPDE_REQUIRE(kind0 >= 0 && kind0 <= 3 && ((kind1 >= 0 && kind1 <= 2) || kind1 == kChebDirichletNeumann), "unknown base kind");
    select_device(device);
    auto* s = new rpde_space2{nullptr, Stream{}, device};
    try {
      s->sp = new Space2Ops(make_base((BaseKind)kind0, n0), make_base((BaseKind)kind1, n1));
    } catch (...) {
      delete s;
      throw;
    }
    *out = s;
  })
}
int rpde_space2_destroy(rpde_space2* s) {
  RPDE_TRY({ if (s) { select_device(s->device); delete s->sp; delete s; dev_trim(); } })
}
static void shape_of(Space2Ops& sp, int which, int* r, int* c, int* e) {
  switch (which) {
    case 0: *r = sp.phys_rows(); *c = sp.phys_cols(); *e = 1; break;
    case 1: *r = sp.spec_rows(); *c = sp.spec_cols(); *e = sp.elem(); break;
    case 2: *r = sp.ortho_rows(); *c = sp.ortho_cols(); *e = sp.elem(); break;
    default: fail("shape selector must be 0 (physical), 1 (spectral) or 2 (orthonormal)");
  }
}
int rpde_space2_shape(rpde_space2* s, int which, int* rows, int* cols, int* is_complex) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s);
    RPDE_REQUIRE(rows && cols && is_complex, "null pointer");
    int e = 1;
    shape_of(*s->sp, which, rows, cols, &e);
    *is_complex = e == 2;
  })
}

}  // extern "C"
namespace {
// host array -> Arr2 of a given shape selector, with length check
Arr2 upload_shape(Space2Ops& sp, int which, const double* h, size_t n, const char* what) {
  int r, c, e;
  shape_of(sp, which, &r, &c, &e);
  RPDE_REQUIRE(h != nullptr, "null pointer");
  RPDE_REQUIRE(n == (size_t)r * c * e, std::string(what) + ": array length does not match the space");
  Arr2 a(r, c, e);
  dev_upload2d(a.p(), a.ld, h, r, (long)c * e);
  return a;
}
Arr2 alloc_shape(Space2Ops& sp, int which, const double* h, size_t n, const char* what) {
  int r, c, e;
  shape_of(sp, which, &r, &c, &e);
  RPDE_REQUIRE(h != nullptr, "null pointer");
  RPDE_REQUIRE(n == (size_t)r * c * e, std::string(what) + ": array length does not match the space");
  return Arr2(r, c, e);
}
void download(const Arr2& a, double* h) { dev_download2d(h, a.p(), a.ld, a.rows, (long)a.cols * a.elem); }
}  // namespace
extern "C" {

int rpde_space2_forward(rpde_space2* s, const double* v, size_t nv, double* vhat, size_t nvhat) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); select_device(s->device);
    Arr2 a = upload_shape(*s->sp, 0, v, nv, "forward input");
    Arr2 b = alloc_shape(*s->sp, 1, vhat, nvhat, "forward output");
    s->sp->forward(a, b, s->st); dev_sync(s->st); download(b, vhat);
  })
}
int rpde_space2_backward(rpde_space2* s, const double* vhat, size_t nvhat, double* v, size_t nv) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); select_device(s->device);
    Arr2 a = upload_shape(*s->sp, 1, vhat, nvhat, "backward input");
    Arr2 b = alloc_shape(*s->sp, 0, v, nv, "backward output");
    s->sp->backward(a, b, s->st); dev_sync(s->st); download(b, v);
  })
}
int rpde_space2_to_ortho(rpde_space2* s, const double* vhat, size_t nvhat, double* out, size_t nout) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); select_device(s->device);
    Arr2 a = upload_shape(*s->sp, 1, vhat, nvhat, "to_ortho input");
    Arr2 b = alloc_shape(*s->sp, 2, out, nout, "to_ortho output");
    s->sp->to_ortho(a, b, s->st); dev_sync(s->st); download(b, out);
  })
}
int rpde_space2_from_ortho(rpde_space2* s, const double* in, size_t nin, double* vhat, size_t nvhat) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); select_device(s->device);
    Arr2 a = upload_shape(*s->sp, 2, in, nin, "from_ortho input");
    Arr2 b = alloc_shape(*s->sp, 1, vhat, nvhat, "from_ortho output");
    s->sp->from_ortho(a, b, s->st); dev_sync(s->st); download(b, vhat);
  })
}
int rpde_space2_gradient(rpde_space2* s, const double* vhat, size_t nvhat, int d0, int d1,
                         double s0, double s1, double* out, size_t nout) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); select_device(s->device);
    RPDE_REQUIRE(d0 >= 0 && d1 >= 0 && d0 <= 4 && d1 <= 4, "derivative order out of range");
    Arr2 a = upload_shape(*s->sp, 1, vhat, nvhat, "gradient input");
    Arr2 b = alloc_shape(*s->sp, 2, out, nout, "gradient output");
    s->sp->gradient(a, d0, d1, s0, s1, b, s->st); dev_sync(s->st); download(b, out);
  })
}

int rpde_hholtz_adi_create(rpde_space2* s, double c0, double c1, rpde_hholtz_adi** out) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); RPDE_REQUIRE(out, "null pointer"); select_device(s->device);
    *out = new rpde_hholtz_adi{new HholtzAdiOp(*s->sp, c0, c1), s};
  })
}
int rpde_hholtz_adi_solve(rpde_hholtz_adi* hs, const double* in, size_t nin, double* out, size_t nout) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(hs); select_device(hs->s->device);
    Arr2 a = upload_shape(*hs->s->sp, 2, in, nin, "HholtzAdi input");
    Arr2 b = alloc_shape(*hs->s->sp, 1, out, nout, "HholtzAdi output");
    hs->op->solve(a, b, hs->s->st); dev_sync(hs->s->st); download(b, out);
  })
}
int rpde_hholtz_adi_destroy(rpde_hholtz_adi* hs) {
  RPDE_TRY({ if (hs) { delete hs->op; delete hs; } })
}
int rpde_poisson_create(rpde_space2* s, double c0, double c1, rpde_poisson** out) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); RPDE_REQUIRE(out, "null pointer"); select_device(s->device);
    *out = new rpde_poisson{new PoissonOp(*s->sp, c0, c1), s};
  })
}
int rpde_poisson_solve(rpde_poisson* ps, const double* in, size_t nin, double* out, size_t nout) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(ps); select_device(ps->s->device);
    Arr2 a = upload_shape(*ps->s->sp, 2, in, nin, "Poisson input");
    Arr2 b = alloc_shape(*ps->s->sp, 1, out, nout, "Poisson output");
    ps->op->solve(a, b, ps->s->st); dev_sync(ps->s->st); download(b, out);
  })
}
static void x_operator_bands(int base_kind, int n, double c0, Bands& ax, Bands& cx) {
  RPDE_REQUIRE(base_kind == RPDE_CHEB_DIRICHLET || base_kind == RPDE_CHEB_NEUMANN, "x spectrum: a composite Chebyshev base with a two-term stencil");
  RPDE_REQUIRE(n >= 4, "x spectrum: at least four points");
  const Base b = make_base((BaseKind)base_kind, n);
  ax = bands_axpy(Bands{Vec(b.m, 0.0), Vec(b.m, 0.0), Vec(b.m, 0.0), Vec(b.m, 0.0)}, c0, hholtz_mat_b(b));
  cx = hholtz_mat_a(b);
}
int rpde_poisson_x_spectrum(int base_kind, int n, double c0, double* lam, size_t m) {
  RPDE_TRY({
    RPDE_REQUIRE(lam && (int)m == n - 2, "x spectrum: m must be n - 2");
    Bands ax, cx;
    x_operator_bands(base_kind, n, c0, ax, cx);
    const Vec l = eigen_spectrum_parity(ax, cx);
    std::copy(l.begin(), l.end(), lam);
  })
}
int rpde_poisson_x_eigenbasis_from_spectrum(int base_kind, int n, double c0, const double* lam, size_t m,
                                            double* lam_refined, double* fwd, double* bwd) {
  RPDE_TRY({
    RPDE_REQUIRE(lam && lam_refined && fwd && bwd && (int)m == n - 2, "eigenbasis from spectrum: m must be n - 2");
    Bands ax, cx;
    x_operator_bands(base_kind, n, c0, ax, cx);
    const EigenX eg = eigenbasis_from_spectrum(ax, cx, Vec(lam, lam + m));
    std::copy(eg.lam.begin(), eg.lam.end(), lam_refined);
    // the layout of PoissonOp::export_eigenbasis: dense m x m over the natural coefficient index
    std::fill(fwd, fwd + m * m, 0.0);
    std::fill(bwd, bwd + m * m, 0.0);
    size_t moff = 0;
    for (int par = 0; par < 2; ++par) {
      const int mb = par ? eg.mo : eg.me, off = par ? eg.me : 0;
      for (int k = 0; k < mb; ++k)
        for (int i = 0; i < mb; ++i) {
          fwd[(size_t)(off + k) * m + (par + 2 * i)] = eg.fwd[moff + (size_t)k * mb + i];
          bwd[(size_t)(par + 2 * i) * m + (off + k)] = eg.bwd[moff + (size_t)i * mb + k];
        }
      moff += (size_t)mb * mb;
    }
  })
}
int rpde_poisson_create_with_spectrum(rpde_space2* s, double c0, double c1, const double* lam, size_t m, rpde_poisson** out) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); RPDE_REQUIRE(out && lam, "null pointer"); select_device(s->device);
    const Vec spectrum(lam, lam + m);
    struct Pending { explicit Pending(const Vec* v) { set_pending_x_spectrum(v); } ~Pending() { set_pending_x_spectrum(nullptr); } } guard(&spectrum);
    *out = new rpde_poisson{new PoissonOp(*s->sp, c0, c1), s};
  })
}
int rpde_hholtz_create(rpde_space2* s, double c0, double c1, rpde_hholtz** out) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(s); RPDE_REQUIRE(out, "null pointer"); select_device(s->device);
    *out = new rpde_hholtz{new TensorHholtzOp(*s->sp, c0, c1), s};
  })
}
int rpde_hholtz_solve(rpde_hholtz* hs, const double* in, size_t nin, double* out, size_t nout) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(hs); select_device(hs->s->device);
    Arr2 a = upload_shape(*hs->s->sp, 2, in, nin, "Hholtz input");
    Arr2 b = alloc_shape(*hs->s->sp, 1, out, nout, "Hholtz output");
    hs->op->solve(a, b, hs->s->st); dev_sync(hs->s->st); download(b, out);
  })
}
int rpde_hholtz_destroy(rpde_hholtz* hs) {
  RPDE_TRY({ if (hs) { delete hs->op; delete hs; } })
}
int rpde_poisson_eigenbasis(rpde_poisson* ps, double* lam, double* fwd, double* bwd, size_t m) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(ps); select_device(ps->s->device);
    RPDE_REQUIRE(lam && fwd && bwd && (int)m == ps->op->me + ps->op->mo, "eigenbasis: m must be the x spectral size");
    ps->op->export_eigenbasis(lam, fwd, bwd);
  })
}
int rpde_navier2d_poisson_eigenbasis(rpde_navier2d* h, double* lam, double* fwd, double* bwd, size_t m) {
  RPDE_TRY({
    RPDE_CHECK_HANDLE(h); select_device(h->device);
    const PoissonOp& po = h->e->poisson();
    RPDE_REQUIRE(lam && fwd && bwd && (int)m == po.me + po.mo, "eigenbasis: m must be the x spectral size (nx - 2)");
    po.export_eigenbasis(lam, fwd, bwd);
  })
}
int rpde_poisson_destroy(rpde_poisson* ps) {
  RPDE_TRY({ if (ps) { delete ps->op; delete ps; } })
}

int rpde_transpose(const double* in, int rows, int cols, int elem, double* out, int device) {
  RPDE_TRY({
    RPDE_REQUIRE(in && out && rows > 0 && cols > 0 && (elem == 1 || elem == 2), "bad argument");
    select_device(device);
    Stream st;
    Arr2 a(rows, cols, elem), b(cols, rows, elem);
    dev_upload2d(a.p(), a.ld, in, rows, (long)cols * elem);
    launch_transpose(a.p(), a.ld, b.p(), b.ld, rows, cols, elem, st);
    dev_sync(st);
    dev_download2d(out, b.p(), b.ld, cols, (long)rows * elem);
  })
}
static void dct_line_entry(int kind, int n, const double* in, int nlines, int deriv, double scale, double* out) {
  RPDE_REQUIRE(in && out && nlines > 0 && kind >= 0 && kind <= 2, "bad argument");
  Stream st;
  const Base b = make_base(kind == 1 ? kChebDirichlet : (kind == 2 ? kChebNeumann : kChebyshev), n);
  AxisTables ax(b);
  Arr2 a(nlines, b.m), v(nlines, n);
  dev_upload2d(a.p(), a.ld, in, nlines, b.m);
  DctLineArgs d{a.p(), a.ld, b.m, v.p(), v.ld, nlines, n - 1, kind == 1 ? 2 : (kind == 2 ? 1 : 0), ax.tw.p, ax.tw2.p, 1.0};
  d.low = kind == 2 ? ax.low.p : nullptr;
  d.deriv = deriv;
  d.dscale = scale;
  RPDE_REQUIRE(ax.fft_n == n - 1 && launch_dct_line(d, st), "whole-line transform kernel: line length not covered");
  dev_sync(st);
  dev_download2d(out, v.p(), v.ld, nlines, n);
}
int rpde_dct_line_backward(int kind, int n, const double* in, int nlines, double* out, int device) {
  RPDE_TRY({ select_device(device); dct_line_entry(kind, n, in, nlines, 0, 1.0, out); })
}
int rpde_dct_line_forward(int n, const double* in, int nlines, int cut, double* out, int device) {
  RPDE_TRY({
    RPDE_REQUIRE(in && out && nlines > 0, "bad argument");
    select_device(device);
    Stream st;
    AxisTables ax(make_base(kChebyshev, n));
    Arr2 v(nlines, n), c(nlines, n);
    dev_upload2d(v.p(), v.ld, in, nlines, n);
    DctLineArgs d{v.p(), v.ld, n, c.p(), c.ld, nlines, n - 1, 0, ax.tw.p, ax.tw2.p, 1.0};
    d.fwd = 1;
    d.cut = cut < 0 ? n : cut;
    RPDE_REQUIRE(ax.fft_n == n - 1 && launch_dct_line(d, st), "whole-line transform kernel: line length not covered");
    dev_sync(st);
    dev_download2d(out, c.p(), c.ld, nlines, n);
  })
}
int rpde_dct_line_gradient(int kind, int n, const double* in, int nlines, double scale, double* out, int device) {
  RPDE_TRY({ select_device(device); dct_line_entry(kind, n, in, nlines, 1, scale, out); })
}
int rpde_conv_line(int n, const double* fx, const double* f0, const double* up, const double* vp, const double* bx,
                   const double* by, int nlines, double dscale, int cut, double* out, int device) {
  RPDE_TRY({
    RPDE_REQUIRE(fx && f0 && up && vp && out && nlines > 0 && n >= 5 && (bx == nullptr) == (by == nullptr), "bad argument");
    select_device(device);
    Stream st;
    AxisTables ax(make_base(kChebDirichlet, n));
    const long ld = pitch(n + 2);
    const size_t sz = (size_t)nlines * ld;
    DBuf dfx(sz), df0(sz), dup(sz), dvp(sz), dbx(bx ? sz : 8), dby(by ? sz : 8), dout(sz);
    dev_upload2d(dfx.p, ld, fx, nlines, n - 2);
    dev_upload2d(df0.p, ld, f0, nlines, n - 2);
    dev_upload2d(dup.p, ld, up, nlines, n);
    dev_upload2d(dvp.p, ld, vp, nlines, n);
    if (bx) { dev_upload2d(dbx.p, ld, bx, nlines, n); dev_upload2d(dby.p, ld, by, nlines, n); }
    const ConvLineArgs c{dfx.p, df0.p, dup.p, dvp.p, bx ? dbx.p : nullptr, by ? dby.p : nullptr, ld, n - 2, dout.p, ld,
                         nlines, n - 1, ax.tw.p, ax.tw2.p, dscale, cut < 0 ? n : cut};
    RPDE_REQUIRE(ax.fft_n == n - 1 && launch_conv_line(c, st), "whole-line convection kernel: line length not covered");
    dev_sync(st);
    dev_download2d(out, dout.p, ld, nlines, n);
  })
}
int rpde_gemm(int M, int N, int K, const double* a, const double* b, int transb, double* c, int device) {
  RPDE_TRY({
    RPDE_REQUIRE(a && b && c && M > 0 && N > 0 && K > 0, "bad argument");
    select_device(device);
    Stream st;
    Arr2 A(M, K), B(transb ? N : K, transb ? K : N), C(M, N);
    dev_upload2d(A.p(), A.ld, a, M, K);
    dev_upload2d(B.p(), B.ld, b, B.rows, B.cols);
    if (transb) launch_gemm_nt(M, N, K, A.p(), A.ld, B.p(), B.ld, C.p(), C.ld, st);
    else launch_gemm_nn(M, N, K, A.p(), A.ld, B.p(), B.ld, C.p(), C.ld, st);
    dev_sync(st);
    dev_download2d(c, C.p(), C.ld, M, N);
  })
}


int rpde_microbench(const char* what, int n, int nlines, int reps, int device, double* ms) {
  RPDE_TRY({
    RPDE_REQUIRE(what && ms && n >= 5 && nlines > 0 && reps > 0, "bad argument");
    select_device(device);
    const std::string w = what;
    Stream st;
#ifndef RPDE_EMU
    // non-line kernels: "gemm_nt" / "gemm_nn" (n x n x n, the Poisson GEMM shapes), "transpose"
    // (n x nlines), "mfma_peak" / "mfma_peak_a" (register-only MFMA loop, accumulators in VGPRs / AGPRs;
    // returns ms, flops = blocks*4*iters*16*2048)
    if (w == "gemm_nt" || w == "gemm_nn" || w == "transpose" || w == "mfma_peak" || w == "mfma_peak_a") {
      const long ld = pitch(n), ld2 = pitch(nlines);
      DBuf A((size_t)std::max(n, nlines) * std::max(ld, ld2)), B((size_t)std::max(n, nlines) * std::max(ld, ld2)),
          Cc((size_t)std::max(n, nlines) * std::max(ld, ld2));
      {
        Vec hbuf(A.n);
        for (size_t i = 0; i < hbuf.size(); ++i) hbuf[i] = std::sin(0.001 * (double)i);
        dev_upload(A.p, hbuf.data(), hbuf.size() * sizeof(double));
        dev_upload(B.p, hbuf.data(), hbuf.size() * sizeof(double));
      }
      auto once = [&]() {
        if (w == "gemm_nt") launch_gemm_nt(n, nlines, n, A.p, ld, B.p, ld, Cc.p, ld2, st);
        else if (w == "gemm_nn") launch_gemm_nn(n, nlines, n, A.p, ld, B.p, ld2, Cc.p, ld2, st);
        else if (w == "transpose") launch_transpose(A.p, ld, Cc.p, ld2, nlines, n, 1, st);
        else launch_mfma_peak(Cc.p, n, nlines, st, w == "mfma_peak_a" ? 1 : 0);   // n = workgroups, nlines = iterations
      };
      once();
      dev_sync(st);
      hipEvent_t e0, e1;
      RPDE_HIP(hipEventCreate(&e0)); RPDE_HIP(hipEventCreate(&e1));
      RPDE_HIP(hipEventRecord(e0, st.s));
      for (int r = 0; r < reps; ++r) once();
      RPDE_HIP(hipEventRecord(e1, st.s));
      RPDE_HIP(hipEventSynchronize(e1));
      float t = 0.f;
      RPDE_HIP(hipEventElapsedTime(&t, e0, e1));
      *ms = t / reps;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      return 0;
    }
#endif
    // the reference's criterion benches of Field2 on cheb_dirichlet(n) x cheb_dirichlet(nlines) (benches/benchmark_transform.rs:
    // field.forward(); benchmark_to_ortho.rs: field.to_ortho(), field.from_ortho(&array)), arrays resident in HBM
    if (w == "forward2d" || w == "backward2d" || w == "to_ortho2d" || w == "from_ortho2d") {
      Space2Ops sp(make_base(kChebDirichlet, n), make_base(kChebDirichlet, nlines));
      Arr2 phys(n, nlines), spec(n - 2, nlines - 2), ortho(n, nlines);
      {
        Vec hbuf((size_t)n * phys.ld);
        for (size_t i = 0; i < hbuf.size(); ++i) hbuf[i] = std::sin(0.001 * (double)i);
        phys.buf.upload(hbuf);
        ortho.buf.upload(hbuf);
        hbuf.resize((size_t)(n - 2) * spec.ld);
        spec.buf.upload(hbuf);
      }
      auto once = [&]() {
        if (w == "forward2d") sp.forward(phys, spec, st);
        else if (w == "backward2d") sp.backward(spec, phys, st);
        else if (w == "to_ortho2d") sp.to_ortho(spec, ortho, st);
        else sp.from_ortho(ortho, spec, st);
      };
      once();
      dev_sync(st);
#ifndef RPDE_EMU
      hipEvent_t e0, e1;
      RPDE_HIP(hipEventCreate(&e0)); RPDE_HIP(hipEventCreate(&e1));
      RPDE_HIP(hipEventRecord(e0, st.s));
      for (int r = 0; r < reps; ++r) once();
      RPDE_HIP(hipEventRecord(e1, st.s));
      RPDE_HIP(hipEventSynchronize(e1));
      float t = 0.f;
      RPDE_HIP(hipEventElapsedTime(&t, e0, e1));
      *ms = t / reps;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#else
      *ms = 0.0;
#endif
      return 0;
    }
    if (w == "dct_line") {   // whole-line backward transform (hdct_line.h)
      AxisTables ax(make_base(kChebDirichlet, n));
      const long ld = pitch(n + 2);
      DBuf in((size_t)nlines * ld), out((size_t)nlines * ld);
      {
        Vec hbuf((size_t)nlines * ld);
        for (size_t i = 0; i < hbuf.size(); ++i) hbuf[i] = std::sin(0.001 * (double)i);
        in.upload(hbuf);
      }
      DctLineArgs d{in.p, ld, n - 2, out.p, ld, nlines, n - 1, 2, ax.tw.p, ax.tw2.p, 1.0};
      RPDE_REQUIRE(launch_dct_line(d, st), "dct_line: shape not covered");
      dev_sync(st);
#ifndef RPDE_EMU
      hipEvent_t e0, e1;
      RPDE_HIP(hipEventCreate(&e0)); RPDE_HIP(hipEventCreate(&e1));
      RPDE_HIP(hipEventRecord(e0, st.s));
      for (int r = 0; r < reps; ++r) launch_dct_line(d, st);
      RPDE_HIP(hipEventRecord(e1, st.s));
      RPDE_HIP(hipEventSynchronize(e1));
      float t = 0.f;
      RPDE_HIP(hipEventElapsedTime(&t, e0, e1));
      *ms = t / reps;
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#else
      *ms = 0.0;
#endif
      return 0;
    }
    const bool fourier = w == "rfft";
    AxisTables ax(make_base(fourier ? kFourierR2c : kChebDirichlet, n));
    const Base& b = ax.base;
    Bands mtx = bands_axpy(hholtz_mat_a(make_base(kChebDirichlet, fourier ? 9 : n)), -1e-6,
                           hholtz_mat_b(make_base(kChebDirichlet, fourier ? 9 : n)));
    fdma_sweep(mtx);
    FdmaDev fd = upload_fdma(fdma_tables(mtx), ax.slot_len);
    const long ld = pitch(n + 2);
    DBuf in((size_t)nlines * ld), out((size_t)nlines * ld);
    {
      Vec hbuf((size_t)nlines * ld);
      for (size_t i = 0; i < hbuf.size(); ++i) hbuf[i] = std::sin(0.001 * (double)i);
      in.upload(hbuf);
    }
    const int nslots = 4;
    ProgramBuilder pb(nslots, ax.slot_len, nlines, 1);
    pb.set_fft(ax);
    const int ai = pb.arr(in.p, ld), ao = pb.arr(out.p, ld);
    const int m = b.m;
    if (w == "copy") { pb.load(0, ai, n); }
    else if (w == "sten") { pb.load(0, ai, m); pb.to_ortho(0, ax); }
    else if (w == "mv3") { pb.load(0, ai, n); pb.pinv_matvec(0, ax); }
    else if (w == "cdiff") { pb.load(0, ai, n); pb.cdiff(0, 0, n, 1.0); }
    else if (w == "fromortho") { pb.load(0, ai, n); pb.from_ortho(0, ax); }
    else if (w == "fdma") { pb.load(0, ai, m); pb.fdma_solve(0, m, fd); }
    else if (w == "dct") { pb.load(0, ai, n); pb.dct(0, n, ax.bwd_pre.p, nullptr); }
    else if (w == "dct0") { pb.load(0, ai, n); pb.dct(0, n, nullptr, nullptr); }   // no scaling tables
    else if (w == "dct2") { pb.load(0, ai, n); pb.dct(0, n, ax.bwd_pre.p, nullptr); pb.dct(0, n, nullptr, ax.fwd_post.p); }
    else if (w == "rfft") { pb.load(0, ai, n); pb.rfft_f(0, n); pb.rfft_b(0, n); }
    else fail("unknown microbench \"" + w + "\"");
    pb.store(0, ao, fourier ? n : (w == "sten" ? n : m));
    pb.run(st);  // warm-up
    dev_sync(st);
#ifndef RPDE_EMU
    hipEvent_t e0, e1;
    RPDE_HIP(hipEventCreate(&e0)); RPDE_HIP(hipEventCreate(&e1));
    RPDE_HIP(hipEventRecord(e0, st.s));
    for (int r = 0; r < reps; ++r) pb.run(st);
    RPDE_HIP(hipEventRecord(e1, st.s));
    RPDE_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    RPDE_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = t / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#else
    for (int r = 0; r < reps; ++r) pb.run(st);
    *ms = 0.0;
#endif
  })
}

}  // extern "C"
