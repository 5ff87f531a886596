/* rustpde_hip.h -- C ABI of the MI355X-native Navier2D time-step engine (librustpde_hip.so).
 *
 * This is the drop-in boundary for the hot path of preiter93/rustpde-mpi: a Rust host keeps
 * `Navier2D::new_confined / new_periodic`, `integrate()` and its HDF5 callbacks and forwards
 * the per-step work to these symbols (binding sketch in INTEGRATION.md).  The reference has no
 * FFI of its own (no `extern "C"` anywhere in src/); each entry point below names the Rust item
 * it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero code on failure; the message of the
 *     last failure on the calling thread is returned by rpde_last_error().  Nothing unwinds
 *     across the boundary.
 *   - handles are opaque; one host thread per handle; each handle owns one HIP stream.
 *   - arrays are caller-owned, row-major f64; complex spectral arrays are interleaved (re, im)
 *     pairs, bit-compatible with num_complex::Complex<f64>.  Host buffers are only touched
 *     during the call.
 *   - physical arrays have shape (nx, ny); spectral arrays have the shape of the reference's
 *     `vhat` of that field (query with rpde_navier2d_spectral_shape).
 */
#ifndef RUSTPDE_HIP_H
#define RUSTPDE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rpde_navier2d rpde_navier2d;   /* Navier2D<f64|Complex<f64>, Space2>  src/navier_stokes/navier.rs:49-89 */
typedef struct rpde_space2 rpde_space2;       /* funspace Space2<B0,B1>              src/bases.rs:11-19, src/field.rs:59-72 */
typedef struct rpde_hholtz_adi rpde_hholtz_adi; /* HholtzAdi<f64,2>                  src/solver/hholtz_adi.rs:31-39 */
typedef struct rpde_hholtz rpde_hholtz;       /* Hholtz<f64,2>                       src/solver/hholtz.rs:29-37 */
typedef struct rpde_lnse2d rpde_lnse2d;       /* Navier2DLnse<f64|Complex<f64>, Space2>     src/navier_stokes_lnse/lnse.rs:24-63 */
typedef struct rpde_adjoint2d rpde_adjoint2d; /* Navier2DAdjoint<f64|Complex<f64>, Space2>  src/navier_stokes/steady_adjoint.rs:67-113 */
typedef struct rpde_poisson rpde_poisson;     /* Poisson<f64,2>                      src/solver/poisson.rs:33-40 */

/* base kinds (funspace BaseKind, used at src/field.rs:172-179) */
enum { RPDE_CHEBYSHEV = 0, RPDE_CHEB_DIRICHLET = 1, RPDE_CHEB_NEUMANN = 2, RPDE_FOURIER_R2C = 3,
       RPDE_CHEB_DIRICHLET_NEUMANN = 4 };   /* axis 1 only: the temperature of bc = "hc", src/navier_stokes/navier.rs:245-248 */
enum { RPDE_PHYSICAL = 0, RPDE_SPECTRAL = 1 };

const char* rpde_last_error(void);
const char* rpde_version(void);
/* 1 when the library was built for the GPU (HIP, gfx950); the test-only host emulation build reports 0 */
int rpde_is_device_build(void);
int rpde_device_count(int* count);
/* Device memory bookkeeping (no counterpart in the reference: ndarray owns host memory).  The library takes its HBM in
 * slabs per DEVICE (csrc/platform.h ArenaT): a handle created with `device` = d only ever gets memory of device d, also when
 * one process drives several GPUs.  rpde_device_memory: bytes held in slabs / bytes in live buffers on `device` (-1: all).
 * rpde_device_trim: slabs without a live buffer go back to the driver (the destroy functions call it; a host that wants the
 * memory back for another library at another time calls it itself).  rpde_arena_check: with RPDE_ARENA_GUARD=1 in the
 * environment every buffer is followed by a guard granule; *violations = buffers whose guard was overwritten since the last
 * call (0 without guards).  rpde_arena_selftest: the keying logic on a host backend with two pretended devices (runs
 * without a GPU); 0 = pass. */
int rpde_device_memory(int device, size_t* slab_bytes, size_t* used_bytes);
int rpde_device_trim(int device, size_t* released_bytes);
int rpde_arena_check(long* violations);
int rpde_arena_selftest(void);

/* ---- engine level: what `impl Integrate for Navier2D` does ----------------------------------- */
/* Navier2D::new_confined(nx, ny, ra, pr, dt, aspect, bc)     src/navier_stokes/navier.rs:215-308 */
int rpde_navier2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect,
                                  const char* bc, int device, rpde_navier2d** out);
/* Navier2D::new_periodic(nx, ny, ra, pr, dt, aspect, bc)     src/navier_stokes/navier.rs:336-428 */
int rpde_navier2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect,
                                  const char* bc, int device, rpde_navier2d** out);
/* The same constructor with the x eigenvalues of the Poisson solver supplied by the host (m = nx - 2 doubles in the order
 * rpde_poisson_x_spectrum returns them): the eigenbasis is then built WITHOUT LAPACK and bit-reproducibly
 * (rpde_poisson_x_eigenbasis_from_spectrum) instead of by dgeev (FdmaTensor::from_matrix, src/solver/fdma_tensor.rs:123-127,
 * src/solver/utils.rs:67-99).  Two uses: a host that caches the spectrum skips the O(n^3) setup of every later run, and a
 * checker on another machine can run the reference algorithm on EXACTLY the engine's setup data -- dgeev's output is not
 * reproducible across thread counts or CPU models and Poisson::new's -1e-10 shift (poisson.rs:84-87) amplifies the
 * difference by 1e10 (tests/golden/make_shared_basis_golden.py). */
int rpde_navier2d_create_confined_with_spectrum(int nx, int ny, double ra, double pr, double dt, double aspect,
                                                const char* bc, int device, const double* lam, size_t m,
                                                rpde_navier2d** out);
/* Pencil-sharded engine = Navier2DMpi (src/navier_stokes_mpi/navier.rs:216-336, 364-486): rank r of
 * nranks owns a slab of every field; layout changes call `alltoallv` (the role of funspace's
 * Decomp2d::transpose_x_to_y / transpose_y_to_x = MPI_Alltoallv, src/field_mpi.rs:456-477).
 * The callback gets buffers that live where the engine's arrays live (HBM); segment q of `send`
 * (sendcounts[q] doubles) goes to rank q, segment s of `recv` arrives from rank s; it must return 0
 * once the data has landed.  rustpde_mpi_amd/dist.py implements it with torch.distributed (RCCL). */
typedef int (*rpde_alltoallv_fn)(void* user, const double* send, const int64_t* sendcounts,
                                 double* recv, const int64_t* recvcounts);
int rpde_navier2d_create_sharded(int periodic, int nx, int ny, double ra, double pr, double dt,
                                 double aspect, const char* bc, int device, int rank, int nranks,
                                 rpde_alltoallv_fn alltoallv, void* user, rpde_navier2d** out);
/* The same sharded engine with the NATIVE transport: every layout change is a grouped
 * ncclSend/ncclRecv all-to-all on the engine's own HIP stream (RCCL over xGMI; no host round trip,
 * no callback).  Rank 0 calls rpde_rccl_unique_id and distributes the 128 bytes to the other
 * ranks by any host-side means (MPI_Bcast in a Rust/MPI host, torch.distributed in bench.py);
 * rpde_navier2d_create_sharded_rccl is collective (ncclCommInitRank) and binds the communicator
 * to `device`.  One process per GPU; an id serves exactly one communicator (take a fresh one per
 * engine). */
int rpde_rccl_unique_id(char* id128);
/* transport self-test: one all-to-all of device buffers on a fresh communicator (collective) */
int rpde_rccl_alltoallv_once(const char* id128, int rank, int nranks, int device, const double* send,
                             const int64_t* sendcounts, double* recv, const int64_t* recvcounts);
int rpde_navier2d_create_sharded_rccl(int periodic, int nx, int ny, double ra, double pr, double dt,
                                      double aspect, const char* bc, int device, int rank, int nranks,
                                      const char* id128, rpde_navier2d** out);
/* bytes this rank sends per time step through `alltoallv`, and the number of exchanges per step */
int rpde_navier2d_comm_stats(rpde_navier2d* h, double* bytes_per_step, int* exchanges_per_step);
int rpde_navier2d_destroy(rpde_navier2d* h);
/* set_velocity / set_temperature / init_random / reset_time  src/navier_stokes/navier.rs:161-187 */
int rpde_navier2d_set_velocity(rpde_navier2d* h, double amp, double m, double n);
int rpde_navier2d_set_temperature(rpde_navier2d* h, double amp, double m, double n);
int rpde_navier2d_init_random(rpde_navier2d* h, double amp, uint64_t seed);
int rpde_navier2d_reset_time(rpde_navier2d* h);
/* field access: name in {"velx","vely","temp","pres","pseu"} = the public Field2 members
 * (navier.rs:52-62); space RPDE_PHYSICAL = `.v` (after backward()), RPDE_SPECTRAL = `.vhat`   */
int rpde_navier2d_spectral_shape(rpde_navier2d* h, const char* name, int* rows, int* cols, int* is_complex);
int rpde_navier2d_set_field(rpde_navier2d* h, const char* name, int space, const double* data, size_t len);
int rpde_navier2d_get_field(rpde_navier2d* h, const char* name, int space, double* data, size_t len);
int rpde_navier2d_get_grid(rpde_navier2d* h, int axis, double* x, size_t len);   /* Field2::x, field.rs:69 */
/* n x Integrate::update()                                    src/navier_stokes/navier.rs:438-466 */
int rpde_navier2d_update(rpde_navier2d* h, int nsteps);
/* device time of the last rpde_navier2d_update call, HIP events on the engine's stream */
int rpde_navier2d_last_update_ms(rpde_navier2d* h, double* ms);
/* measurement hooks (bench.py): per-launch HIP-event profile of `nsteps` steps as a text table   *
 * "tag\tlaunches\tms_total\talgorithmic_bytes_per_launch\tflops_per_launch\n"; and live timing of *
 * the launches whose tag contains `tag` inside rpde_navier2d_update (pass "" to switch it off)  */
int rpde_navier2d_profile(rpde_navier2d* h, int nsteps, char* buf, size_t len);
int rpde_navier2d_set_timed_tag(rpde_navier2d* h, const char* tag);
/* the launches of one step in issue order, "tag\talgorithmic_bytes\tflops\tkernels\tkind\n" per launch (used *
 * to attribute rocprofv3 per-dispatch counters to the step's kernels, tools/pmc_traffic.py; kind = "line      *
 * program", "whole-line transform" ...: which form of a stage this engine chose at construction)            */
int rpde_navier2d_describe_step(rpde_navier2d* h, char* buf, size_t len);
int rpde_navier2d_get_timed(rpde_navier2d* h, double* ms_total, long* launches);
/* diagnostics (tools/trace_ops.py): runs one step with the first line program whose tag contains `tag`     *
 * instrumented: thread 0 of every workgroup records the shader clock when it reaches an op and when it    *
 * leaves a barrier.  "tag\tworkgroups\tspan_ms\tmarks\n", then one row per mark in program order,          *
 * "id\tname\tmean\tp10\tmedian\tp90\n" = clocks since the previous mark over the workgroups (id >= 0: op  *
 * id starts, -1: a barrier inside the op, first row (-2) = the whole program).  The traced step is a real *
 * time step: the fields and the time advance by one dt.  The host-emulation build of the library (tests)  *
 * returns the header line with zero workgroups and no rows.                                              */
int rpde_navier2d_trace_launch(rpde_navier2d* h, const char* tag, char* buf, size_t len);
/* Integrate::get_time / get_dt                               src/navier_stokes/navier.rs:468-474 */
int rpde_navier2d_time(rpde_navier2d* h, double* t);
int rpde_navier2d_dt(rpde_navier2d* h, double* dt);
/* params map ("ra","pr","nu","ka")                           src/navier_stokes/navier.rs:229-233 */
int rpde_navier2d_param(rpde_navier2d* h, const char* key, double* value);
/* Integrate::exit(): 1 when ||div|| is NaN                   src/navier_stokes/navier.rs:482-489 */
int rpde_navier2d_exit(rpde_navier2d* h, int* flag);
/* DivNorm::div_norm                                          src/navier_stokes/navier_eq.rs:33-51 */
int rpde_navier2d_div_norm(rpde_navier2d* h, double* value);
/* eval_nu / eval_nuvol / eval_re (callback diagnostics)     src/navier_stokes/functions.rs:146-233 */
int rpde_navier2d_diagnostics(rpde_navier2d* h, double* nu, double* nuvol, double* re);
/* Snapshots in the reference's HDF5 layout (Navier2D::write / read, src/navier_stokes/navier_io.rs:21-62;  *
 * per field src/field/io.rs:74-110): groups ux, uy, temp, pres, tempbc with x, dx, y, dy,    *
 * v, vhat (vhat_re + vhat_im when complex) and root scalars time, ra, pr, nu, ka.  read restores vhat of     *
 * ux, uy, temp, pres and `time`; another resolution is truncated / zero-padded in spectral space           *
 * (field/io.rs:151-176).  Files are classic-format HDF5 (what libhdf5 writes by default) produced and        *
 * parsed by the library itself -- a Rust host that links libhdf5 can keep its own writer instead.           */
int rpde_navier2d_write(rpde_navier2d* h, const char* filename);
int rpde_navier2d_read(rpde_navier2d* h, const char* filename);
/* Integrate::callback (navier.rs:476-480) = callback_from_filename("data/flow{time:0>8.2}.h5",             *
 * "data/info.txt", false, write_intervall); callback_from_filename: navier_io.rs:84-149 (snapshot on its   *
 * interval -- pass a negative interval for `None` = always --, then "time |div| Nu Nuv Re" on stdout and    *
 * "time nu nuv re" appended to info_name unless suppress_io)                                               */
int rpde_navier2d_set_write_intervall(rpde_navier2d* h, double dt_save);
int rpde_navier2d_callback(rpde_navier2d* h);
int rpde_navier2d_callback_from_filename(rpde_navier2d* h, const char* flow_name, const char* info_name,
                                         int suppress_io, double write_flow_intervall);
/* Statistics (src/navier_stokes/statistics.rs:11-108; `navier.statistics = Some(Statistics::new(&navier, save_stat,   *
 * write_stat))`, navier.rs:88): four fields of the orthonormal `field` space kept on the device -- `temp` = running  *
 * mean of temp.to_ortho(), `ux` / `uy` = the last velx / vely .to_ortho() (the reference assigns, statistics.rs:98-99), *
 * `nusselt` = the Nusselt field of the last snapshot (statistics.rs:248-271) -- plus avg_time, tot_time, num_save.     *
 * Once enabled, rpde_navier2d_callback* updates them on `save_stat` and writes data/statistics.h5 on `write_stat`     *
 * (navier_io.rs:105-121).  _write: groups temp, ux, uy, nusselt (x, dx, y, dy, v, vhat) + tot_time, avg_time,          *
 * num_save (unsigned 64-bit, a Rust usize) + ra, pr, nu, ka (statistics.rs:142-161); _read: statistics.rs:116-130.    *
 * _get: name in {temp, ux, uy, nusselt}, coefficients of the `field` space (nx x ny doubles, or (nx/2+1) x ny          *
 * complex interleaved when periodic).                                                                                 */
int rpde_navier2d_statistics_enable(rpde_navier2d* h, double save_stat, double write_stat);
/* `navier.statistics = Some(stats)` (on != 0) / `= None` (on == 0): the callback acts on attached statistics only     *
 * (navier_io.rs:105 `if let Some(..)`); detaching keeps the accumulated fields.  _enable leaves them attached.        */
int rpde_navier2d_statistics_attach(rpde_navier2d* h, int on);
int rpde_navier2d_statistics_update(rpde_navier2d* h);
int rpde_navier2d_statistics_write(rpde_navier2d* h, const char* filename);
int rpde_navier2d_statistics_read(rpde_navier2d* h, const char* filename);
int rpde_navier2d_statistics_get(rpde_navier2d* h, const char* name, double* out, size_t len);
int rpde_navier2d_statistics_scalars(rpde_navier2d* h, double* avg_time, double* tot_time, long long* num_save);
/* the same HDF5 subset for hosts without libhdf5: contiguous f64 datasets of rank 1 or 2, one group level  *
 * (src/io/read_write_hdf5.rs:38-188: read_from_hdf5 / write_to_hdf5 "create or append, overwrite")          */
int rpde_h5_shape(const char* filename, const char* path, int* rank, uint64_t* dims2);
int rpde_h5_read(const char* filename, const char* path, double* out, size_t len);
int rpde_h5_write(const char* filename, const char* path, int rank, const uint64_t* dims, const double* data);
int rpde_h5_list(const char* filename, char* buf, size_t len);   /* newline-separated dataset paths */
/* integrate(&mut pde, max_time, None) without callbacks      src/lib.rs:187-219 ; returns steps taken */
int rpde_navier2d_integrate(rpde_navier2d* h, double max_time, int exit_check_every, long* steps);

/* ---- adjoint descent to steady states: what `impl Integrate for Navier2DAdjoint` does ------------------------------ *
 * (first slice.)  Fields stay in HBM; one update = a forward step with DT_NAVIER for the residual,  *
 * three tensor Helmholtz solves for its norm, an explicit adjoint step.  bc = "rbc" only (see csrc/adjoint.h).             */
/* Navier2DAdjoint::new_confined(nx, ny, ra, pr, dt, aspect, bc)   src/navier_stokes/steady_adjoint.rs:215-370 */
int rpde_adjoint2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect,
                                   const char* bc, int device, rpde_adjoint2d** out);
/* Navier2DAdjoint::new_periodic(nx, ny, ra, pr, dt, aspect, bc)   src/navier_stokes/steady_adjoint.rs:372-531 */
int rpde_adjoint2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect,
                                   const char* bc, int device, rpde_adjoint2d** out);
int rpde_adjoint2d_destroy(rpde_adjoint2d* h);
int rpde_adjoint2d_set_velocity(rpde_adjoint2d* h, double amp, double m, double n);      /* steady_adjoint.rs:183-186 */
int rpde_adjoint2d_set_temperature(rpde_adjoint2d* h, double amp, double m, double n);   /* steady_adjoint.rs:190-192 */
int rpde_adjoint2d_reset_time(rpde_adjoint2d* h);                                        /* steady_adjoint.rs:207-209 */
/* name in {"velx","vely","temp","pres","pseu","velx_adj","vely_adj","temp_adj","pres_adj","tempbc"} = the public Field2
 * members (steady_adjoint.rs:69-91); "tempbc" is read-only */
int rpde_adjoint2d_spectral_shape(rpde_adjoint2d* h, const char* name, int* rows, int* cols, int* is_complex);
int rpde_adjoint2d_set_field(rpde_adjoint2d* h, const char* name, int space, const double* data, size_t len);
int rpde_adjoint2d_get_field(rpde_adjoint2d* h, const char* name, int space, double* data, size_t len);
/* n x Integrate::update()                                         src/navier_stokes/steady_adjoint.rs:541-608 */
int rpde_adjoint2d_update(rpde_adjoint2d* h, int nsteps);
int rpde_adjoint2d_time(rpde_adjoint2d* h, double* time);          /* Integrate::get_time */
int rpde_adjoint2d_dt(rpde_adjoint2d* h, double* dt);              /* Integrate::get_dt */
int rpde_adjoint2d_param(rpde_adjoint2d* h, const char* key, double* value);   /* params: "ra" "pr" "nu" "ka" */
/* Integrate::exit(): NaN divergence, or (|velx_adj| + |vely_adj| + |temp_adj|) / 3 < RES_TOL = 1e-7 ("Steady state
 * converged!")                                                    src/navier_stokes/steady_adjoint.rs:624-638 */
int rpde_adjoint2d_exit(rpde_adjoint2d* h, int* stop);
/* DivNorm::div_norm / DivNorm::norm_residual -> [|velx_adj|, |vely_adj|, |temp_adj|]   steady_adjoint_eq.rs:36-50 */
int rpde_adjoint2d_div_norm(rpde_adjoint2d* h, double* norm);
int rpde_adjoint2d_norm_residual(rpde_adjoint2d* h, double* res3);
/* Navier2DAdjoint::write / read: ux uy temp pres tempbc + time + params in the reference's HDF5 layout; read takes ux, uy,
 * temp and time, also from a snapshot written by Navier2D (the restart.h5 of examples/navier_rbc_steady.rs) and at another
 * resolution                                                        src/navier_stokes/steady_adjoint_io.rs:22-33, 48-71 */
int rpde_adjoint2d_write(rpde_adjoint2d* h, const char* filename);
int rpde_adjoint2d_read(rpde_adjoint2d* h, const char* filename);

/* ---- linearised Navier-Stokes about mean fields: what `impl Integrate for Navier2DLnse` does -------------------- *
 * (second slice: the forward LNSE step; the adjoint-gradient drivers of src/navier_stokes_lnse are   *
 * not built.)  bc = "rbc".                                                                                                   */
/* Navier2DLnse::new_confined / new_periodic(nx, ny, ra, pr, dt, aspect, bc)     src/navier_stokes_lnse/lnse.rs:98-176, 196-253
 * mean_file: the snapshot MeanFields::read_from_* takes the mean flow from ("ux/v", "uy/v", "temp/v" + "tempbc/v"; NULL = the
 * reference's "mean.h5"); if it does not exist: the boundary condition's default mean (no flow, conduction profile),
 * src/navier_stokes_lnse/meanfield.rs:28-49, 92-127.  (The reference also writes "mean_field.h5" from its constructor; a
 * host that wants it calls rpde_lnse2d_get_mean.) */
int rpde_lnse2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                const char* mean_file, int device, rpde_lnse2d** out);
int rpde_lnse2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                const char* mean_file, int device, rpde_lnse2d** out);
int rpde_lnse2d_destroy(rpde_lnse2d* h);
int rpde_lnse2d_set_velocity(rpde_lnse2d* h, double amp, double m, double n);       /* functions.rs:85-126 on velx / vely */
int rpde_lnse2d_set_temperature(rpde_lnse2d* h, double amp, double m, double n);
int rpde_lnse2d_reset_time(rpde_lnse2d* h);
/* name in {"velx","vely","temp","pres","pseu","tempbc"} (lnse.rs:26-37) */
int rpde_lnse2d_spectral_shape(rpde_lnse2d* h, const char* name, int* rows, int* cols, int* is_complex);
int rpde_lnse2d_set_field(rpde_lnse2d* h, const char* name, int space, const double* data, size_t len);
int rpde_lnse2d_get_field(rpde_lnse2d* h, const char* name, int space, double* data, size_t len);
/* MeanFields: name in {"velx","vely","temp"}, physical (nx x ny) arrays; set = assign + forward (meanfield.rs:237-259) */
int rpde_lnse2d_set_mean(rpde_lnse2d* h, const char* name, const double* data, size_t len);
int rpde_lnse2d_get_mean(rpde_lnse2d* h, const char* name, double* data, size_t len);
/* n x Integrate::update()                                         src/navier_stokes_lnse/lnse.rs:263-288 */
int rpde_lnse2d_update(rpde_lnse2d* h, int nsteps);
int rpde_lnse2d_time(rpde_lnse2d* h, double* time);
int rpde_lnse2d_dt(rpde_lnse2d* h, double* dt);
int rpde_lnse2d_param(rpde_lnse2d* h, const char* key, double* value);
int rpde_lnse2d_exit(rpde_lnse2d* h, int* stop);                   /* NaN divergence, lnse.rs:305-313 */
int rpde_lnse2d_div_norm(rpde_lnse2d* h, double* norm);            /* lnse_eq.rs:36-41 */
int rpde_lnse2d_write(rpde_lnse2d* h, const char* filename);       /* the Field2 snapshot layout (ux uy temp pres tempbc + time + params) */
int rpde_lnse2d_read(rpde_lnse2d* h, const char* filename);
/* Navier2DNonLin::new_confined / new_periodic                      src/navier_stokes_lnse/nonlin.rs:84-262: the non-linear
   equations for the deviation from the mean fields (nonlin_eq.rs), same fields, same handle type -- every rpde_lnse2d_* entry
   serves both solvers, the constructor decides the equations.  rpde_lnse2d_write adds the groups ux_base, uy_base, temp_base
   (nonlin_io.rs:44-66). */
int rpde_nonlin2d_create_confined(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                  const char* mean_file, int device, rpde_lnse2d** out);
int rpde_nonlin2d_create_periodic(int nx, int ny, double ra, double pr, double dt, double aspect, const char* bc,
                                  const char* mean_file, int device, rpde_lnse2d** out);
/* n x update_direct(): Navier2DLnse: == update (lnse_adj_grad.rs:43-68); Navier2DNonLin: update + one entry of field_history per
   step (nonlin_adj_grad.rs:43-81; the history lives in HBM: three spectral arrays per step).  rpde_lnse2d_update_adjoint on a
   Navier2DNonLin removes the LAST entry per step (:190-193) and fails on an empty history. */
int rpde_lnse2d_update_direct(rpde_lnse2d* h, int nsteps);
int rpde_lnse2d_history_len(rpde_lnse2d* h, long* n);
int rpde_lnse2d_clear_history(rpde_lnse2d* h);
/* ---- adjoint-based sensitivity of the final energy (src/navier_stokes_lnse/lnse_adj_grad.rs, lnse_adj_eq.rs) ---- */
/* n x Navier2DLnse::update_adjoint()                               lnse_adj_grad.rs:71-99 (update_direct == rpde_lnse2d_update, :43-68) */
int rpde_lnse2d_update_adjoint(rpde_lnse2d* h, int nsteps);
/* rustpde::integrate(pde, max_time, None)                          src/lib.rs:187-219; timesteps may be null */
int rpde_lnse2d_integrate(rpde_lnse2d* h, double max_time, long* timesteps);
/* functions::energy: 0.5 sum(b1 u^2 + b1 v^2 + b2 T^2) of the physical fields (functions.rs:11-58); with a target (three physical
   nx*ny arrays, or all NULL) of the fields minus the target (lnse_adj_grad.rs:141-155) */
int rpde_lnse2d_energy(rpde_lnse2d* h, double beta1, double beta2, const double* target_velx, const double* target_vely,
                       const double* target_temp, size_t len, double* energy);
/* Navier2DLnse::grad_adjoint(max_time, None, beta1, beta2, target) lnse_adj_grad.rs:105-202: forward loop, energy (-> fun_val),
   adjoint loop, gradient = -(adjoint fields).  grad_*: physical nx*ny arrays (the `.v` of the returned Field2s; `.vhat` =
   Space2 forward of them).  filename: where the reference writes "data/grad_adjoint.h5" (groups ux, uy, temp), NULL = no file.
   The engine is left holding the adjoint fields, like the reference.  timesteps may be NULL.  save_intervall > 0: the snapshots
   "data/flow{time:0>8.2}.h5" / "data/adjoint{time:0>8.2}.h5" and info files of the two loops (:122-130, :176-181), <= 0: None. */
int rpde_lnse2d_grad_adjoint(rpde_lnse2d* h, double max_time, double save_intervall, double beta1, double beta2, const double* target_velx,
                             const double* target_vely, const double* target_temp, size_t len, const char* filename, double* fun_val,
                             double* grad_velx, double* grad_vely, double* grad_temp, long* timesteps);
/* callback_from_filename(flow_name, info_name, suppress_io, write_flow_intervall)   lnse_io.rs:73-126 / nonlin_io.rs:72-142;
   write_flow_intervall < 0 = None (OUTPUT_INTERVALL = 1).  rpde_lnse2d_diagnostics: out7 = |div|, Nu, Nuv, Re (Navier2DNonLin::
   eval_nu / eval_nuvol / eval_re, nonlin_io.rs:145-198; NaN for Navier2DLnse), <u^2>, <v^2>, <T^2> (weighted averages) */
int rpde_lnse2d_callback_from_filename(rpde_lnse2d* h, const char* flow_name, const char* info_name, int suppress_io,
                                       double write_flow_intervall);
int rpde_lnse2d_diagnostics(rpde_lnse2d* h, double* out7);
/* Navier2DLnse::grad_fd(max_time, None, beta1, beta2)              lnse_fd_grad.rs:31-157: one integration per perturbed grid point,
   eps = 1e-5 (a test device in the reference too).  points: npoints triples (field 0 velx / 1 vely / 2 temp, i, j), NULL = all */
int rpde_lnse2d_grad_fd(rpde_lnse2d* h, double max_time, double beta1, double beta2, const int* points, long npoints, size_t len,
                        const char* filename, double* grad_velx, double* grad_vely, double* grad_temp);
/* the same with the reference's `save_intervall: Option<f64>` (lnse_fd_grad.rs:35, 54): > 0 = Some: the BASE run calls
   Integrate::callback on the interval (data/flow{time:0>8.2}.h5, data/info.txt, lnse.rs:298-302); <= 0 = None */
int rpde_lnse2d_grad_fd_save(rpde_lnse2d* h, double max_time, double save_intervall, double beta1, double beta2, const int* points,
                             long npoints, size_t len, const char* filename, double* grad_velx, double* grad_vely, double* grad_temp);
/* functions::l2_norm                                               src/navier_stokes_lnse/functions.rs:30-58 (host arrays) */
int rpde_l2_norm(size_t len, const double* a1, const double* a2, const double* b1, const double* b2, const double* c1, const double* c2,
                 double beta1, double beta2, double* out);
/* opt_routines::steepest_descent_energy_constrained                src/navier_stokes_lnse/opt_routines.rs:16-56 (host arrays; the
   gradients are projected in place, the new state is written to *_new; alpha > 2 pi is an error like the reference's assert) */
int rpde_steepest_descent_energy_constrained(size_t len, const double* velx_0, const double* vely_0, const double* temp_0, double* grad_velx,
                                             double* grad_vely, double* grad_temp, double* velx_new, double* vely_new, double* temp_new,
                                             double beta1, double beta2, double alpha);

/* ---- operator level: funspace Space2 methods as called by rustpde ---------------------------- */
/* Space2::new(&base0(n0), &base1(n1)); base1 must be a Chebyshev-family base                     */
int rpde_space2_create(int kind0, int n0, int kind1, int n1, int device, rpde_space2** out);
int rpde_space2_destroy(rpde_space2* s);
/* shapes in elements; which: 0 physical, 1 spectral (vhat), 2 orthonormal                         */
int rpde_space2_shape(rpde_space2* s, int which, int* rows, int* cols, int* is_complex);
/* forward_inplace_par / backward_inplace_par                 src/field.rs:103-110                */
int rpde_space2_forward(rpde_space2* s, const double* v, size_t nv, double* vhat, size_t nvhat);
int rpde_space2_backward(rpde_space2* s, const double* vhat, size_t nvhat, double* v, size_t nv);
/* to_ortho_par / from_ortho                                  src/field.rs:113-123                */
int rpde_space2_to_ortho(rpde_space2* s, const double* vhat, size_t nvhat, double* out, size_t nout);
int rpde_space2_from_ortho(rpde_space2* s, const double* in, size_t nin, double* vhat, size_t nvhat);
/* gradient_par(vhat, [d0,d1], Some([s0,s1])); pass s0 = s1 = 1 for `None`   src/field.rs:127-129 */
int rpde_space2_gradient(rpde_space2* s, const double* vhat, size_t nvhat, int d0, int d1,
                         double s0, double s1, double* out, size_t nout);

/* HholtzAdi::new(&field, [c0, c1]) / Solve::solve            src/solver/hholtz_adi.rs:48-76,149-169 */
int rpde_hholtz_adi_create(rpde_space2* s, double c0, double c1, rpde_hholtz_adi** out);
int rpde_hholtz_adi_solve(rpde_hholtz_adi* hs, const double* in_ortho, size_t nin, double* out, size_t nout);
int rpde_hholtz_adi_destroy(rpde_hholtz_adi* hs);
/* Poisson::new(&field, [c0, c1]) / Solve::solve              src/solver/poisson.rs:54-94,195-236 */
int rpde_poisson_create(rpde_space2* s, double c0, double c1, rpde_poisson** out);
int rpde_poisson_solve(rpde_poisson* ps, const double* in_ortho, size_t nin, double* out, size_t nout);
int rpde_poisson_destroy(rpde_poisson* ps);
/* Host-only setup mathematics of Poisson::new for a Chebyshev x axis of n points (base_kind RPDE_CHEB_DIRICHLET or
 * RPDE_CHEB_NEUMANN; c0 as in rpde_poisson_create): no device is touched.
 *  rpde_poisson_x_spectrum: the m = n - 2 eigenvalues of inv(C_x) A_x (LAPACK dgeev, values only), order [even-parity
 *    block | odd-parity block], each descending, before the -1e-10 shift.
 *  rpde_poisson_x_eigenbasis_from_spectrum: lam -> lam_refined, fwd = Q^-1 C^-1 and bwd = Q (dense m x m, row-major,
 *    natural coefficient index; row / column k belongs to lam[k]) by Rayleigh-quotient / inverse iteration on the BANDED
 *    pencil A_x - lam C_x, bit-reproducible (no LAPACK, fixed operation order): what an engine created with the same
 *    spectrum uses.  `lam` only has to identify the eigenvalues (dgeev on the dense inv(C) A is 1e-7 relative at n = 1025
 *    for the small ones); lam_refined are the eigenvalues of the pencil to round-off.     src/solver/fdma_tensor.rs:106-154 */
int rpde_poisson_x_spectrum(int base_kind, int n, double c0, double* lam, size_t m);
int rpde_poisson_x_eigenbasis_from_spectrum(int base_kind, int n, double c0, const double* lam, size_t m,
                                            double* lam_refined, double* fwd, double* bwd);
/* Poisson::new with a supplied x spectrum (see rpde_navier2d_create_confined_with_spectrum) */
int rpde_poisson_create_with_spectrum(rpde_space2* s, double c0, double c1, const double* lam, size_t m, rpde_poisson** out);
/* Hholtz::new(&field, [c0, c1]) / Solve::solve: (I - c0 Dxx - c1 Dyy) vhat = A f with the TENSOR solver (x diagonalised,
 * one banded solve per x-row) -- the norm of Navier2DAdjoint's residual          src/solver/hholtz.rs:72-106, 164-187 */
int rpde_hholtz_create(rpde_space2* s, double c0, double c1, rpde_hholtz** out);
int rpde_hholtz_solve(rpde_hholtz* hs, const double* in_ortho, size_t nin, double* out, size_t nout);
int rpde_hholtz_destroy(rpde_hholtz* hs);
/* The x eigen-decomposition Poisson::new builds once (FdmaTensor::from_matrix,                     *
 * src/solver/fdma_tensor.rs:123-127; LAPACK dgeev/dgetri, src/solver/utils.rs:67-107): m = nx - 2    *
 * eigenvalues `lam` (before the -1e-10 shift of poisson.rs:84-87), fwd = Q^-1 C^-1 and bwd = Q as     *
 * dense m x m row-major matrices.  Setup data; read by checkers that must run the reference's       *
 * algorithm on the SAME decomposition (the solve amplifies dgeev's own round-off).       *
 * Chebyshev x axis only (a Fourier axis is already diagonal).                                       */
int rpde_poisson_eigenbasis(rpde_poisson* ps, double* lam, double* fwd, double* bwd, size_t m);
int rpde_navier2d_poisson_eigenbasis(rpde_navier2d* h, double* lam, double* fwd, double* bwd, size_t m);

/* pencil transposes (single device: LDS-tiled kernel).  out[c][r] = in[r][c]; elem = 1 | 2        *
 * funspace Decomp2d::transpose_x_to_y / transpose_y_to_x     src/field_mpi.rs:456-477             */
int rpde_transpose(const double* in, int rows, int cols, int elem, double* out, int device);
/* funspace `backward` along contiguous lines (src/field.rs:108-111) through the whole-line transform kernel   *
 * (csrc/dct_line.h: n / 16 threads per line, data in registers, four workgroups per CU): `nlines` lines of     *
 * n_in coefficients (kind 0 chebyshev: n_in = n, kind 1 cheb_dirichlet: n_in = n - 2) -> n physical values.     *
 * n = 4097 (HIP and emulation builds) or 257 (emulation build); other shapes return an error.                 */
int rpde_dct_line_backward(int kind, int n, const double* in, int nlines, double* out, int device);
/* the same with the derivative along the line in between: out = backward_ortho(scale * d/dx to_ortho(in))       *
 * (funspace `gradient` + `backward` of the orthonormal space, src/field.rs:127-129); kind 2 = cheb_neumann      *
 * (n_in = n - 2) is accepted by both entries                                                                    */
int rpde_dct_line_gradient(int kind, int n, const double* in, int nlines, double scale, double* out, int device);
/* funspace `forward` of the orthonormal Chebyshev base along contiguous lines (src/field.rs:103-106) through the same  *
 * kernel: n physical values -> n coefficients, zero from index `cut` on (cut < 0: keep all; the 2/3 rule of            *
 * src/navier_stokes/functions.rs:56-82 is cut = 2 n / 3)                                                               */
int rpde_dct_line_forward(int n, const double* in, int nlines, int cut, double* out, int device);
/* One convection term along contiguous y-lines (conv_term, src/navier_stokes/functions.rs:56-69, summed and dealiased *
 * like conv_velx / conv_vely / conv_temp, src/navier_stokes/navier_eq.rs:56-101) through the whole-line kernel:         *
 *   out = forward_ortho[ up * (backward(fx) + bx) + vp * (backward_ortho(dscale * d/dy to_ortho(f0)) + by) ], zero from   *
 * `cut` on.  fx, f0: `nlines` lines of n - 2 cheb_dirichlet coefficients; up, vp, bx, by: n physical values per line    *
 * (bx, by may both be NULL); out: n orthonormal coefficients per line.  Same shapes as rpde_dct_line_backward.          */
int rpde_conv_line(int n, const double* fx, const double* f0, const double* up, const double* vp, const double* bx,
                   const double* by, int nlines, double dscale, int cut, double* out, int device);
/* f64 GEMM used by the Poisson solve (ndarray `dot` -> dgemm, src/solver/poisson.rs:216,234):     *
 * c[M,N] = a[M,K] . b  with b given as [N,K] (transb = 1) or [K,N] (transb = 0); host buffers     */
int rpde_gemm(int M, int N, int K, const double* a, const double* b, int transb, double* c, int device);

/* micro-benchmark of one line-program shape (bench/tests only): LOAD + <what> + STORE on `nlines`  *
 * Chebyshev-Dirichlet lines of n points; what in {copy, sten, mv3, cdiff, fromortho, fdma, dct,     *
 * dct2, rfft}; returns the mean device time of one launch in ms (HIP events, `reps` launches).     *
 * what in {forward2d, backward2d, to_ortho2d, from_ortho2d}: Field2 on cheb_dirichlet(n) x          *
 * cheb_dirichlet(nlines), the operations the reference's criterion benches time                      *
 * (benches/benchmark_transform.rs:6-22, benchmark_to_ortho.rs:6-40), arrays resident in HBM          */
int rpde_microbench(const char* what, int n, int nlines, int reps, int device, double* ms);

#ifdef __cplusplus
}
#endif
#endif
