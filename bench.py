#!/usr/bin/env python
"""bench.py -- timesteps/s of the 2-D Rayleigh-Benard step (f64) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nx 4097 --ny 4097] [--no-cpu-baseline]

A "step" is one `Navier2D::update()` (src/navier_stokes/navier.rs:438-466 of the reference) on the
confined 4097 x 4097 case (BASELINE.json: the configuration the >= 50 timesteps/s target is quoted
on; it fits one GPU).  Inputs (the deterministic initial condition of examples/navier_rbc.rs) are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME case is pencil-sharded over
the N GPUs (the reference's Navier2DMpi path, BASELINE.json configs[3]); every layout change of
the step is an all-to-all over RCCL (native grouped ncclSend/ncclRecv on the engine's stream by
default; RPDE_TRANSPORT=torch for torch.distributed's all_to_all_single).  Total work is fixed, so
`scaling` is "strong" at every N and `value` is the steps/s of the whole job.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X f64 matrix peak (spec)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--nx", type=int, default=4097)
    p.add_argument("--ny", type=int, default=4097)
    p.add_argument("--ra", type=float, default=1e8)
    p.add_argument("--dt", type=float, default=2e-4)
    p.add_argument("--periodic", action="store_true")
    p.add_argument("--aspect", type=float, default=1.0)
    p.add_argument("--bc", default="rbc", choices=["rbc", "hc"], help="boundary condition of the temperature (navier.rs:245-248)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-steps", type=int, default=2)
    p.add_argument("--profile-steps", type=int, default=3)
    # harness self-test on a box without a GPU (tests/test_bench_dryrun.py): runs the SAME script
    # against the host emulation build of the kernel sources.  Never a measurement: the JSON line
    # says so in `data` and carries "dry_run": true.
    p.add_argument("--dry-run-emu", action="store_true", help=argparse.SUPPRESS)
    # also report the parity with an oracle that ran its own dgeev (about a minute more at 4097^2)
    p.add_argument("--parity-independent", action="store_true")
    p.add_argument("--no-extended-golden", action="store_true", help="stop the independent-golden comparison at step 200")
    p.add_argument("--no-cpu-single-thread", action="store_true")
    # the other solvers of SURVEY 8f-4 (not the driver's line): Navier2DAdjoint::update (steady_adjoint.rs:541-608), Navier2DLnse::update /
    # update_adjoint (lnse.rs:263-288, lnse_adj_grad.rs:71-99), Navier2DNonLin::update (nonlin.rs:264-296)
    p.add_argument("--solver", default="navier", choices=["navier", "adjoint", "lnse", "lnse_adjoint", "nonlin"])
    return p.parse_args()


def cpu_baseline(args, eig=None):
    """The oracle (CPU restatement of rustpde, OpenBLAS path) timed on the host cores: a bounded
    sample of the same workload.  Returns (baseline dict, oracle instance after 1 + cpu_steps
    steps) -- the instance is the checker of the `parity` object.  `eig`: the engine's x
    eigen-decomposition (setup data, rpde_navier2d_poisson_eigenbasis) for the oracle to run on;
    None: the oracle calls LAPACK itself."""
    from oracle import navier as N
    ctor = N.Navier2D.new_periodic if args.periodic else N.Navier2D.new_confined
    t0 = time.perf_counter()
    kw = {"eig_mode": "parity"} if eig is None else {"eig_override": eig}
    ora = ctor(args.nx, args.ny, args.ra, 1.0, args.dt, args.aspect, args.bc, **kw)
    ora.set_velocity(0.2, 1.0, 1.0)
    ora.set_temperature(0.2, 1.0, 1.0)
    setup = time.perf_counter() - t0
    ora.update()  # warm-up (FFT plans, page faults)
    from oracle import timing
    timing.reset(True)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        ora.update()
    dt = time.perf_counter() - t0
    phases = {k: v / args.cpu_steps for k, v in timing.ACC.items()}
    phases["other (elementwise, copies)"] = dt / args.cpu_steps - sum(phases.values())
    timing.reset(False)
    # threads actually used: OpenBLAS' pool for the two Poisson GEMMs and pocketfft's workers for the
    # transforms (scipy.fft workers = -1); the banded sweeps and NumPy elementwise code run on one core
    cores = 1
    try:
        from threadpoolctl import threadpool_info
        cores = max([int(i.get("num_threads", 1)) for i in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count()
    base = {"value": args.cpu_steps / dt, "unit": "timesteps/s", "cores": cores, "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{args.cpu_steps} steps of the same {args.nx}x{args.ny} case after 1 warm-up step "
                      f"(NumPy/SciPy oracle, OpenBLAS dgemm on {cores} threads + pocketfft on all cores; setup {setup:.1f}s not timed)",
            "seconds_per_step_by_phase": {k: round(v, 4) for k, v in sorted(phases.items(), key=lambda kv: -kv[1])}}
    if not args.no_cpu_single_thread:
        # the reference's advised setting OPENBLAS_NUM_THREADS=1 (README.md:45-47): one more step on a COPY of
        # nothing -- the same instance simply takes one more step with BLAS limited to one thread; the
        # parity engine below takes the same number of steps
        try:
            from threadpoolctl import threadpool_limits
            import scipy.fft as _sfft
            with threadpool_limits(limits=1), _sfft.set_workers(1):
                from oracle import bases as _B
                saved = _B.WORKERS
                _B.WORKERS = 1
                t0 = time.perf_counter()
                ora.update()
                one = time.perf_counter() - t0
                _B.WORKERS = saved
            base["single_thread"] = {"value": 1.0 / one, "unit": "timesteps/s", "cores": 1,
                                     "sample": "1 further step with OpenBLAS and pocketfft limited to one thread"}
            return base, ora, 2 + args.cpu_steps
        except Exception as exc:   # noqa: BLE001
            base["single_thread"] = {"error": repr(exc)}
    return base, ora, 1 + args.cpu_steps


PARITY_TOL = 1e-10   # BASELINE.json: u, v, T, p match the CPU reference within 1e-10 relative L2 (f64)


from tests.bounds import independent_golden_bound   # noqa: E402  (the one definition, shared with tests/checks.py)


def parity_vs_oracle(make, ora, nsteps, shared):
    """A FRESH engine, the same deterministic initial condition, the same number of steps as the
    oracle instance of the cpu_baseline leg has taken: relative L2 of u, v, T, p in physical space."""
    import numpy as np
    nav = make(None)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    nav.update(nsteps)
    got, want = nav.physical_fields(), ora.physical_fields()
    rel = {k: float(np.linalg.norm(got[k] - want[k]) / np.linalg.norm(want[k])) for k in want}
    del nav
    return {"steps": nsteps, "rel_l2": rel, "tol": PARITY_TOL,
            "checker": "oracle/navier.py (NumPy/SciPy restatement of Navier2D::update)",
            "poisson_eigenbasis": ("shared: the oracle runs on the engine's dgeev output (setup data, outside the time "
                                   "step; the reference's Poisson solve amplifies dgeev's own round-off -- DESIGN.md 4)"
                                   if shared else "independent: each side called LAPACK itself"),
            "ok": all(v < PARITY_TOL for v in rel.values())}


def parity_independent_golden(make, args):
    """The engine on its OWN setup (one dgeev per parity block, C++) against committed samples of the oracle run in the
    REFERENCE's setup (ONE dgeev of the whole x operator, src/solver/utils.rs:67-99) -- tests/golden/make_headline_golden.py,
    up to 200 steps of this workload.  Per snapshot and field: relative L2 over the sample points, next to the oracle's own
    full-vs-parity difference at that step (the start-up transient of DESIGN.md section 4).  None when no golden file
    exists for the workload."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", f"headline_{args.nx}_full.npz")
    if args.periodic or args.bc != "rbc" or args.nx != args.ny or not os.path.exists(path):
        return None
    g = np.load(path)
    if abs(float(g["ra"]) - args.ra) > 0 or abs(float(g["dt"]) - args.dt) > 0 or abs(args.aspect - 1.0) > 0:
        return None
    stride = int(g["stride"])
    nav = make(None)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    rows, done = [], 0
    for s in [int(v) for v in g["snaps"]]:
        if f"velx_{s}" not in g.files:
            continue
        nav.update(s - done)
        done = s
        f = nav.physical_fields()
        rel = {k: float(np.linalg.norm(f[k][::stride, ::stride] - g[f"{k}_{s}"]) / np.linalg.norm(g[f"{k}_{s}"]))
               for k in ("velx", "vely", "temp", "pres")}
        fvp = {k: float(g[f"{k}_{s}_full_vs_parity"]) for k in rel}
        bound = {k: independent_golden_bound(fvp[k], n=args.nx, step=s, field=k) for k in rel}
        fvp = {k: (v if v == v else None) for k, v in fvp.items()}     # not measured: null (NaN is not JSON)
        rows.append({"steps": s, "rel_l2": rel, "oracle_full_vs_parity": fvp, "bound": bound,
                     "ok": all(rel[k] < bound[k] for k in rel)})
    # the second, longer run of the same oracle setup (headline_<n>_full_extended.npz, to step 800): the same engine goes
    # on; PLAIN tol, no envelope -- u, v, T below it at every snapshot, the pressure from `first` on (start-up transient of
    # two independent eigen-decompositions)
    ext_path = os.path.join(ROOT, "tests", "golden", f"headline_{args.nx}_full_extended.npz")
    if os.path.exists(ext_path) and not args.no_extended_golden:
        ge = np.load(ext_path)
        for s in [int(v) for v in ge["snaps"]]:
            if s <= done or f"velx_{s}" not in ge.files:
                continue
            nav.update(s - done)
            done = s
            f = nav.physical_fields()
            rel = {k: float(np.linalg.norm(f[k][::stride, ::stride] - ge[f"{k}_{s}"]) / np.linalg.norm(ge[f"{k}_{s}"]))
                   for k in ("velx", "vely", "temp", "pres")}
            rows.append({"steps": s, "golden": os.path.relpath(ext_path, ROOT), "rel_l2": rel, "oracle_full_vs_parity": None,
                         "bound": {k: PARITY_TOL for k in rel},
                         "ok": all(rel[k] < PARITY_TOL for k in ("velx", "vely", "temp"))})
    del nav
    first = None
    for r in reversed(rows):        # the first snapshot from which ALL fields stay below tol to the end of the run
        if all(v < PARITY_TOL for v in r["rel_l2"].values()):
            first = r["steps"]
        else:
            break
    if rows and rows[-1]["steps"] > 200 and first is None:
        rows[-1]["ok"] = False      # an extended run has to end below tol in all four fields
    first_uvt = next((r["steps"] for r in rows if all(r["rel_l2"][k] < PARITY_TOL for k in ("velx", "vely", "temp"))), None)
    return {"golden": os.path.relpath(path, ROOT), "sample_stride": stride,
            "setup": "engine: own dgeev per parity block; golden: oracle with ONE dgeev of the whole operator (the reference's algorithm)",
            "snapshots": rows, "first_snapshot_all_fields_below_tol": first, "first_snapshot_u_v_T_below_tol": first_uvt,
            "tol": PARITY_TOL, "ok": all(r["ok"] for r in rows)}


def parity_shared_basis_golden(args, library, device):
    """SAME INPUTS over the full horizon: the engine is created with the x eigenvalues of the committed golden
    (tests/golden/make_shared_basis_golden.py: `x_spectrum=` -- the library rebuilds the oracle's eigenbasis bit for bit,
    no LAPACK), runs this workload for the golden's 200 steps, and every snapshot is compared with the CPU oracle's samples:
    plain tol (1e-10) on u, v, T and p from the first step.  None when no golden exists for the workload."""
    import numpy as np
    import rustpde_mpi_amd as R
    tag = "" if (args.ra == 1e8 and args.dt == 2e-4) else f"_ra{args.ra:g}_dt{args.dt:g}"
    path = os.path.join(ROOT, "tests", "golden", f"shared_basis_{args.nx}{tag}.npz")
    if args.periodic or args.bc != "rbc" or args.nx != args.ny or abs(args.aspect - 1.0) > 0 or not os.path.exists(path):
        return None
    g = np.load(path)
    if abs(float(g["ra"]) - args.ra) > 0 or abs(float(g["dt"]) - args.dt) > 0:
        return None
    stride = int(g["stride"])
    nav = R.Navier2D.new_confined(args.nx, args.ny, args.ra, 1.0, args.dt, args.aspect, "rbc", device=device, library=library,
                                  init_random=None, x_spectrum=np.ascontiguousarray(g["x_spectrum"]))
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    rows, done = [], 0
    for s in [int(v) for v in g["snaps"]]:
        if f"velx_{s}" not in g.files:
            continue
        nav.update(s - done)
        done = s
        f = nav.physical_fields()
        rel = {k: float(np.linalg.norm(f[k][::stride, ::stride] - g[f"{k}_{s}"]) / np.linalg.norm(g[f"{k}_{s}"]))
               for k in ("velx", "vely", "temp", "pres")}
        nrm = {k: float(abs(np.linalg.norm(f[k]) - float(g[f"{k}_{s}_norm"])) / float(g[f"{k}_{s}_norm"])) for k in rel}
        rows.append({"steps": s, "rel_l2": rel, "full_field_norm_rel_diff": nrm,
                     "ok": all(v < PARITY_TOL for v in rel.values()) and all(v < PARITY_TOL for v in nrm.values())})
    del nav
    return {"golden": os.path.relpath(path, ROOT), "sample_stride": stride, "tol": PARITY_TOL,
            "setup": "both sides on the eigenbasis the library rebuilds from the golden's x eigenvalues (bit-identical, no LAPACK)",
            "snapshots": rows, "steps": rows[-1]["steps"] if rows else 0,
            "worst_rel_l2": {k: max(r["rel_l2"][k] for r in rows) for k in ("velx", "vely", "temp", "pres")} if rows else None,
            "ok": bool(rows) and all(r["ok"] for r in rows)}


def pmc_traffic(workload, tag):
    """HBM bytes per launch of kernel `tag` from the committed rocprofv3 --pmc passes
    (tools/pmc_step.py + tools/pmc_traffic.py -> profiles/*pmc_traffic*.json, FETCH_SIZE and
    WRITE_SIZE collected in separate passes and corrected as MI355X_MICROARCH.md prescribes).
    None when no pass for this workload has been committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("workload") == workload and tag in d.get("per_launch", {}):
            src = {"file": os.path.relpath(path, ROOT), "commit": d.get("commit"),
                   # the counters were collected on these kernel sources?  (sha256 over rustpde_mpi_amd/csrc/*.{h,cc})
                   "stale": d.get("csrc_sha256") != csrc_sha256()}
            if d.get("step_total_algorithmic_bytes"):
                src["step_ratio"] = d["step_total_traffic_bytes"] / d["step_total_algorithmic_bytes"]
            return d["per_launch"][tag]["traffic_bytes"], src
    return None, None


def csrc_sha256():
    import hashlib
    root = os.path.join(ROOT, "rustpde_mpi_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".h", ".cc")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()


def cpu_baseline_other_solver(args, name, which, step):
    """The oracle of one of the SURVEY 8f-4 solvers timed on the host cores (bounded sample: 1 warm-up + cpu_steps updates);
    returns (baseline dict, the oracle instance after those updates -- the checker of the parity leg, its name)."""
    from oracle import adjoint as OA, lnse as OL
    ocls = OA.Navier2DAdjoint if args.solver == "adjoint" else getattr(OL, name)
    ora = getattr(ocls, which)(args.nx, args.ny, args.ra, 1.0, args.dt, args.aspect, "rbc", eig_mode="parity")
    ora.set_velocity(0.2, 1.0, 1.0)
    ora.set_temperature(0.2, 1.0, 1.0)
    step(ora, 1, False)
    t0 = time.perf_counter()
    step(ora, args.cpu_steps, False)
    dt = time.perf_counter() - t0
    base = {"value": args.cpu_steps / dt, "unit": "updates/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"{args.cpu_steps} updates of the same case after 1 warm-up update (NumPy/SciPy oracle)"}
    return base, ora, f"oracle ({ocls.__module__}.{name})"


def bench_other_solver(args):
    """`--solver adjoint | lnse | lnse_adjoint | nonlin`: updates per second of one of the SURVEY 8f-4 solvers on one GPU, with the
    CPU oracle timed beside it and the parity of a fresh engine against that oracle instance (spectral u, v, T: tol; pressure: 1e-8,
    tests/test_adjoint.py).  One JSON line in the format of the main bench (no per-launch roofline: these steps are
    compositions of the generic device operators, profiled as a whole in profiles/)."""
    import numpy as np
    import rustpde_mpi_amd as R
    name = {"adjoint": "Navier2DAdjoint", "lnse": "Navier2DLnse", "lnse_adjoint": "Navier2DLnse", "nonlin": "Navier2DNonLin"}[args.solver]
    ecls = getattr(R, name)
    which = "new_periodic" if args.periodic else "new_confined"
    kw = {} if args.solver == "adjoint" else {"mean_file": "/nonexistent/mean.h5"}

    def make():
        nav = getattr(ecls, which)(args.nx, args.ny, args.ra, 1.0, args.dt, args.aspect, "rbc", **kw)
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        return nav

    def step(z, n, engine):
        if args.solver == "lnse_adjoint":
            if engine:
                z.update_adjoint(n)
            else:
                for _ in range(n):
                    z.update_adjoint()
        elif engine:
            z.update(n)
        else:
            for _ in range(n):
                z.update()

    nav = make()
    step(nav, args.warmup, True)
    nav.div_norm()                       # drains the stream
    t0 = time.perf_counter()
    step(nav, args.steps, True)
    nav.div_norm()
    elapsed = time.perf_counter() - t0
    bad = nav.exit() if args.solver != "adjoint" else (nav.div_norm() != nav.div_norm())
    del nav
    out = {"metric": f"updates/sec ({name}{'::update_adjoint' if args.solver == 'lnse_adjoint' else '::update'}, f64)",
           "value": args.steps / elapsed, "unit": "updates/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic (set_velocity(0.2,1,1), set_temperature(0.2,1,1); default mean fields)",
           "config": {"workload": f"{name}::{which} {args.nx}x{args.ny} Ra={args.ra:g} Pr=1 dt={args.dt:g} aspect={args.aspect:g} bc=rbc",
                      "parallelism": "single GPU"},
           "roofline": None, "nan": bool(bad)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"], ora, checker = cpu_baseline_other_solver(args, name, which, step)
        nav = make()
        step(nav, 1 + args.cpu_steps, True)
        names = ("velx", "vely", "temp", "pres")
        got, want = nav.spectral_fields(names), ora.spectral_fields(names)   # spectral coefficients: every solver's oracle has them
        rel = {k: float(np.linalg.norm(got[k] - want[k]) / max(np.linalg.norm(want[k]), 1e-300)) for k in names}
        out["parity"] = {"steps": 1 + args.cpu_steps, "rel_l2": rel, "tol": PARITY_TOL, "tol_pres": 1e-8,
                         "checker": f"{checker}, independent eigen-decompositions",
                         "ok": all(rel[k] < (1e-8 if k == "pres" else PARITY_TOL) for k in names)}
    print(json.dumps(out))
    if "parity" in out and not out["parity"]["ok"]:
        sys.exit(f"parity vs the oracle: {out['parity']['rel_l2']}")


def main():
    args = parse()
    if args.solver != "navier":
        bench_other_solver(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # RPDE_BENCH_SHARE_GPU=1 (tests on a 1-GPU box): all ranks on device 0, gloo transport
        share = os.environ.get("RPDE_BENCH_SHARE_GPU") == "1"
        if share or args.dry_run_emu:
            local_rank = 0
        if not args.dry_run_emu:                 # (the emulation dry run of the harness has no device: gloo, host buffers)
            torch.cuda.set_device(local_rank)
        if share or args.dry_run_emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import rustpde_mpi_amd as R

    comm = None
    transport = "none"
    ctor = R.Navier2D.new_periodic if args.periodic else R.Navier2D.new_confined

    library = None
    if args.dry_run_emu:
        from tests.emu.build_emu import build as build_emu
        from rustpde_mpi_amd._capi import Lib
        library = Lib(build_emu())

    def make(comm):
        return ctor(args.nx, args.ny, args.ra, 1.0, args.dt, args.aspect, args.bc, device=local_rank, comm=comm,
                    library=library)

    if world > 1:
        from rustpde_mpi_amd.dist import RcclComm, TorchComm
        import torch
        # default: the native transport (grouped ncclSend/ncclRecv on the engine's stream);
        # RPDE_TRANSPORT=torch selects the torch.distributed all_to_all_single callback instead.
        # If the native communicator cannot be created on ANY rank, all ranks fall back together.
        want_native = dist.get_backend() == "nccl" and os.environ.get("RPDE_TRANSPORT", "rccl") == "rccl"
        nav = None
        if want_native:
            ok = 1
            try:
                nav = make(RcclComm())
                transport = "rccl-native"
            except Exception as exc:   # noqa: BLE001 - any failure means: use the other transport
                print(f"[bench] rank {rank}: native RCCL transport unavailable ({exc}); falling back", flush=True)
                ok = 0
            flag = torch.tensor([ok], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                nav = None
        if nav is None:
            comm = TorchComm(device_buffers=not args.dry_run_emu)
            nav = make(comm)
            transport = "torch-" + dist.get_backend()
    else:
        nav = make(None)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)

    def barrier():
        if dist is not None:
            import torch
            if not args.dry_run_emu:
                torch.cuda.synchronize()
            dist.barrier()

    nav.update(args.warmup)
    # which kernel dominates?  (per-launch HIP events, outside the timed region)
    prof = nav.profile(args.profile_steps)
    sched = nav.schedule()
    tot = sum(r["ms_total"] for r in prof)
    # roofline kernel: the dominant compute kernel (line programs "S*", GEMMs "G*"); when sharded the
    # exchanges ("T*", halos "H*") are xGMI-bound and reported in `exchange` instead
    cand = [r for r in prof if r["tag"][0] in "SG"] if world > 1 else prof
    dom = max(cand, key=lambda r: r["ms_total"])
    nav.set_timed_tag(dom["tag"])

    barrier()
    t0 = time.perf_counter()
    nav.update(args.steps)            # returns after the stream has drained (HIP events)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = nav.last_update_ms()
    tag_ms, tag_n = nav.get_timed()
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the loop a host that keeps rustpde::integrate runs (src/lib.rs:187-219): update(); exit(); per step
    barrier()
    t1 = time.perf_counter()
    loop_steps = max(1, min(args.steps, 20))
    bad_loop = False
    for _ in range(loop_steps):
        nav.update(1)
        bad_loop = nav.exit() or bad_loop
    barrier()
    loop_ms = 1e3 * (time.perf_counter() - t1) / loop_steps

    bad = nav.exit() or bad_loop          # collective when sharded: every rank takes part
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    assert not bad, "NaN in the divergence after the timed run"

    per_launch_ms = max(tag_ms / max(tag_n, 1), 1e-9)
    if dom["flops"] > 0:
        achieved = dom["flops"] / (per_launch_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F64_PEAK_TFLOPS, "traffic": None}
    else:
        achieved = dom["bytes"] / (per_launch_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None}
    traffic, traffic_src = pmc_traffic(f"{'periodic' if args.periodic else 'confined'} {args.nx}x{args.ny}", dom["tag"])
    if world == 1 and traffic is not None:
        roof["traffic"] = traffic
        roof["traffic_source"] = traffic_src
        # HBM bytes of the WHOLE step from the same counter passes over its algorithmic bytes (1.0 = nothing read twice)
        roof["pmc_step_ratio"] = traffic_src.pop("step_ratio", None)
    roof["algorithmic_per_launch"] = dom["flops"] if dom["flops"] > 0 else dom["bytes"]
    roof["kernel"] = dom["tag"]
    roof["launches_timed"] = tag_n
    roof["ms_per_launch"] = per_launch_ms
    # per-phase view of one step (profile pass): time share and algorithmic throughput
    phases = []
    for r in sorted(prof, key=lambda r: -r["ms_total"])[:8]:
        ms = r["ms_total"] / r["launches"]
        phases.append({"kernel": r["tag"], "share": round(r["ms_total"] / tot, 4), "ms_per_launch": round(ms, 4),
                       "GB/s": round(r["bytes"] / (ms * 1e-3) / 1e9, 1),
                       "TFLOP/s": round(r["flops"] / (ms * 1e-3) / 1e12, 2)})
    # transform pass: the reference op sequence has 13 two-dimensional transforms =
    # 416 nx ny bytes.  Here all of them live in the stages S1-S3 (which also carry the fused
    # stencils, gradients, products, RHS assembly and the x part of the Helmholtz solves) plus the
    # pencil transposes T1/T2 between the x and the y pass.
    def stage_ms(prefixes):
        return sum(r["ms_total"] for r in prof if r["tag"].split(" ")[0] in prefixes) / args.profile_steps

    def stage_bytes(prefixes):   # algorithmic bytes the launches of these stages move in ONE step
        return sum(r["bytes"] * r["launches"] for r in prof if r["tag"].split(" ")[0] in prefixes) / args.profile_steps
    line_ms = stage_ms(("S1", "S2", "S3"))
    moved = stage_bytes(("S1", "S2", "S3"))
    moved_t = moved + stage_bytes(("T1", "T2"))
    ms_t = line_ms + stage_ms(("T1", "T2"))
    ref_bytes = 416.0 * args.nx * args.ny
    pure = [r for r in prof if r["tag"] in ("S1 x: state -> phys-x + d/dx", "S1 x: state -> phys-x", "S2 y: velx, vely -> phys",
                                            "S2 y: velx -> phys", "S2 y: vely -> phys", "S2 y: velx -> phys + vely -> phys")]
    transform_pass = {
        # the headline figure: bytes the transform stages REALLY move (their loads and stores) over their time
        "bytes_moved_S1_S2_S3": moved,
        "ms_stages_S1_S2_S3": line_ms,
        "GB/s": moved / (line_ms * 1e-3) / 1e9 if line_ms > 0 else None,
        "frac_of_hbm_peak": moved / (line_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if line_ms > 0 else None,
        "with_transposes_T1_T2": {"bytes_moved": moved_t, "ms": ms_t,
                                  "frac_of_hbm_peak": moved_t / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_t > 0 else None},
        # for comparison only: the reference op sequence has 13 two-dimensional transforms = 416 nx ny bytes
        # (SURVEY.md 8d); this engine executes 11 of them (the lift's gradients are constants)
        # SURVEY.md 8d's unit (one 2-D transform = 32 nx ny bytes) times the 11 transforms this engine executes per step
        "executed_transforms": 11,
        "executed_transforms_bytes": 352.0 * args.nx * args.ny,
        "executed_transforms_frac": 352.0 * args.nx * args.ny / (line_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if line_ms > 0 else None,
        "reference_sequence_bytes": ref_bytes,
        "reference_sequence_equiv_frac": ref_bytes / (line_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if line_ms > 0 else None,
        "pure_1d_transform_kernels": [
            {"kernel": r["tag"], "GB/s": round(r["bytes"] / (r["ms_total"] / r["launches"] * 1e-3) / 1e9, 1),
             "frac_of_hbm_peak": round(r["bytes"] / (r["ms_total"] / r["launches"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            for r in pure],
    }
    # the scalars of the transform pass inside `roofline` (the object the driver's record keeps): north_star's ">= 40 % of the HBM
    # roofline on the transform pass" is `transform_pass_frac` (bytes the stages S1-S3 really move / their time / 8 TB/s);
    # `transform_pass_ref_equiv_frac` prices the same time with the reference sequence's 13 transforms
    roof["transform_pass_frac"] = transform_pass["frac_of_hbm_peak"]
    roof["transform_pass_ref_equiv_frac"] = transform_pass["reference_sequence_equiv_frac"]
    # round 6: the stages no longer read what is redundant in the time-independent lift arrays (RPDE_LIFT_STRUCT; 0.54 GB per step
    # at 4097^2 with "rbc"): `transform_pass_frac` counts the bytes still moved and reads LOWER for a pass that got faster -- the
    # time of the pass and SURVEY 8d's own unit (11 executed transforms x 32 nx ny bytes) are reported beside it
    roof["transform_pass_ms"] = line_ms
    roof["transform_pass_executed_transforms_frac"] = transform_pass["executed_transforms_frac"]
    out = {
        "metric": "timesteps/sec (2D RBC, f64)",
        "value": args.steps / elapsed,
        "unit": "timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "device_ms_per_step": dev_ms / args.steps,
        "ms_per_step_update_plus_exit": loop_ms,   # `update(1); exit()` per step, as rustpde::integrate drives it
        "higher_is_better": True,
        "scaling": "strong",      # the SAME case at every N (pencil-sharded when N > 1): total work is fixed
        "vs_baseline": None,
        "dtype": "f64",
        "data": ("DRY RUN on the host emulation build -- NOT a measurement" if args.dry_run_emu else
                 "synthetic (deterministic IC of examples/navier_rbc.rs: set_velocity(0.2,1,1), set_temperature(0.2,1,1))"),
        "config": {"workload": f"Navier2D::new_{'periodic' if args.periodic else 'confined'} "
                               f"{args.nx}x{args.ny} Ra={args.ra:g} Pr=1 dt={args.dt:g} aspect={args.aspect:g} bc={args.bc}",
                   "parallelism": "single GPU" if world == 1 else
                                  f"pencil-sharded over {world} GPUs (x-/y-pencils, all-to-all over RCCL, transport {transport})"},
        "roofline": roof,
        "phases": phases,
        "transform_pass": transform_pass,
        # which stages of this engine run whole-line kernels (the default wherever they cover the line length; the
        # RPDE_*_LINE switches are A/B overrides)
        "step_kernels": {k: sorted({t for t, _, _, _, kind in sched if kind == k})
                         for k in sorted({kind for _, _, _, _, kind in sched}) if k.startswith("whole-line")},
    }
    if args.dry_run_emu:
        out["dry_run"] = True
    if world > 1:
        sent, nx_ = nav.comm_stats()
        out["exchange"] = {"alltoalls_per_step": nx_, "bytes_sent_per_gpu_per_step": sent,
                           "GB/s_per_gpu": sent * args.steps / elapsed / 1e9,
                           "xgmi_peak_GB/s_per_gpu": 7 * 153.0,
                           # every exchange of one step as rank 0 saw it in the profile pass (HIP events around pack + all-to-all +
                           # unpack, serial order): the measured form of DESIGN.md section 6's table; bytes = algorithmic bytes of
                           # the launch on this rank (an array transpose: its 1 / P share read and written once)
                           "per_exchange": [{"tag": r["tag"], "per_step": r["launches"] / args.profile_steps,
                                             "ms": round(r["ms_total"] / r["launches"], 4), "bytes": r["bytes"]}
                                            for r in prof if r["tag"][0] in "TH" or "summary" in r["tag"]],
                           "ms_per_step_in_exchanges": round(sum(r["ms_total"] for r in prof if r["tag"][0] in "TH" or "summary" in r["tag"])
                                                             / args.profile_steps, 4)}
    if world == 1 and not args.no_cpu_baseline:
        eig = None if args.periodic else nav.poisson_eigenbasis()
        del nav   # free the timed engine's HBM before the parity engine is built
        out["cpu_baseline"], ora, osteps = cpu_baseline(args, eig)
        out["parity"] = parity_vs_oracle(make, ora, osteps, shared=eig is not None)
        shared_gold = parity_shared_basis_golden(args, library, local_rank)
        if shared_gold is not None:
            out["parity_shared_basis_golden"] = shared_gold
        gold = parity_independent_golden(make, args)
        if gold is not None:
            out["parity_independent_golden"] = gold
        if args.parity_independent and eig is not None:
            del ora, eig
            _, ora2, osteps = cpu_baseline(args, None)
            ind = parity_vs_oracle(make, ora2, osteps, shared=False)
            out["parity_independent_setup"] = {k: ind[k] for k in ("steps", "rel_l2", "poisson_eigenbasis")}
    # the per-snapshot arrays of the golden comparisons (30 KB at 4097^2) go to a side file; the line keeps, per comparison,
    # the snapshot steps, the worst relative L2 per field and the verdict -- what the exit code below is decided from
    detail = {}
    for key in ("parity_shared_basis_golden", "parity_independent_golden"):
        if key in out:
            rows = out[key].pop("snapshots")
            detail[key] = rows
            out[key]["snapshot_steps"] = [r["steps"] for r in rows]
            out[key]["worst_rel_l2"] = {k: max(r["rel_l2"][k] for r in rows) for k in ("velx", "vely", "temp", "pres")} if rows else None
            out[key]["failed_snapshots"] = [r for r in rows if not r["ok"]]
    if detail:
        dpath = os.environ.get("RPDE_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_parity_detail.json"))
        try:
            os.makedirs(os.path.dirname(dpath), exist_ok=True)
            with open(dpath, "w") as f:
                json.dump(detail, f)
            out["parity_detail_file"] = os.path.relpath(dpath, ROOT)
        except OSError as exc:
            out["parity_detail_file"] = f"not written: {exc}"
    print(json.dumps(out))
    if "parity" in out and not out["parity"]["ok"]:
        sys.exit(f"parity vs the oracle above {PARITY_TOL}: {out['parity']['rel_l2']}")
    if "parity_shared_basis_golden" in out and not out["parity_shared_basis_golden"]["ok"]:
        bad = out["parity_shared_basis_golden"]["failed_snapshots"]
        sys.exit(f"parity vs the same-inputs golden above {PARITY_TOL}: {bad}")
    if "parity_independent_golden" in out and not out["parity_independent_golden"]["ok"]:
        bad = out["parity_independent_golden"]["failed_snapshots"]
        sys.exit(f"parity vs the independent-setup golden above its bound: {bad}")
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
