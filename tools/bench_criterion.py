"""The reference's criterion benches on the engine, with the CPU oracle timed beside them (tools only; bench.py is the driver's line).

  benches/benchmark_navier.rs:6-37     Navier2D::new_confined(n, n, 1e5, 1., 0.01, 1., "rbc").update()   n = 128, 129, 264, 265, 512, 513
  benches/benchmark_transform.rs:6-22  Field2(cheb_dirichlet(n)^2).forward()                             n = 128, 264, 512, 1024
  benches/benchmark_to_ortho.rs:6-40   .to_ortho() / .from_ortho(&array)                                 n = 128, 264, 512

usage: python tools/bench_criterion.py [--out file.json] [--no-cpu]
One JSON object: per bench and size the device time per call (HIP events, arrays resident in HBM; steps: wall time of
update(K) / K) and the oracle's wall time per call on the host cores (NumPy / SciPy, pocketfft on all cores)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--sizes", default="128,129,264,265,512,513,1024,1025")
    a = ap.parse_args()
    import rustpde_mpi_amd as R
    from oracle import bases as B, navier as N
    out = {"navier_update": [], "transform_forward": [], "to_ortho": [], "from_ortho": []}
    for n in [int(v) for v in a.sizes.split(",")]:
        nav = R.Navier2D.new_confined(n, n, 1e5, 1.0, 0.01, 1.0, "rbc")
        nav.set_velocity(0.2, 1.0, 1.0)
        nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(20)
        steps = 400 if n < 600 else 200
        t0 = time.perf_counter()
        nav.update(steps)
        ms = 1e3 * (time.perf_counter() - t0) / steps
        assert nav.exit() is False
        kinds = sorted({kind for _, _, _, _, kind in nav.schedule()})
        row = {"n": n, "gpu_ms_per_update": ms, "launches_per_step": len(nav.schedule()), "kernels": kinds}
        del nav
        if not a.no_cpu:
            ora = N.Navier2D.new_confined(n, n, 1e5, 1.0, 0.01, 1.0, "rbc", eig_mode="parity")
            ora.set_velocity(0.2, 1.0, 1.0)
            ora.set_temperature(0.2, 1.0, 1.0)
            ora.update()
            k = 10 if n < 600 else 4
            t0 = time.perf_counter()
            for _ in range(k):
                ora.update()
            row["cpu_oracle_ms_per_update"] = 1e3 * (time.perf_counter() - t0) / k
        out["navier_update"].append(row)
        print(row, flush=True)
    for key, what, sizes in (("transform_forward", "forward2d", (128, 264, 512, 1024, 129, 513, 1025)), ("to_ortho", "to_ortho2d", (128, 264, 512)),
                             ("from_ortho", "from_ortho2d", (128, 264, 512))):
        for n in sizes:
            row = {"n": n, "gpu_ms_per_call": R.microbench(what, n, n, reps=200)}
            if not a.no_cpu:
                sp = B.Space2(B.cheb_dirichlet(n), B.cheb_dirichlet(n))
                v = np.arange(n * n, dtype=float).reshape(n, n)
                vh = sp.forward(v)
                arg = {"forward2d": v, "to_ortho2d": vh, "from_ortho2d": v}[what]
                fn = {"forward2d": sp.forward, "to_ortho2d": sp.to_ortho, "from_ortho2d": sp.from_ortho}[what]
                fn(arg)
                t0 = time.perf_counter()
                for _ in range(20):
                    fn(arg)
                row["cpu_oracle_ms_per_call"] = 1e3 * (time.perf_counter() - t0) / 20
            out[key].append(row)
            print(key, row, flush=True)
    try:
        from threadpoolctl import threadpool_info
        out["host"] = {"cores": os.cpu_count(), "blas_threads": max([int(i.get("num_threads", 1)) for i in threadpool_info()] + [1])}
    except Exception:   # noqa: BLE001
        out["host"] = {"cores": os.cpu_count()}
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
