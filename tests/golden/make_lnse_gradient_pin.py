"""The reference's own validation of the LNSE adjoint gradient, run on the CPU oracle: examples/navier_lnse_test_gradient.rs
(periodic 18 x 13, Ra = 3e3, Pr = 0.1, dt = 0.01, horizon 10, beta = 0.5 / 0.5, random initial state of amplitude 1e-3) compares
the adjoint gradient (lnse_adj_grad.rs) with finite differences over every grid point (lnse_fd_grad.rs) and accepts
|g_fd - g_adj| / |g_adj| <= 0.3 per field.  The oracle (oracle/lnse.py) restates both; this script runs the example's comparison
and writes tests/golden/lnse_gradient_pin.json -- the pin of the restated adjoint equations (a sign or a term wrong in
lnse_adj_eq.rs's restatement gives O(1) here).

    python tests/golden/make_lnse_gradient_pin.py [max_time=10] [procs=8]        (about 50 CPU-minutes at max_time = 10)
"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
NX, NY, RA, PR, DT, BETA, SEED, AMP = 18, 13, 3e3, 0.1, 0.01, 0.5, 0, 1e-3


def make():
    from oracle import lnse as L
    nav = L.Navier2DLnse.new_periodic(NX, NY, RA, PR, DT, 1.0, "rbc")
    nav.init_random(AMP, SEED)
    return nav


def fd_chunk(job):
    max_time, points = job
    os.environ["OMP_NUM_THREADS"] = "1"
    nav = make()
    g = nav.grad_fd(max_time, BETA, BETA, points=points)
    return points, [float({"velx": g[0], "vely": g[1], "temp": g[2]}[k][i, j]) for k, i, j in points]


def main():
    max_time = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    t0 = time.time()
    nav = make()
    fun_val, g_adj = nav.grad_adjoint(max_time, BETA, BETA)
    g_adj = [-g for g in g_adj]                      # the example flips the sign (MAXIMIZE = false)
    points = [(k, i, j) for k in ("velx", "vely", "temp") for i in range(NX) for j in range(NY)]
    chunks = [points[c::procs * 4] for c in range(procs * 4)]
    g_fd = {k: np.zeros((NX, NY)) for k in ("velx", "vely", "temp")}
    with Pool(procs) as pool:
        for pts, vals in pool.imap_unordered(fd_chunk, [(max_time, c) for c in chunks]):
            for (k, i, j), v in zip(pts, vals):
                g_fd[k][i, j] = v
    out = dict(nx=NX, ny=NY, ra=RA, pr=PR, dt=DT, max_time=max_time, beta1=BETA, beta2=BETA, amp=AMP, seed=SEED, fun_val=fun_val,
               reference_acceptance=0.3, seconds=None)
    for k, ga in zip(("velx", "vely", "temp"), g_adj):
        out[k] = dict(rel_diff=float(np.linalg.norm(ga - g_fd[k]) / np.linalg.norm(ga)), norm_adj=float(np.linalg.norm(ga)),
                      norm_fd=float(np.linalg.norm(g_fd[k])))
        print(k, out[k], flush=True)
    out["seconds"] = time.time() - t0
    tag = "" if max_time == 10.0 else f"_T{max_time:g}"
    path = os.path.join(ROOT, "tests", "golden", f"lnse_gradient_pin{tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
