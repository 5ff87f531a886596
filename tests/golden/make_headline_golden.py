"""Generate tests/golden/headline_<n>_full.npz: Navier2D::new_confined n x n, Ra = 1e8, Pr = 1,
dt = 2e-4, aspect 1, bc "rbc" (n = 4097: the bench.py workload, BASELINE.json configs[3] on one GPU),
deterministic IC of examples/navier_rbc.rs, advanced 200 steps with the CPU oracle in the REFERENCE's
setup: ONE dgeev of the whole x operator (eig_mode="full", src/solver/utils.rs:67-99,
fdma_tensor.rs:106-154) -- independent of the engine, which runs one dgeev per parity block in its own
C++ setup code.  A second oracle run with eig_mode="parity" measures how far two valid LAPACK
eigenbases of the same matrix are apart at each snapshot (the start-up transient of DESIGN.md section 4).

    python tests/golden/make_headline_golden.py run 4097 full   &     (about 1.5 h on 4 cores)
    python tests/golden/make_headline_golden.py run 4097 parity &
    python tests/golden/make_headline_golden.py combine 4097
    python tests/golden/make_headline_golden.py compare 4097 <dir A> <dir B> out.json     (two "full" runs against each other)

`run` leaves stride-8 samples of every snapshot in /tmp/rpde_golden; `combine` writes the committed
file: stride-64 samples (65 x 65 per field) and full-field norms of the "full" run, and the relative L2
difference full-vs-parity per snapshot and field.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

RA, PR, DT = 1e8, 1.0, 2e-4
SNAPS = (1, 2, 4, 10, 20, 50, 100, 150, 200)
# a longer run: RPDE_GOLDEN_SNAPS="1,2,4,10,20,50,100,150,200,300,400,500,600,700,800" (round 4: the 4097 golden past the
# step where the pressure of an independent eigenbasis falls below 1e-10; only the "full" run is extended -- snapshots
# without a "parity" partner carry full_vs_parity = NaN and the readers apply the plain 1e-10 bar to them)
if os.environ.get("RPDE_GOLDEN_SNAPS"):
    SNAPS = tuple(int(v) for v in os.environ["RPDE_GOLDEN_SNAPS"].split(","))
TMP = os.environ.get("RPDE_GOLDEN_TMP", "/tmp/rpde_golden")
FIELDS = ("velx", "vely", "temp", "pres")


def stride_for(n):
    return max(1, (n - 1) // 64)


def run(n, mode):
    from oracle import navier as N
    os.makedirs(TMP, exist_ok=True)
    t0 = time.time()
    nav = N.Navier2D.new_confined(n, n, RA, PR, DT, 1.0, "rbc", eig_mode=mode)
    print(f"setup {time.time() - t0:.1f} s", flush=True)
    nav.set_velocity(0.2, 1.0, 1.0)
    nav.set_temperature(0.2, 1.0, 1.0)
    t0 = time.time()
    for s in range(1, max(SNAPS) + 1):
        nav.update()
        if s in SNAPS:
            f = nav.physical_fields()
            out = {}
            for k, v in f.items():
                out[k] = v[::8, ::8].copy() if n > 1025 else v.copy()
                out[k + "_norm"] = np.array(np.linalg.norm(v))
            out["div_norm"] = np.array(nav.div_norm())
            np.savez(os.path.join(TMP, f"{n}_{mode}_{s}.npz"), **out)
            print(s, f"{time.time() - t0:.0f} s", {k: float(np.abs(v).max()) for k, v in f.items()}, flush=True)


def combine(n):
    st = stride_for(n) // (8 if n > 1025 else 1)
    out = dict(nx=n, ny=n, ra=RA, pr=PR, dt=DT, stride=stride_for(n), snaps=np.array(SNAPS))
    for s in SNAPS:
        fa, fb = os.path.join(TMP, f"{n}_full_{s}.npz"), os.path.join(TMP, f"{n}_parity_{s}.npz")
        if not os.path.exists(fa):      # a partial golden: the readers skip missing snapshots
            continue
        a = np.load(fa)
        b = np.load(fb) if os.path.exists(fb) else None
        for k in FIELDS:
            out[f"{k}_{s}"] = a[k][::st, ::st].copy()
            out[f"{k}_{s}_norm"] = a[k + "_norm"]
            out[f"{k}_{s}_full_vs_parity"] = np.array(np.linalg.norm(a[k] - b[k]) / np.linalg.norm(a[k]) if b is not None else np.nan)
        out[f"div_norm_{s}"] = a["div_norm"]
        print(s, {k: float(out[f"{k}_{s}_full_vs_parity"]) for k in FIELDS})
    # RPDE_GOLDEN_OUT: another file name (the extended 4097 run of round 4 is committed NEXT to the golden the tests read:
    # headline_4097_full_extended.npz -- its snapshots behind step 200 have no "parity" partner and no engine comparison yet)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("RPDE_GOLDEN_OUT", f"headline_{n}_full.npz"))
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


def compare(n, dir_a, dir_b, out):
    """Two runs of the SAME oracle setup (mode "full": the reference's one dgeev of the whole operator) in two processes --
    e.g. with different BLAS thread counts: relative L2 of their stride-8 samples per snapshot.  dgeev's output is not
    reproducible across thread counts, and the Poisson solve amplifies its round-off by 1e10 (poisson.rs:84-87): this is how
    far the REFERENCE ALGORITHM sits from itself."""
    import json
    rows = []
    for s in SNAPS:
        fa, fb = os.path.join(dir_a, f"{n}_full_{s}.npz"), os.path.join(dir_b, f"{n}_full_{s}.npz")
        if not (os.path.exists(fa) and os.path.exists(fb)):
            continue
        a, b = np.load(fa), np.load(fb)
        rows.append({"steps": s, "rel_l2": {k: float(np.linalg.norm(a[k] - b[k]) / np.linalg.norm(a[k])) for k in FIELDS}})
        print(s, {k: f"{v:.2e}" for k, v in rows[-1]["rel_l2"].items()})
    json.dump({"n": n, "ra": RA, "dt": DT, "what": "oracle eig_mode=full, run A vs run B (two processes, different BLAS thread counts)",
               "snapshots": rows}, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == "compare":      # compare <n> <dir of run A> <dir of run B> <out.json>
        compare(int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5])
    else:
        combine(int(sys.argv[2]))
