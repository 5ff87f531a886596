"""tests/golden/reference_setup_spread.json: how far independent runs of the REFERENCE's own setup sit from each other.

The reference's Poisson setup is one LAPACK dgeev of the whole x operator (src/solver/utils.rs:67-99); its output depends on
the BLAS thread count and CPU, and Poisson::new's -1e-10 shift (poisson.rs:84-87) amplifies the difference by 1e10 during the
first steps.  `make_headline_golden.py run <n> full` was run several times per size with OPENBLAS_NUM_THREADS = 8 / 2 / 1 / 3
(run A = the committed golden headline_<n>_full.npz, B, C, D) plus once with one dgeev per parity block (P).  This script
takes their stride-8 dumps and records, per size, snapshot and field, the LARGEST pairwise relative L2 distance -- the
measured envelope of "a valid implementation of the reference's setup" that tests/bounds.py turns into the bar of the
independent-golden comparison (instead of a global factor on one pair).

    python tests/golden/make_reference_spread.py 1025:/tmp/rpde_golden,/tmp/rpde_golden_B,... 2049:... 4097:...
"""
import itertools
import json
import os
import sys

import numpy as np

FIELDS = ("velx", "vely", "temp", "pres")
SNAPS = (1, 2, 4, 10, 20, 50, 100, 150, 200)
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out_path = os.path.join(HERE, "reference_setup_spread.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for spec in sys.argv[1:]:
        n, dirs = spec.split(":")
        dirs = dirs.split(",")
        runs = {}
        for i, d in enumerate(dirs):
            if os.path.exists(os.path.join(d, f"{n}_full_1.npz")):
                runs["full@" + os.path.basename(d.rstrip("/"))] = (d, "full")
            if i == 0 and os.path.exists(os.path.join(d, f"{n}_parity_1.npz")):
                runs["parity@" + os.path.basename(d.rstrip("/"))] = (d, "parity")
        entry = {"runs": sorted(runs), "snapshots": {}}
        for s in SNAPS:
            data = {}
            for name, (d, mode) in runs.items():
                f = os.path.join(d, f"{n}_{mode}_{s}.npz")
                if os.path.exists(f):
                    data[name] = np.load(f)
            if len(data) < 2:
                continue
            row = {}
            for k in FIELDS:
                dist = [float(np.linalg.norm(data[a][k] - data[b][k]) / np.linalg.norm(data[a][k]))
                        for a, b in itertools.combinations(sorted(data), 2)]
                row[k] = {"max": max(dist), "min": min(dist), "pairs": len(dist)}
            entry["snapshots"][str(s)] = row
            print(n, s, {k: f"{v['min']:.2e}..{v['max']:.2e}" for k, v in row.items()})
        out[str(n)] = entry
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
