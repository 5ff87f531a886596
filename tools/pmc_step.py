#!/usr/bin/env python
"""Run K steps of the bench workload launch by launch (no hipGraph) so that rocprofv3 --pmc sees one
dispatch per launch of the step; writes the step's schedule (tags in issue order) as JSON.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o step -- \
        python tools/pmc_step.py --out $OUT/schedule.json
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o step -- \
        python tools/pmc_step.py --out $OUT/schedule.json
    python tools/pmc_traffic.py --fetch $OUT/fetch --write $OUT/write --schedule $OUT/schedule.json \
        --out profiles/r01_pmc_traffic.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--nx", type=int, default=4097)
p.add_argument("--ny", type=int, default=4097)
p.add_argument("--ra", type=float, default=1e8)
p.add_argument("--dt", type=float, default=2e-4)
p.add_argument("--steps", type=int, default=3)
p.add_argument("--periodic", action="store_true")
p.add_argument("--out", required=True)
a = p.parse_args()
ctor = R.Navier2D.new_periodic if a.periodic else R.Navier2D.new_confined
nav = ctor(a.nx, a.ny, a.ra, 1.0, a.dt, 1.0, "rbc")
nav.set_velocity(0.2, 1.0, 1.0)
nav.set_temperature(0.2, 1.0, 1.0)
nav.profile(1)                       # warm-up step, launch by launch
rows = nav.profile(a.steps)          # the LAST steps * len(schedule) dispatches of the process
sched = nav.schedule()
json.dump({"workload": f"{'periodic' if a.periodic else 'confined'} {a.nx}x{a.ny}", "steps": a.steps,
           "schedule": [{"tag": t, "bytes": b, "flops": f, "dispatches": n, "kind": kind} for t, b, f, n, kind in sched],
           "event_ms": {r["tag"]: r["ms_total"] / r["launches"] for r in rows}}, open(a.out, "w"))
