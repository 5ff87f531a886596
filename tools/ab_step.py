#!/usr/bin/env python
"""Device time per step of the bench workload through the hipGraph replay, for whatever RPDE_* switches the environment
sets (they are read once per process): the A/B measurements of one gpurun call run this once per side.

    RPDE_LINE_BATCH=15 python tools/ab_step.py [nx ny [steps [repeats]]]        -> one line: label, ms per step (min / median of the repeats)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
ny = int(sys.argv[2]) if len(sys.argv) > 2 else 4097
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
nav = R.Navier2D.new_confined(nx, ny, 1e8, 1.0, 2e-4, 1.0, "rbc")
nav.set_velocity(0.2, 1.0, 1.0)
nav.set_temperature(0.2, 1.0, 1.0)
nav.update(10)
ms = []
for _ in range(reps):
    nav.update(steps)
    ms.append(nav.last_update_ms() / steps)
ms.sort()
label = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("RPDE_") and k not in ("RPDE_EIG_CACHE", "RPDE_EVIDENCE_SHORT")) or "default"
print(f"AB {label:40s} {nx}x{ny}  min {ms[0]:.4f}  median {ms[len(ms) // 2]:.4f} ms/step  ({len(nav.schedule())} launches per step)", flush=True)
