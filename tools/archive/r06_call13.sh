#!/bin/bash
# Round 6, GPU call 13: S6 with one factor row per line (derived q1 / q2 / r2): parity, A/B timing per launch and per step, configs 2 / 3 / 5
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06p
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "s6_derived or round5_ab_switches or periodic_step or confined_step or config2 or poisson or solvers" 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2; do for w in 0 1; do
  RPDE_S6_DERIVE=$w timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S6|^total" | sed "s/^/s6_derive=$w rep=$rep  /" >> $O/ab_s6.txt
done; done
cat $O/ab_s6.txt
(RPDE_S6_DERIVE=0 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py; RPDE_S6_DERIVE=0 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py) | tee $O/ab_step.txt
for w in 0 1; do
  RPDE_S6_DERIVE=$w timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | sed "s/^/config3 s6_derive=$w /" | tee -a $O/bench_cfg.txt
  RPDE_S6_DERIVE=$w timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | sed "s/^/config2 s6_derive=$w /" | tee -a $O/bench_cfg.txt
done
# the 200-step same-inputs golden at the bench size with the derived factors (parity_shared_basis_golden of the bench line)
(unset RPDE_EIG_CACHE; timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err); python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06p/bench.json") if l.startswith("{")][-1])
print("bench", d["value"], {k: d[k] for k in d if k.startswith("parity")})
PY
