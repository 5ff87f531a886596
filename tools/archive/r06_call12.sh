#!/bin/bash
# Round 6, GPU call 12: lift structure (analyse_lift): bit-identity, A/B timing per launch and per step, configs 2 / 3
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06o
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lift_structure or periodic_step or confined_step or config2" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do for w in 0 1; do
  RPDE_LIFT_STRUCT=$w timeout 200 python tools/profile_step.py 2>&1 | grep -E "^S2|^S3|^total" | sed "s/^/lift_struct=$w rep=$rep  /" >> $O/ab_lift.txt
done; done
cat $O/ab_lift.txt
(RPDE_LIFT_STRUCT=0 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py; RPDE_LIFT_STRUCT=0 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py) | tee $O/ab_step.txt
for w in 0 1; do
  RPDE_LIFT_STRUCT=$w timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | sed "s/^/config3 lift_struct=$w /" | tee -a $O/bench_cfg.txt
  RPDE_LIFT_STRUCT=$w timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 | sed "s/^/config2 lift_struct=$w /" | tee -a $O/bench_cfg.txt
done
