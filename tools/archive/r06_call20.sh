#!/bin/bash
# Round 6, GPU call 20: RPDE_FORK=1 (C7 -> S8 beside S9 -> C10 on two graph branches): bit-identity tests, step times at configs 2 / 3 and 4097^2
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06w
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "forked_tail" > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
for rep in 1 2 3; do for f in 0 1; do
  RPDE_FORK=$f timeout 120 python tools/ab_step.py 1025 1025 200 5 | sed "s/^/fork=$f /"
  RPDE_FORK=$f timeout 120 python tools/ab_step.py 513 513 200 5 | sed "s/^/fork=$f /"
  RPDE_FORK=$f timeout 120 python tools/ab_step.py 2049 2049 100 5 | sed "s/^/fork=$f /"
  RPDE_FORK=$f timeout 120 python tools/ab_step.py | sed "s/^/fork=$f /"
done; done | tee $O/ab_fork.txt
for f in 0 1 0 1; do
  RPDE_FORK=$f timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config3 fork=$f /" | tee -a $O/bench_cfg.txt
  RPDE_FORK=$f timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120 | sed "s/^/config2 fork=$f /" | tee -a $O/bench_cfg.txt
done
