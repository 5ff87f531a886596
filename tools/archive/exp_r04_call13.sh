#!/bin/bash
# round 4, call 13: Fourier whole-line kernels for lines of 16384 / 8192 reals (config 5): parity, A/B against the line programs
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04m; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config5 or periodic_fourier" 2>&1 | tail -4
for cfg in "new:X=1" "lineprog:RPDE_S1_LINE=0 RPDE_S3_LINE=0"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 16384 2049 periodic > $O/p5_$name.txt 2>&1
  echo "== config 5 $name"; grep "S1 x\|S3 x\|total" $O/p5_$name.txt
  env $envs python tools/profile_step.py 8192 1025 periodic > $O/p8k_$name.txt 2>&1
  echo "== 8192 x 1025 $name"; grep "S1 x\|S3 x\|total" $O/p8k_$name.txt
done
python bench.py --no-cpu-baseline --periodic --nx 16384 --ny 2049 --aspect 8 --ra 1e9 --dt 1e-4 --steps 30 > $O/bench_config5.json 2>$O/bench_config5.err; head -c 300 $O/bench_config5.json; echo
