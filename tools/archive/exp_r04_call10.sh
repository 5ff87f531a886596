#!/bin/bash
# round 4, call 10: periodic step with whole-line Fourier kernels (rfft_line.h: S1 pair, S3 rhs) -- parity, A/B against the line
# programs at config 3 (4096 x 1025); convection term of 1025-point lines at 1 / 2 / 3 waves per SIMD; S3 phase trace at 4097
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04j; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "periodic" 2>&1 | tail -4
for cfg in "new:X=1" "lineprog:RPDE_S1_LINE=0 RPDE_S3_LINE=0" "w2:RPDE_CONV1K_WPC=2" "w3:RPDE_CONV1K_WPC=3"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 4096 1025 periodic > $O/p3_$name.txt 2>&1
  echo "== config 3 $name"; grep "S1 x\|S3 x\|conv\|total" $O/p3_$name.txt
done
for cfg in "w1:X=1" "w2:RPDE_CONV1K_WPC=2" "w3:RPDE_CONV1K_WPC=3"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs python tools/profile_step.py 1025 1025 > $O/p1025_$name.txt 2>&1
  echo "== 1025 $name"; grep "conv\|total" $O/p1025_$name.txt
done
python bench.py --no-cpu-baseline --periodic --nx 4096 --ny 1025 --steps 100 > $O/bench_periodic.json 2>$O/bench_periodic.err; head -c 350 $O/bench_periodic.json; echo; tail -2 $O/bench_periodic.err
python bench.py --no-cpu-baseline --periodic --nx 16384 --ny 2049 --aspect 8 --ra 1e9 --dt 1e-4 --steps 30 > $O/bench_config5.json 2>$O/bench_config5.err; head -c 350 $O/bench_config5.json; echo
python tools/trace_phases.py 4097 4097 "S3 x: rhs + hholtz-x velx" "S3 x: rhs + hholtz-x vely" > $O/trace_s3.txt 2>&1; cat $O/trace_s3.txt
