#!/bin/bash
# Round 6, GPU call 5: the periodic step with element-wise S5 / S8 / S9 and the whole-line S6: A/B tests, configs 3 and 5 against
# the line programs in one call; then the SURVEY 8f-4 solvers again (tools/r06_call4.sh) with the fused gradient+backward.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06f
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "periodic" 2>&1 | grep -v "Gloo" | tail -6 > $O/pytest_periodic.txt; cat $O/pytest_periodic.txt
for cfg in "--periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200" "--periodic --nx 16384 --ny 2049 --ra 1e9 --dt 1e-4 --aspect 8 --steps 30"; do
  for sw in "" "RPDE_PER_ROWS=0" "RPDE_S6_LINE=0" "RPDE_PER_ROWS=0 RPDE_S6_LINE=0"; do
    echo "== $cfg [$sw]" >> $O/ab_periodic.txt
    env $sw timeout 300 python bench.py $cfg --no-cpu-baseline 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms')" >> $O/ab_periodic.txt
  done
done
cat $O/ab_periodic.txt
python tools/profile_step.py 4096 1025 periodic > $O/profile_step_config3.txt 2>&1; cat $O/profile_step_config3.txt
python tools/profile_step.py 16384 2049 periodic > $O/profile_step_config5.txt 2>&1; tail -25 $O/profile_step_config5.txt
RPDE_CALL_DIR=r06e bash tools/r06_call4.sh
