#!/bin/bash
# Round 6, GPU call 8: 2049-point whole-line kernels: tests, config 5 A/B against the line programs in one call
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06i
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "2049 or config5" 2>&1 | grep -v "Gloo" | tail -5 > $O/pytest_2049.txt; cat $O/pytest_2049.txt
for sw in "" "RPDE_DCT_LINE=0" "RPDE_CONV_LINE=0" "RPDE_DCT_LINE=0 RPDE_CONV_LINE=0"; do
  echo "== config 5 [$sw]" >> $O/ab_config5.txt
  env $sw timeout 300 python bench.py --periodic --nx 16384 --ny 2049 --ra 1e9 --dt 1e-4 --aspect 8 --steps 30 --no-cpu-baseline 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms')" >> $O/ab_config5.txt
done
cat $O/ab_config5.txt
python tools/profile_step.py 16384 2049 periodic > $O/profile_step_config5.txt 2>&1; tail -24 $O/profile_step_config5.txt
for sw in "" "RPDE_DCT_LINE=0 RPDE_CONV_LINE=0"; do
  echo "== confined 2049^2 [$sw]" >> $O/ab_2049.txt
  env $sw timeout 300 python bench.py --nx 2049 --ny 2049 --ra 1e8 --dt 5e-4 --steps 100 --no-cpu-baseline 2>> $O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms')" >> $O/ab_2049.txt
done
cat $O/ab_2049.txt; tail -3 $O/ab.err
