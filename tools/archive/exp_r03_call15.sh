#!/bin/bash
# round 3, call 15: the convection term of 1025-point lines on the half-length core (one wave per line), kernel trace of
# the 1025^2 step, "hc" sharded on the HIP build
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03o
rm -rf $O; mkdir -p $O
(timeout 250 python -m pytest tests/test_gpu_parity.py tests/test_hc.py -m gpu -q -x -k "config2 or independent_reference_setup and 1025 or hc_step_1025" 2>&1 | tail -4) | tee $O/pytest.txt
timeout 120 python tools/profile_step.py 1025 1025 > $O/profile_1025.txt 2>&1; cut -c1-100 $O/profile_1025.txt
RPDE_CONV_LINE=0 timeout 120 python tools/profile_step.py 1025 1025 > $O/profile_1025_convprog.txt 2>&1; grep -E "conv|^total" $O/profile_1025_convprog.txt | cut -c1-100
R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b1025 -- python $R/bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 400 > $O/bench_1025.json 2> $O/bench_1025.err
cd $R
python -c "import json; d = json.loads(open('$O/bench_1025.json').read().strip().splitlines()[-1]); print('1025^2', d['value'], 'steps/s', d['ms_per_step'], 'ms')"
head -30 $O/trace/*kernel_stats.csv | cut -c1-160
rm -f $O/trace/*.db $O/trace/*/*.db $O/trace/*kernel_trace.csv
(timeout 250 python -m pytest tests/test_sharded.py -m gpu -q -x -k "oracle_hip and 2" 2>&1 | tail -4) | tee $O/pytest_sharded.txt
