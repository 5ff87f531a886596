#!/bin/bash
# pair DCT (OP_DCT2) on / off: parity first, then the per-launch table and the bench line
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02g
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or periodic_step or confined_257 or prandtl or headline" 2>&1 | tail -3
python tools/profile_step.py > $O/profile_step_pair.txt 2>&1; cat $O/profile_step_pair.txt
RPDE_DCT_PAIR=0 python tools/profile_step.py > $O/profile_step_single.txt 2>&1; grep -E "S1|S2 y|conv|total|TOTAL" $O/profile_step_single.txt
python bench.py --no-cpu-baseline > $O/bench_pair.json 2> $O/bench_pair.err; cat $O/bench_pair.json
