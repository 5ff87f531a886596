#!/bin/bash
# DPP wave scans + fused stencil-axpby: parity, per-launch table, bench line
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02k
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or periodic_step or confined_257 or prandtl or headline or exit or solvers or space_ops" 2>&1 | tail -3
timeout 200 python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02k/bench.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_update_plus_exit'])
PY
