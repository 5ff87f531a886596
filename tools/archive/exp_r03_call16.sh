#!/bin/bash
# round 3, call 16: 1025^2 -- the whole-line launches of a stage's three fields as one launch (LineBatch), A/B on one box
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03p
rm -rf $O; mkdir -p $O
(timeout 250 python -m pytest tests/test_gpu_parity.py tests/test_hc.py -m gpu -q -x -k "config2 or hc_step_1025 or test_confined_257" 2>&1 | tail -4) | tee $O/pytest.txt
for v in 1 0 1 0; do
  RPDE_LINE_BATCH=$v timeout 200 python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 1000 --warmup 200 > $O/bench_1025_batch$v.json 2>/dev/null
  python -c "import json; d = json.loads(open('$O/bench_1025_batch$v.json').read().strip().splitlines()[-1]); print('1025^2 batch=$v', round(d['value'], 1), 'steps/s', round(d['ms_per_step'], 4), 'ms')"
done
timeout 120 python tools/profile_step.py 1025 1025 > $O/profile_1025.txt 2>&1; cut -c1-110 $O/profile_1025.txt
