#!/bin/bash
# paired loads + split-twiddle prefetch: parity, per-launch table, per-phase trace, bench line
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02j
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or periodic_step or confined_257 or prandtl or headline or exit" 2>&1 | tail -3
timeout 200 python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
timeout 300 python tools/trace_ops.py > $O/trace_phases.txt 2>&1; cat $O/trace_phases.txt
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02j/bench.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_update_plus_exit'], d.get('parity'))
PY
