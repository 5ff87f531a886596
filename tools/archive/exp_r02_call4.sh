#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02d
rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "confined_step or periodic_step or confined_257 or prandtl" 2>&1 | tail -3
python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
python bench.py --no-cpu-baseline > $O/bench_nocpu.json 2>$O/bench.err; python -c "
import json; d=json.loads([l for l in open('$O/bench_nocpu.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_update_plus_exit'])"
