#!/bin/bash
# unpredicated, batched global loads in the line programs: per-launch table, parity, bench
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02q
rm -rf $O; mkdir -p $O
timeout 200 python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
timeout 200 python tools/trace_ops.py 4097 4097 "S2 y: velx" "conv_velx" "S3 x: rhs + hholtz-x velx" "S9" > $O/trace.txt 2>&1; cat $O/trace.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not headline and not config5" 2>&1 | tail -4
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('4097:', d['value'], d['ms_per_step'], d.get('parity'))"
