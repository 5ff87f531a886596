#!/bin/bash
# small-tile GEMM: shapes, the 1025^2 step, parity incl. the sharded GPU tests and the config-2 golden run
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02m
rm -rf $O; mkdir -p $O
timeout 200 python tools/gemm_shapes.py 2>&1
timeout 200 python tools/profile_step.py 1025 1025 > $O/profile_step_1025.txt 2>&1; cat $O/profile_step_1025.txt | grep -E "^G|total|S1|C4 y: hh"
python bench.py --nx 1025 --ny 1025 --no-cpu-baseline > $O/bench_1025.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_1025.json')); print('1025:', d['value'], d['ms_per_step'])"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -q -x -k "solvers or confined_257 or config2 or sharded or world or geometry or confined_step" 2>&1 | tail -4
