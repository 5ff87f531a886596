#!/bin/bash
# round 3, call 9: C7 (velocity correction in y as a column scan behind a transposed-store G2; no S7, no T5),
# new defaults (rhs_line at three workgroups per CU, S1 on the half-length core), A/B of S1 at three workgroups per CU
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03i
rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python tools/profile_step.py > $O/profile_$name.txt 2>&1; echo "--- $name: $*"; grep -E "$PAT" $O/profile_$name.txt | cut -c1-100; }
PAT="." run default RPDE_X=0
PAT="^S1|^total" run s1wpc3 RPDE_S1_WPC=3
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_snapshots.py -m gpu -q -x -k "headline or step or parity or sharded_equals or snapshot" 2>&1 | tail -6 | tee $O/pytest.txt
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
