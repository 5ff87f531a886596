#!/bin/bash
# round 3, call 10: column scans across ranks on the HIP build (ranks share the GPU), batched transposes, carry kernels
# with small workgroups, 1025^2
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03j
rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python tools/profile_step.py > $O/profile_$name.txt 2>&1; echo "--- $name: $*"; grep -E "$PAT" $O/profile_$name.txt | cut -c1-100; }
PAT="." run default RPDE_X=0
PAT="^C4|^C7|^C10|^T1|^T2|^total" run carry256_nobatch RPDE_COL_CARRY_T=256 RPDE_TP_BATCH=0
for b in 1 0; do RPDE_TP_BATCH=$b timeout 200 python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('1025^2 batch=$b', d['value'], 'steps/s', d['ms_per_step'], 'ms')"; done
timeout 900 python -m pytest tests/test_sharded.py tests/test_gpu_parity.py -m gpu -q -x -k "sharded or headline_config or step_parity or config2" 2>&1 | tail -6 | tee $O/pytest.txt
