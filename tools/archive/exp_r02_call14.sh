#!/bin/bash
# GEMM: VGPR accumulators, clamped edge tiles, even + odd in one launch; variants 0 / 4
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02l
rm -rf $O; mkdir -p $O
for v in 0 4; do echo "== variant $v"; RPDE_GEMM_VARIANT=$v timeout 200 python tools/gemm_shapes.py 2>&1 | head -6
RPDE_GEMM_VARIANT=$v timeout 200 python tools/profile_step.py > $O/profile_step_v$v.txt 2>&1; grep -E "^G|total" $O/profile_step_v$v.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "solvers or confined_257 or headline or eigenbasis or confined_step or prandtl" 2>&1 | tail -3
RPDE_GEMM_VARIANT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "solvers or confined_257 or headline" 2>&1 | tail -3
