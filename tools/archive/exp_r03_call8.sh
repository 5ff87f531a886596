#!/bin/bash
# round 3, call 8: two-pass column Helmholtz (C4), XCD-aware GEMM tile order, A/B of register budget of rhs_line,
# S1 on the half-length core, array skew
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03h
rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python tools/profile_step.py > $O/profile_$name.txt 2>&1; echo "--- $name: $*"; grep -E "^S1|^S2|^S3|^C4|^G1|^G2|^total|Error|error" $O/profile_$name.txt | cut -c1-100; }
run default RPDE_X=0
run noswz RPDE_GEMM_SWIZZLE=0
run rhs3 RPDE_RHS_WPC=3
run hdct3 RPDE_HDCT=3
run skew544 RPDE_ARRAY_SKEW=544
run skew8208 RPDE_ARRAY_SKEW=8208
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "headline or step or solvers or mfma or gemm" 2>&1 | tail -4 | tee $O/pytest.txt
