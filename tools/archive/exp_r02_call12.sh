#!/bin/bash
# GEMM variants: timing on the Poisson shapes, then parity with the pipelined variant
export TMPDIR=/tmp
for v in 0 4; do echo "== variant $v"; RPDE_GEMM_VARIANT=$v timeout 200 python tools/gemm_shapes.py; done
RPDE_GEMM_VARIANT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "solvers or confined_257 or headline or eigenbasis" 2>&1 | tail -3
