#!/bin/bash
# Round 6, GPU call 3: A/B of the persistent form of the GEMM pair, its bit-identity test; the "hc" bench line with parity + CPU legs.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06c
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
(timeout 120 python tools/ab_step.py; RPDE_GEMM_PERSIST=1 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py; RPDE_GEMM_PERSIST=1 timeout 120 python tools/ab_step.py) > $O/ab_step.txt 2>$O/ab_step.err
cat $O/ab_step.txt
for p in 0 1; do RPDE_GEMM_PERSIST=$p timeout 120 python tools/profile_step.py 2>&1 | grep -E "^G1|^G2|^total" | sed "s/^/persist=$p  /" >> $O/ab_gemm_persist.txt; done
cat $O/ab_gemm_persist.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gemm_persist or mfma_gemm" 2>&1 | tail -5 > $O/pytest_persist.txt; cat $O/pytest_persist.txt
(unset RPDE_EIG_CACHE; timeout 600 python bench.py --bc hc --steps 30 > $O/bench_hc.json 2> $O/bench_hc.err); tail -c 1500 $O/bench_hc.json; tail -3 $O/bench_hc.err
cp gpurun_out/bench_parity_detail.json $O/ 2>/dev/null
