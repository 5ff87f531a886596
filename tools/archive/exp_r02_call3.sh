#!/bin/bash
# Round 2, GPU call 3: column scans + GEMM fast path: bench, per-launch profile, selected parity tests.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02c
rm -rf $O; mkdir -p $O
for w in "gemm_nt 2048 4095 50" "gemm_nn 2048 4095 50" "transpose 4097 4097 50"; do python tools/microbench_one.py $w >> $O/mb.txt 2>&1; done
cat $O/mb.txt
python tools/profile_step.py > $O/profile_step.txt 2>&1; cat $O/profile_step.txt
python tools/profile_step.py 4096 1025 periodic > $O/profile_step_periodic.txt 2>&1; tail -12 $O/profile_step_periodic.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "bench_sizes or headline or confined_step or periodic_step or exit_flag or prandtl or confined_257 or config2 or mfma_gemm or config3_first" 2>&1 | tail -15 > $O/pytest_sel.txt
cat $O/pytest_sel.txt
