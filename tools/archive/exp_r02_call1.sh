#!/bin/bash
# Round 2, GPU call 1: microbenchmarks (line programs, GEMM, MFMA peak + clocks, transposes),
# bench.py with the parity object, the full GPU test suite.  Writes gpurun_out/r02a/.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02a
rm -rf $O; mkdir -p $O
python -c "import torch" 2>/dev/null   # page the image in once
echo "== microbench (line programs)" > $O/mb.txt
python tools/microbench.py 4097 4096 >> $O/mb.txt 2>&1
echo "== gemm / mfma peak / transposes" >> $O/mb.txt
for w in "gemm_nt 2048 2048 100" "gemm_nn 2048 2048 100" "gemm_nt 2048 4095 50" "gemm_nn 2048 4095 50" "mfma_peak 1024 20000 5" "mfma_peak 512 20000 5" "transpose 4097 4097 50"; do
  python tools/microbench_one.py $w >> $O/mb.txt 2>&1
done
RPDE_TP_TILE=32 python tools/microbench_one.py transpose 4097 4097 50 >> $O/mb.txt 2>&1
# shader clock / power while the f64 MFMA GEMM runs (and while the register-only MFMA loop runs)
(python tools/microbench_one.py gemm_nt 2048 4095 8000 > $O/gemm_long.txt 2>&1) &
GP=$!
sleep 2.0
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -Ei "sclk|mclk|fclk|power" | head -6 >> $O/clocks_gemm.txt; echo "--" >> $O/clocks_gemm.txt; sleep 0.4; done
wait $GP
(python tools/microbench_one.py mfma_peak 1024 20000 300 > $O/mfma_long.txt 2>&1) &
GP=$!
sleep 2.0
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -Ei "sclk|mclk|fclk|power" | head -6 >> $O/clocks_mfma.txt; echo "--" >> $O/clocks_mfma.txt; sleep 0.4; done
wait $GP
rocm-smi --showclocks --showpower > $O/clocks_idle.txt 2>&1
cat $O/mb.txt $O/gemm_long.txt $O/mfma_long.txt
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
timeout 1500 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -40 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
