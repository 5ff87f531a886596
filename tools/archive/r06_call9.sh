#!/bin/bash
# Round 6, GPU call 9: the GEMM's 128-tile by eight waves (four waves per SIMD): bit-identity, A/B timing per kernel and per step, MFMA counters
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06j
rm -rf $O; mkdir -p $O
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
export RPDE_TOOLS_SPECTRUM=/tmp/spec4097.npy
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "eight_waves or mfma_gemm" 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do for w in 4 8; do
  RPDE_GEMM_WAVES=$w timeout 200 python tools/profile_step.py 2>&1 | grep -E "^G1|^G2|^total" | sed "s/^/waves=$w rep=$rep  /" >> $O/ab_gemm_waves.txt
done; done
cat $O/ab_gemm_waves.txt
(timeout 120 python tools/ab_step.py; RPDE_GEMM_WAVES=8 timeout 120 python tools/ab_step.py; timeout 120 python tools/ab_step.py; RPDE_GEMM_WAVES=8 timeout 120 python tools/ab_step.py) | tee $O/ab_step.txt
cd /tmp
RPDE_GEMM_WAVES=8 timeout 300 rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/mfma -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_mfma.log 2>&1
cd $R
python tools/pmc_counters.py $O/mfma $O/schedule.json 2>&1 | grep -E "tag|G1|G2" | tee $O/mfma_counters_waves8.txt
rm -f $O/*/*.db $O/*/*/*.db; find $O -name '*.csv' -size +4M -delete
