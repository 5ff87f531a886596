#!/bin/bash
# round 3, call 13: S8 as a whole-line kernel (corr_line.h) and the XCD-paired block passes of C7 -- A/B on one box
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03m
rm -rf $O; mkdir -p $O
(timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "RPDE_S8_LINE or headline_config_4097" 2>&1 | tail -4) | tee $O/pytest.txt
timeout 120 python tools/profile_step.py > $O/profile_new.txt 2>&1; grep -E "^C4|^C7|^S8|^S9|^total" $O/profile_new.txt | cut -c1-100
RPDE_S8_LINE=0 RPDE_COL_PAIR=0 timeout 120 python tools/profile_step.py > $O/profile_old.txt 2>&1; grep -E "^C4|^C7|^S8|^S9|^total" $O/profile_old.txt | cut -c1-100
timeout 120 python tools/profile_step.py 1025 1025 > $O/profile_1025.txt 2>&1; grep -E "^C7|^S8|^total" $O/profile_1025.txt | cut -c1-100
