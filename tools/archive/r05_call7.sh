#!/bin/bash
# round 5, GPU call 7: the adjoint / LNSE / NonLin tests of the third and fourth slices of SURVEY 8f-4 and the hc A/B test on the device
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05h; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_adjoint.py tests/test_hc.py -m gpu -q -s --durations=10 > $O/pytest.txt 2>&1; echo "pytest exit $?"
grep -E "^lnse|^nonlin|passed|failed|FAILED|Error" $O/pytest.txt | cut -c1-220 | tail -40
