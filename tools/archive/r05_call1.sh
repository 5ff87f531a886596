#!/bin/bash
# round 5, GPU call 1: the new parity test (800-step extended golden), the arena tests, smoke, the bench line with the
# extended independent-golden comparison
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_arena.py tests/test_gpu_parity.py -m gpu -x -q -s \
  -k "arena or guard or mapping or extended or headline_config_4097 or col" > $O/pytest.txt 2>&1
echo "pytest rc $?" >> $O/pytest.txt
tail -25 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc $?"; tail -3 $O/bench.err; cut -c1-400 $O/bench.json
