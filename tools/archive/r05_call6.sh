#!/bin/bash
# round 5, GPU call 6: the blocked PdmaPlus2 column solve of "hc" (pdma.h) -- parity tests, then the 4097^2 hc bench A/B on one box
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r05g; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_hc.py -m gpu -x -q > $O/pytest_hc.txt 2>&1; echo "pytest hc exit $?"; tail -3 $O/pytest_hc.txt
for v in 1 0 1 0; do
  RPDE_HC_BLOCKED=$v timeout 400 python bench.py --bc hc --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_hc_blocked$v.json 2> $O/bench_hc_blocked$v.err
  echo "== RPDE_HC_BLOCKED=$v"; python - <<PY
import json
for l in open("$O/bench_hc_blocked$v.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"])
PY
done
