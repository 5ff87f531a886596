#!/bin/bash
# round 3, call 12: 1025^2 with the one-wave whole-line kernels (N = 1024) and sixteen-block carry batches; graph replay
export TMPDIR=/tmp
O=$PWD/gpurun_out/r03l
rm -rf $O; mkdir -p $O
timeout 200 python tools/profile_step.py 1025 1025 > $O/profile_1025.txt 2>&1; cut -c1-100 $O/profile_1025.txt
RPDE_WHOLE_LINE=0 timeout 200 python tools/profile_step.py 1025 1025 > $O/profile_1025_lineprog.txt 2>&1; grep -E "^S1|^S2|^S3|^total" $O/profile_1025_lineprog.txt | cut -c1-100
timeout 200 python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2>/dev/null; python -c "import json; d = json.load(open('$O/bench_1025.json')); print('1025^2', d['value'], 'steps/s', d['ms_per_step'], 'ms')"
python - <<'PY' 2>&1 | tee $O/graph.txt
import time, rustpde_mpi_amd as R
for n, ra, dt in ((1025, 1e7, 1e-3), (513, 1e6, 2e-3), (4097, 1e8, 2e-4)):
    nav = R.Navier2D.new_confined(n, n, ra, 1.0, dt, 1.0, "rbc")
    nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
    nav.update(20)
    k = 400 if n < 4000 else 50
    t0 = time.time(); nav.update(k); w = time.time() - t0
    print(f"{n}^2 graph replay: {w/k*1e3:.4f} ms/step wall, device {nav.last_update_ms()/k:.4f} ms/step", flush=True)
    del nav
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config2 or headline_independent" 2>&1 | tail -4
timeout 120 python tools/profile_step.py > $O/profile_4097.txt 2>&1; grep -E "^C4|^C7|^C10|^T1|^T2|^total" $O/profile_4097.txt | cut -c1-100
