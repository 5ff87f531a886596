#!/bin/bash
# round 4, call 12: periodic step with the y-correction as a column scan (C7, as confined) instead of S7 + two transposes; new A/B tests
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04l; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "periodic or equal_line_programs_1025" 2>&1 | tail -4
python tools/profile_step.py 4096 1025 periodic > $O/p3.txt 2>&1; cat $O/p3.txt
python bench.py --no-cpu-baseline --periodic --nx 4096 --ny 1025 --steps 100 > $O/bench_periodic.json 2>$O/bench_periodic.err; head -c 300 $O/bench_periodic.json; echo; tail -2 $O/bench_periodic.err
python bench.py --no-cpu-baseline --periodic --nx 16384 --ny 2049 --aspect 8 --ra 1e9 --dt 1e-4 --steps 30 > $O/bench_config5.json 2>$O/bench_config5.err; head -c 300 $O/bench_config5.json; echo
python bench.py --no-cpu-baseline --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2>$O/bench_1025.err; head -c 300 $O/bench_1025.json; echo; tail -2 $O/bench_1025.err
