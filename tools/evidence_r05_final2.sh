#!/bin/bash
# Round-5 evidence on the FINAL sources in ONE gpurun call (about twenty GPU-minutes were left when this ran): ordered by
# importance, every step under its own timeout.  The full GPU test suite is NOT part of it (the driver runs it at round end;
# the suite's last full run of the round is profiles/r05_pytest_gpu_full_214f117.txt, the tests of everything added since ran in
# calls 4 - 7): here the A/B tests of what this session added, the A/B timings of its switches, then the headline evidence --
# PMC traffic (the file bench.py reads roofline.traffic from, and whose source hash tests/test_layout.py pins), the bench line,
# the rocprofv3 kernel trace + stats of the bench command, SQ and LDS counters.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05g
rm -rf $O; mkdir -p $O
cat $R/.evidence_commit > $O/commit.txt 2>/dev/null
export EVIDENCE_COMMIT=$(cat $O/commit.txt 2>/dev/null)
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE      # setup data only (csrc/hostmath.cc); bench.py runs without it below
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s" | tee -a $O/laps.txt; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
# 1. the A/B tests of this session's additions: 4097-point batches, S6 / S9 as whole-line kernels, against the forms they replace
[ -z "$RPDE_EVIDENCE_SHORT" ] && (timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --durations=5 \
   -k "(round5_ab_switches and (LINE_BATCH or GEMM_LDS or S6_KEEP)) or (whole_line_stage_equals_line_program_4097 and (S3 or S6 or S9 or S1 or CONV)) or whole_line_kernels_equal_line_programs_1025 or mfma_gemm or conv_line_4097 or dct_line" 2>&1 \
   | grep -v "socket.cpp\|amdgpu.ids\|Gloo\] Rank" | tail -40) > $O/pytest_gpu_new.txt
lap "new A/B tests"
# 2. A/B timings, hipGraph replay of the bench workload: default, one launch per field, S6 / S9 as line programs, everything of this session off
(timeout 120 python tools/ab_step.py; RPDE_S6_KEEP=0 timeout 120 python tools/ab_step.py; RPDE_GEMM_LDS=0 timeout 120 python tools/ab_step.py;
 RPDE_LINE_BATCH=15 timeout 120 python tools/ab_step.py; RPDE_S6_LINE=0 timeout 120 python tools/ab_step.py; RPDE_S9_LINE=0 timeout 120 python tools/ab_step.py;
 RPDE_GEMM_LDS=0 RPDE_LINE_BATCH=15 RPDE_S6_LINE=0 RPDE_S9_LINE=0 timeout 120 python tools/ab_step.py;
 timeout 120 python tools/ab_step.py) > $O/ab_step.txt 2>$O/ab_step.err
[ -z "$RPDE_EVIDENCE_SHORT" ] && for rep in 1 2; do for keep in 1 0; do
  RPDE_S6_KEEP=$keep timeout 120 python tools/profile_step.py 2>&1 | grep -E "^S6|^total" | sed "s/^/keep=$keep rep=$rep  /" >> $O/ab_s6_keep.txt
done; done
lap "A/B timings"
# 2b. keep the faster form of each as the DEFAULT before any evidence is collected: a form that loses by more than 0.3 % against
# the mean of the two default runs has its default flipped in engine.cc (one greppable line each) and the library rebuilt here,
# so that the PMC file's source hash, the bench line and the trace all belong to the sources that get committed
python - "$O" <<'PY' > $O/defaults.txt 2>&1
import re, subprocess, sys
O = sys.argv[1]
rows = {}
for l in open(O + "/ab_step.txt"):
    m = re.match(r"AB (.*?)\s+\d+x\d+\s+min ([0-9.]+)\s+median ([0-9.]+)", l)
    if m: rows.setdefault(m.group(1).strip(), []).append(float(m.group(2)))
print(rows)
base = sum(rows["default"]) / len(rows["default"])
flips = []
E, G = "rustpde_mpi_amd/csrc/engine.cc", "rustpde_mpi_amd/csrc/gemm.cc"
for label, path, old, new in (("RPDE_LINE_BATCH=15", E, "constexpr int kLineBatchDefaultMask = 255;", "constexpr int kLineBatchDefaultMask = 15;"),
                              ("RPDE_S6_LINE=0", E, "constexpr bool kS6LineDefault = true;", "constexpr bool kS6LineDefault = false;"),
                              ("RPDE_S9_LINE=0", E, "constexpr bool kS9LineDefault = true;", "constexpr bool kS9LineDefault = false;"),
                              ("RPDE_S6_KEEP=0", E, "constexpr int kS6KeepDefault = 1;", "constexpr int kS6KeepDefault = 0;")):   # (the GEMM's LDS layout is not decided here: equal per kernel in call 10, and this A/B cannot resolve 0.5 %)
    import os
    failed = os.path.exists(O + "/pytest_gpu_new.txt") and any(l.startswith("FAILED") and label.split("=")[0].replace("RPDE_", "") in l for l in open(O + "/pytest_gpu_new.txt"))
    if failed: print(label, "its A/B test FAILED on this box: the new form is switched off")
    if failed or (label in rows and min(rows[label]) < 0.997 * base):
        src = open(path).read()
        assert old in src
        open(path, "w").write(src.replace(old, new, 1))
        flips.append(label)
    print(label, rows.get(label), "vs default", base, "-> FLIP" if label in flips else "-> keep")
if flips:
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], capture_output=True, text=True)
    print("rebuild rc", r.returncode, r.stderr[-500:])
print("FLIPS", flips)
PY
cp rustpde_mpi_amd/csrc/engine.cc $O/engine.cc.final; cp rustpde_mpi_amd/csrc/gemm.cc $O/gemm.cc.final
(timeout 120 python tools/ab_step.py) >> $O/ab_step.txt 2>>$O/ab_step.err      # the defaults that go into the evidence
lap "defaults decided"
cd /tmp
# 3. PMC traffic on the final sources (separate passes, as the guide prescribes)
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch $O/fetch --write $O/write --schedule $O/schedule.json --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json     # bench.py picks roofline.traffic up from here
lap "PMC traffic"
# 4. the bench line (no eig cache: the command the driver runs)
(unset RPDE_EIG_CACHE; timeout 420 python bench.py > $O/bench.json 2> $O/bench.err)
lap "bench"
# 5. rocprofv3 kernel trace + stats of the bench command
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
python tools/trace_by_tag.py $O/trace $O/schedule.json $O/trace_by_tag.csv 2> $O/trace_by_tag.log
lap "rocprofv3 stats"
# 6. SQ counters per launch (instruction mix), then the LDS counters (what the whole-line kernels' LDS traffic costs)
cd /tmp
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_sq.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/lds -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_lds.log 2>&1
cd $R
python tools/pmc_counters.py $O/sq $O/schedule.json > $O/sq_counters.txt 2>&1
python tools/pmc_counters.py $O/lds $O/schedule.json > $O/lds_counters.txt 2>&1
lap "SQ / LDS counters"
# 7. per-launch HIP-event table, kernel resources
timeout 200 python tools/profile_step.py > $O/profile_step.txt 2>&1
bash tools/kernel_resources.sh > $O/kernel_resources.txt 2>/dev/null
lap "profile_step"
# 8. config 2 (1025^2: S6 / S9 are new kernels there too): A/B timing and its bench line with the parity legs; the "hc" line
(timeout 60 python tools/ab_step.py 1025 1025 300 5; RPDE_S6_LINE=0 RPDE_S9_LINE=0 timeout 60 python tools/ab_step.py 1025 1025 300 5) > $O/ab_step_1025.txt 2>&1
timeout 300 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2> $O/bench_1025.err
lap "config 2"
timeout 200 python bench.py --no-cpu-baseline --bc hc --steps 30 > $O/bench_hc.json 2> $O/bench_hc.err
lap "hc"
# 9. more of the GPU suite on these sources, as far as the call's budget goes (the driver runs all of it at round end): the "hc" step,
# the arena checks, the pencil-sharded step with ranks sharing the GPU, the adjoint / LNSE / NonLin parity at the small sizes
[ -z "$RPDE_EVIDENCE_SHORT" ] && (timeout 420 python -m pytest tests/test_hc.py tests/test_arena.py tests/test_sharded.py tests/test_adjoint.py tests/test_c_host.py -m gpu -q -x --durations=5 \
   -k "not 4097 and not 2049" 2>&1 | grep -v "socket.cpp\|amdgpu.ids\|Gloo\] Rank\|mean.h5" | tail -25) > $O/pytest_gpu_more.txt
lap "more GPU tests"
rm -f $O/*/*.db $O/*/*/*.db
find $O -name '*kernel_trace.csv' -size +8M -delete
rm -f $O/fetch/*counter_collection.csv $O/write/*counter_collection.csv $O/sq/*counter_collection.csv $O/lds/*counter_collection.csv $O/*/*/*counter_collection.csv
tail -12 $O/pytest_gpu_new.txt | cut -c1-200; tail -6 $O/pytest_gpu_more.txt | cut -c1-200; cat $O/ab_step.txt; tail -c 1800 $O/bench.json; cat $O/trace_by_tag.log; head -32 $O/pmc_traffic.txt; cat $O/laps.txt
