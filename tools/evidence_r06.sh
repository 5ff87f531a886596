#!/bin/bash
# Round-6 evidence on the FINAL sources in ONE gpurun call, every step under its own timeout: smoke, the A/B timings of the round's
# switches (recorded, nothing is decided here), PMC traffic (the file bench.py reads roofline.traffic from and whose source hash
# tests/test_layout.py pins), the bench line, the rocprofv3 kernel trace + stats of the bench command, SQ / LDS / clock counters,
# the per-launch table, configs 2 / 3 / 5 and "hc" with their parity and CPU legs, the other solvers' bench lines, the reference's
# criterion benches.  The full GPU test suite is a separate call (tools/r06_call6.sh style) -- the driver runs it at round end anyway.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_final
rm -rf $O; mkdir -p $O
cat $R/.evidence_commit > $O/commit.txt 2>/dev/null
export EVIDENCE_COMMIT=$(cat $O/commit.txt 2>/dev/null)
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE      # setup data only (csrc/hostmath.cc); bench.py runs without it below
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s" | tee -a $O/laps.txt; }
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
# 1. A/B timings through the hipGraph replay of the bench workload (one process per side; the default twice)
(timeout 120 python tools/ab_step.py; RPDE_GEMM_CTSWAP=0 timeout 120 python tools/ab_step.py; RPDE_LIFT_STRUCT=0 timeout 120 python tools/ab_step.py;
 RPDE_S6_DERIVE=1 timeout 120 python tools/ab_step.py; RPDE_GEMM_WAVES=4 timeout 120 python tools/ab_step.py; RPDE_LINE_BATCH=15 timeout 120 python tools/ab_step.py;
 timeout 120 python tools/ab_step.py) > $O/ab_step.txt 2>$O/ab_step.err
lap "A/B timings"
cd /tmp
# 2. PMC traffic (separate passes, as the guide prescribes)
timeout 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch $O/fetch --write $O/write --schedule $O/schedule.json --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json     # bench.py picks roofline.traffic up from here
lap "PMC traffic"
# 3. the bench line (no eig cache: the command the driver runs)
(unset RPDE_EIG_CACHE; RPDE_BENCH_DETAIL=$O/bench_parity_detail.json timeout 480 python bench.py > $O/bench.json 2> $O/bench.err)
lap "bench"
# 4. rocprofv3 kernel trace + stats of the bench command
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
cd $R
python tools/trace_by_tag.py $O/trace $O/schedule.json $O/trace_by_tag.csv 2> $O/trace_by_tag.log
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
lap "rocprofv3 stats"
# 5. SQ counters per launch, the LDS counters, clock and MFMA-pipe counters
cd /tmp
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/sq -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_sq.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/lds -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_lds.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_mfma.log 2>&1
cd $R
python tools/pmc_counters.py $O/sq $O/schedule.json > $O/sq_counters.txt 2>&1
python tools/pmc_counters.py $O/lds $O/schedule.json > $O/lds_counters.txt 2>&1
python tools/pmc_counters.py $O/mfma $O/schedule.json > $O/mfma_counters.txt 2>&1
lap "SQ / LDS / MFMA counters"
# 6. per-launch HIP-event table, kernel resources
timeout 200 python tools/profile_step.py > $O/profile_step.txt 2>&1
bash tools/kernel_resources.sh > $O/kernel_resources.txt 2>/dev/null
lap "profile_step"
# 7. BASELINE configs 2, 3, 5 (one GPU) and "hc" with parity + CPU legs; per-launch tables of 2 / 3 / 5
timeout 400 python bench.py --nx 1025 --ny 1025 --ra 1e7 --dt 1e-3 --steps 200 > $O/bench_1025.json 2> $O/bench_1025.err
timeout 300 python bench.py --periodic --nx 4096 --ny 1025 --ra 1e8 --dt 5e-4 --steps 200 --cpu-steps 2 > $O/bench_periodic.json 2> $O/bench_periodic.err
timeout 400 python bench.py --periodic --nx 16384 --ny 2049 --ra 1e9 --dt 1e-4 --aspect 8 --steps 30 --cpu-steps 1 --no-cpu-single-thread > $O/bench_config5_1gpu.json 2> $O/bench_config5.err
timeout 400 python bench.py --bc hc --steps 30 > $O/bench_hc.json 2> $O/bench_hc.err
timeout 100 python tools/profile_step.py 1025 1025 > $O/profile_step_config2.txt 2>&1
timeout 100 python tools/profile_step.py 4096 1025 periodic > $O/profile_step_config3.txt 2>&1
timeout 100 python tools/profile_step.py 16384 2049 periodic > $O/profile_step_config5.txt 2>&1
lap "configs 2 3 5 hc"
# 8. the other solvers and the reference's criterion benches
for sv in adjoint lnse lnse_adjoint nonlin; do
  timeout 400 python bench.py --solver $sv --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --cpu-steps 2 2>> $O/bench_f4.err | grep '^{' > $O/bench_${sv}_1025.json
done
timeout 300 python bench.py --solver adjoint --nx 4097 --ny 4097 --ra 1e8 --dt 1e-4 --steps 4 --warmup 2 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_adjoint_4097.json
timeout 300 python bench.py --solver lnse --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_lnse_4097.json
RPDE_LNSE_FUSED=0 timeout 400 python bench.py --solver lnse --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_lnse_generic_1025.json
RPDE_LNSE_FUSED=0 timeout 300 python bench.py --solver lnse --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_lnse_generic_4097.json
timeout 300 python bench.py --solver nonlin --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_nonlin_4097.json
RPDE_LNSE_FUSED=0 timeout 400 python bench.py --solver nonlin --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_nonlin_generic_1025.json
RPDE_LNSE_FUSED=0 timeout 300 python bench.py --solver nonlin --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_nonlin_generic_4097.json
RPDE_LNSE_FUSED=0 timeout 400 python bench.py --solver lnse_adjoint --nx 1025 --ny 1025 --ra 1e5 --dt 0.01 --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_lnse_adjoint_generic_1025.json
timeout 300 python bench.py --solver lnse_adjoint --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_lnse_adjoint_4097.json
RPDE_LNSE_FUSED=0 timeout 300 python bench.py --solver lnse_adjoint --nx 4097 --ny 4097 --ra 1e8 --dt 2e-4 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/bench_f4.err | grep '^{' > $O/bench_lnse_adjoint_generic_4097.json
timeout 600 python tools/bench_criterion.py --out $O/criterion.json > $O/criterion.txt 2>&1
lap "other solvers, criterion"
rm -f $O/*/*.db $O/*/*/*.db
find $O -name '*kernel_trace.csv' -size +8M -delete
rm -f $O/fetch/*counter_collection.csv $O/write/*counter_collection.csv $O/sq/*counter_collection.csv $O/lds/*counter_collection.csv $O/mfma/*counter_collection.csv $O/*/*/*counter_collection.csv
cat $O/ab_step.txt; tail -c 1500 $O/bench.json; echo; cat $O/trace_by_tag.log; head -22 $O/pmc_traffic.txt; cat $O/laps.txt
