#!/bin/bash
# gpurun_out/r05 -> profiles/r05_* (run after tools/evidence_r05.sh came back)
O=gpurun_out/r05; P=profiles
c=$(cut -c1-7 $O/commit.txt 2>/dev/null)
cp $O/commit.txt $P/r05_commit.txt
cp $O/pytest_gpu.txt $P/r05_pytest_gpu_full_${c}.txt
for f in bench bench_profiled bench_1025 bench_periodic bench_config5_1gpu bench_hc pmc_traffic schedule; do cp $O/$f.json $P/r05_$f.json 2>/dev/null; done
for f in pmc_traffic sq_counters profile_step profile_step_config2 profile_step_config3 profile_step_config5 kernel_resources; do cp $O/$f.txt $P/r05_$f.txt 2>/dev/null; done
cp $O/trace_by_tag.csv $P/r05_trace_by_tag.csv 2>/dev/null
cp $O/trace/*/*kernel_stats.csv $P/r05_kernel_stats.csv 2>/dev/null || cp $O/trace/*kernel_stats.csv $P/r05_kernel_stats.csv 2>/dev/null
for n in config2 config3 config5; do cp $O/kernel_stats_$n.csv $P/r05_kernel_stats_$n.csv 2>/dev/null; done
ls -la $P | grep r05
