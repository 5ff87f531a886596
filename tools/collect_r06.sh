#!/bin/bash
# copy the judged summaries of tools/evidence_r06.sh from gpurun_out/r06_final (scratch) into profiles/ (tracked)
S=gpurun_out/r06_final; P=profiles
set -e
cp $S/commit.txt $P/r06_commit.txt
for f in bench bench_profiled bench_1025 bench_periodic bench_config5_1gpu bench_hc bench_adjoint_1025 bench_lnse_1025 bench_lnse_adjoint_1025 bench_nonlin_1025 bench_adjoint_4097 bench_lnse_4097 bench_lnse_generic_1025 bench_lnse_generic_4097 bench_nonlin_4097 bench_nonlin_generic_1025 bench_nonlin_generic_4097 bench_lnse_adjoint_4097 bench_lnse_adjoint_generic_1025 bench_lnse_adjoint_generic_4097 criterion schedule pmc_traffic; do
  [ -s $S/$f.json ] && cp $S/$f.json $P/r06_$f.json
done
[ -s $S/bench_parity_detail.json ] && cp $S/bench_parity_detail.json $P/r06_bench_parity_detail.json
for f in ab_step pmc_traffic sq_counters lds_counters mfma_counters profile_step profile_step_config2 profile_step_config3 profile_step_config5 kernel_resources laps smoke; do
  [ -s $S/$f.txt ] && cp $S/$f.txt $P/r06_$f.txt
done
cp $S/trace_by_tag.csv $P/r06_trace_by_tag.csv
cp $S/kernel_stats.csv $P/r06_kernel_stats.csv
ls -la $P/r06_* | wc -l
