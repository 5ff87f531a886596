export TMPDIR=/tmp
mkdir -p gpurun_out/r06z
export RPDE_EIG_CACHE=/tmp/rpde_eig; mkdir -p $RPDE_EIG_CACHE
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r06z/pytest_full.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r06z/pytest_full.txt | tail -3
unset RPDE_EIG_CACHE
bash tools/evidence_r06.sh
