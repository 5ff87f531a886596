set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r01final
rm -rf $O; mkdir -p $O
python tools/profile_step.py > $O/profile_step.txt 2>&1
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o step -- python $R/tools/pmc_step.py --out $O/schedule.json > $O/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch $O/fetch --write $O/write --schedule $O/schedule.json --out $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1
rm -rf $O/fetch/*/*.db $O/write/*/*.db
cat $O/profile_step.txt $O/pmc_traffic.txt
