#!/bin/bash
# round 4, call 8: where does the host heap get corrupted (call 7: "malloc(): invalid size" in every process within 0.3 s)
export TMPDIR=/tmp; export PYTHONPATH=$PWD
O=$PWD/gpurun_out/r04h; rm -rf $O; mkdir -p $O
cat > /tmp/t.py <<'P'
import faulthandler, sys; faulthandler.enable()
import rustpde_mpi_amd as R
print("lib", R.lib().version, flush=True)
nav = R.Navier2D.new_confined(33, 33, 1e5, 1.0, 0.01, 1.0, "rbc")
print("engine", flush=True)
nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
nav.update(2)
print("stepped", flush=True)
P
python /tmp/t.py > $O/t1.txt 2>&1; tail -25 $O/t1.txt
which gdb && gdb -batch -ex run -ex bt --args python /tmp/t.py > $O/gdb.txt 2>&1; grep -n "^#" $O/gdb.txt | head -40
RPDE_COL_ONEPASS=0 python /tmp/t.py > $O/t2.txt 2>&1; tail -5 $O/t2.txt
