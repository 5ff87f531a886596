#!/usr/bin/env python
"""First hardware run of the whole-line convection kernel (csrc/dct_line.h conv_line, RPDE_CONV_LINE): written after round
2's GPU minutes were spent, so it is kept out of the test suite (a fault in a kernel that has never run on hardware would
take the whole pytest process with it).  Run this first in round 3:

    gpurun -- 'timeout 120 python tools/check_conv_line_gpu.py'

It compares RPDE_CONV_LINE=auto (on-device comparison with the line program, then the faster form; the decision is
printed to stderr) with the default over three steps at ny = 4097, confined and periodic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustpde_mpi_amd as R

for periodic in (False, True):
    fields = {}
    ctor = R.Navier2D.new_periodic if periodic else R.Navier2D.new_confined
    for flag in ("0", "auto"):
        os.environ["RPDE_CONV_LINE"] = flag
        nav = ctor(16 if periodic else 33, 4097, 1e7, 1.0, 1e-3, 1.0, "rbc", init_random=None)
        nav.set_velocity(0.2, 1.0, 1.0); nav.set_temperature(0.2, 1.0, 1.0)
        nav.update(3)
        fields[flag] = nav.physical_fields()
    for k in fields["0"]:
        e = np.linalg.norm(fields["auto"][k] - fields["0"][k]) / np.linalg.norm(fields["0"][k])
        print("periodic" if periodic else "confined", k, e)
        assert e < 1e-11, (k, e)
print("ok")
