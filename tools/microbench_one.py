import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys
import rustpde_mpi_amd as R
what, n, nl, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ms = R.microbench(what, n, nl, reps)
print(f"{what} n={n} lines={nl}: {ms:.4f} ms/launch")
