import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, rustpde_mpi_amd as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4097
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for w in ("copy","sten","mv3","cdiff","fromortho","fdma","dct","dct2"):
    ms = R.microbench(w, n, nl)
    print(f"{w:10s} n={n} lines={nl}: {ms:.4f} ms  ({ms*1e3/ (nl/256):.2f} us per line-round @256 WGs)  {2*8*n*nl/ms/1e6:.1f} GB/s")
nx = n - 1
ms = R.microbench("rfft", nx, nl); print(f"rfft f+b   nx={nx}: {ms:.4f} ms")
