#!/usr/bin/env python
"""The product's f64 MFMA GEMM on a few shapes (rpde_microbench gemm_nt / gemm_nn: M = K = n, N = lines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustpde_mpi_amd as R
for n, nl in ((2048, 4095), (2047, 4095), (2048, 4096), (4096, 4096), (4096, 8192), (1024, 2047), (512, 1023)):
    for w in ("gemm_nt", "gemm_nn"):
        ms = R.microbench(w, n, nl)
        print(f"{w} M=K={n} N={nl}: {ms:.4f} ms  {2.0 * n * n * nl / ms / 1e9:.2f} TFLOP/s")
