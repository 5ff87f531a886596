#!/usr/bin/env python
"""Single-process self test of the native RCCL transport (rpde_rccl_alltoallv_once, world size 1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustpde_mpi_amd as R
L = R.lib()
uid = C.create_string_buffer(128)
L.call("rpde_rccl_unique_id", uid)
print("uid", uid.raw[:32].hex(), flush=True)
torch.cuda.set_device(0)
a = torch.arange(5000, dtype=torch.float64, device="cuda")
b = torch.zeros(5000, dtype=torch.float64, device="cuda")
sc = (C.c_int64 * 1)(5000); rc = (C.c_int64 * 1)(5000)
torch.cuda.synchronize()
L.call("rpde_rccl_alltoallv_once", uid.raw, 0, 1, 0, C.c_void_p(a.data_ptr()), C.cast(sc, C.c_void_p),
       C.c_void_p(b.data_ptr()), C.cast(rc, C.c_void_p))
print("alltoallv ok:", bool(torch.equal(a, b)))
