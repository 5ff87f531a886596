"""All-to-all for the pencil-sharded engine through torch.distributed (one process per GPU).

`TorchComm` supplies the callback of `rpde_navier2d_create_sharded` (include/rustpde_hip.h): the
engine hands over raw pointers to its send / receive buffers plus per-rank counts (in doubles) and
expects the data to have landed on return.  This replaces the reference's MPI transposes
(funspace Decomp2d via src/field_mpi.rs:456-477; rank / universe plumbing src/mpi/mod.rs:2-12).

* backend "nccl" (= RCCL over xGMI on ROCm), HIP build: the device pointers are wrapped as torch
  tensors (CUDA array interface, zero copy) and exchanged with `all_to_all_single`.
* backend "gloo": host pointers (emulation build, CPU tests) are wrapped through NumPy; device
  pointers (several ranks sharing one GPU in a test) are staged through host memory.  gloo has no
  all-to-all, so it is composed from isend / irecv.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._capi import ALLTOALLV_FN


class _DevPtr:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class TorchComm:
    def __init__(self, device_buffers: bool, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device_buffers = device_buffers
        self.calls = 0
        self.bytes_sent = 0
        self.c_callback = ALLTOALLV_FN(self._callback)

    # -- raw pointer -> 1-D float64 tensor
    def _wrap(self, ptr, n):
        if n == 0:
            return torch.empty(0, dtype=torch.float64, device="cuda" if self.device_buffers else "cpu")
        if self.device_buffers:
            return torch.as_tensor(_DevPtr(ptr, n), device="cuda")
        arr = np.ctypeslib.as_array((C.c_double * n).from_address(ptr))
        return torch.from_numpy(arr)

    def _callback(self, user, send, scounts, recv, rcounts):
        try:
            sc = [int(scounts[q]) for q in range(self.size)]
            rc = [int(rcounts[q]) for q in range(self.size)]
            self.alltoallv(send or 0, sc, recv or 0, rc)
            return 0
        except Exception as exc:  # never unwind into C
            print(f"[rustpde_mpi_amd.dist] all-to-all failed on rank {self.rank}: {exc!r}", flush=True)
            return 1

    def alltoallv(self, send_ptr, sc, recv_ptr, rc):
        self.calls += 1
        self.bytes_sent += 8 * (sum(sc) - sc[self.rank])
        s = self._wrap(send_ptr, sum(sc))
        r = self._wrap(recv_ptr, sum(rc))
        if self.backend == "nccl":
            dist.all_to_all_single(r, s, rc, sc, group=self.group)
            torch.cuda.synchronize()
            return
        # gloo: compose from point-to-point messages (host staging for device buffers)
        hs = s.cpu() if self.device_buffers else s
        hr = torch.empty(sum(rc), dtype=torch.float64) if self.device_buffers else r
        so = np.concatenate([[0], np.cumsum(sc)]).astype(int)
        ro = np.concatenate([[0], np.cumsum(rc)]).astype(int)
        reqs = []
        for q in range(self.size):
            if q != self.rank and rc[q]:
                reqs.append(dist.irecv(hr[ro[q]:ro[q + 1]], src=q, group=self.group))
        for q in range(self.size):
            if q != self.rank and sc[q]:
                reqs.append(dist.isend(hs[so[q]:so[q + 1]].contiguous(), dst=q, group=self.group))
        me = self.rank
        if sc[me]:
            hr[ro[me]:ro[me + 1]].copy_(hs[so[me]:so[me + 1]])
        for w in reqs:
            w.wait()
        if self.device_buffers:
            r.copy_(hr)
            torch.cuda.synchronize()


class RcclComm:
    """Native transport: the engine owns an RCCL communicator and runs every pencil exchange as a
    grouped ncclSend/ncclRecv all-to-all on its own HIP stream (include/rustpde_hip.h,
    rpde_navier2d_create_sharded_rccl).  torch.distributed is only used here to hand rank 0's
    ncclUniqueId to the other ranks (the job of MPI_Bcast in an MPI host)."""
    native_rccl = True

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def unique_id(self, library) -> bytes:
        """A FRESH ncclUniqueId per call (collective): an id is valid for exactly one
        ncclCommInitRank, so every engine built on this object gets its own (two engines from one
        RcclComm -- e.g. a confined and a periodic one, or a rebuild after a failure -- must not
        share one)."""
        box = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            library.call("rpde_rccl_unique_id", buf)
            box[0] = buf.raw
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group else 0,
                                   group=self.group)
        uid = bytes(box[0])
        assert len(uid) == 128
        return uid
