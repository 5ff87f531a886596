"""Build tests/emu/librustpde_emu.so (host emulation of the kernel sources; see README.md).

One object per source under tests/emu/build/, compiled in parallel and rebuilt when the source or any header is newer."""
import fcntl
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "..", "rustpde_mpi_amd", "csrc"))
OUT = os.path.join(HERE, "librustpde_emu.so")
BDIR = os.path.join(HERE, "build")
SOURCES = ["kernels.cc", "gemm.cc", "hostmath.cc", "ops.cc", "rccl_transport.cc", "h5lite.cc", "engine.cc", "adjoint.cc", "capi.cc"]
FLAGS = ["-std=c++17", "-O2", "-DRPDE_EMU", "-fPIC", "-Wno-unknown-pragmas"]


def build(force=False):
    # one builder at a time (pytest-xdist workers, the ranks of a multi-process test): the others wait and find it built
    os.makedirs(BDIR, exist_ok=True)
    with open(os.path.join(BDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build(force)


def _build(force):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.normpath(os.path.join(HERE, "..", "..", "include", "rustpde_hip.h")))
    hdr_t = max(map(os.path.getmtime, headers))
    os.makedirs(BDIR, exist_ok=True)
    jobs, objs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(BDIR, s.replace(".cc", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append(["g++", *FLAGS, "-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(subprocess.check_call, jobs))
    if jobs or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(map(os.path.getmtime, objs)):
        tmp = OUT + ".tmp%d" % os.getpid()
        subprocess.check_call(["g++", "-shared", "-fPIC", *objs, "-o", tmp, "-ldl"])
        os.replace(tmp, OUT)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
