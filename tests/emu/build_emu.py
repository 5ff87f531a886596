"""Build tests/emu/librustpde_emu.so (host emulation of the kernel sources; see README.md)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "..", "rustpde_mpi_amd", "csrc"))
OUT = os.path.join(HERE, "librustpde_emu.so")
SOURCES = ["kernels.cc", "gemm.cc", "hostmath.cc", "ops.cc", "rccl_transport.cc", "h5lite.cc", "engine.cc", "adjoint.cc", "capi.cc"]


def build(force=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cc", ".h"))]
    deps.append(os.path.normpath(os.path.join(HERE, "..", "..", "include", "rustpde_hip.h")))
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) > max(map(os.path.getmtime, deps)):
        return OUT
    cmd = ["g++", "-std=c++17", "-O2", "-DRPDE_EMU", "-shared", "-fPIC", "-Wno-unknown-pragmas"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT, "-ldl"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
