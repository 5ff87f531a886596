"""Build the HIP library in-tree: rustpde_mpi_amd/librustpde_hip.so (gfx950 only).

    python -m rustpde_mpi_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects are cached under rustpde_mpi_amd/csrc/build/ and
rebuilt when a source or header is newer.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "librustpde_hip.so")
SOURCES = ["kernels.cc", "gemm.cc", "hostmath.cc", "ops.cc", "rccl_transport.cc", "h5lite.cc", "engine.cc", "adjoint.cc", "capi.cc"]
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _newest_header():
    t = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(root, f)))
    return t


def build(force: bool = False, verbose: bool = True, variant: str = "", flags=()) -> str:
    """`variant` / `flags`: an A/B build with extra compiler flags into librustpde_hip_<variant>.so (experiments only;
    the package loads librustpde_hip.so)."""
    hipcc = _hipcc()
    out = OUT if not variant else OUT.replace(".so", f"_{variant}.so")
    bdir = os.path.join(CSRC, "build" if not variant else f"build_{variant}")
    os.makedirs(bdir, exist_ok=True)
    hdr_t = _newest_header()
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src.replace(".cc", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_t)):
            continue
        jobs.append((sp, obj))

    def compile_one(job):
        sp, obj = job
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip",
               "-Wno-unused-result", *flags, "-c", sp, "-o", obj + ".tmp"]
        # hipcc reads the sources twice (device pass, host pass): an edit in between gives an object whose host and device
        # halves disagree -- compile again until the sources stood still for the whole compile
        while True:
            seen = max(os.path.getmtime(sp), _newest_header())
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            if max(os.path.getmtime(sp), _newest_header()) == seen:
                break
        os.replace(obj + ".tmp", obj)

    if jobs:   # the translation units are independent: all of them at once (kernels.cc alone is most of the wall time)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, jobs))
    if (force or not os.path.exists(out)
            or os.path.getmtime(out) < max(os.path.getmtime(o) for o in objs)):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    # python -m rustpde_mpi_amd.build [--force] [variant -DFLAG ...]
    print(build(force="--force" in sys.argv, variant=args[0] if args else "", flags=tuple(args[1:])))
