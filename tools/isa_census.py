"""Static instruction census of the device code of one or more kernels.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S rustpde_mpi_amd/csrc/kernels.cc -o /tmp/k.s
    python tools/isa_census.py /tmp/k.s 'hdct_line_kernel<4096, false>' 'conv_line_kernel<4096>' ...

Counts the instructions of the kernel body by class (the whole-line kernels are straight-line code: every pass is
unrolled, loops only in the waits of nobody, so the static count is the count a wave issues -- compare with
SQ_INSTS_VALU per wave in profiles/r0N_sq_counters.txt).  Classes:
  f64 fma / mul / add    the arithmetic a transform needs
  v_mov / v_cndmask / cvt  register traffic, selects, conversions
  int valu               index and address arithmetic on the vector unit (v_add_u32, v_lshl..., v_and..., v_mad_u64 ...)
  dpp / readlane         cross-lane
  ds_read / ds_write     LDS
  vmem load / store      global / buffer / scratch
  salu, smem, s_waitcnt, s_barrier, branch
"""
import re
import subprocess
import sys
from collections import Counter


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return out


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64") or op.startswith("v_pk_fma"):
        return "f64 fma"
    if op.startswith("v_mul_f64"):
        return "f64 mul"
    if op.startswith("v_add_f64") or op.startswith("v_sub_f64"):
        return "f64 add"
    if op.startswith("v_mfma"):
        return "mfma"
    if "_f64" in op and op.startswith("v_"):
        return "f64 other"
    if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_swap"):
        return "v_mov" + (" dpp" if "dpp" in op else "")
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_cvt"):
        return "v_cvt"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or op.startswith("v_permlane"):
        return "lane ops"
    if op.startswith("v_cmp"):
        return "v_cmp"
    if op.startswith("v_"):
        return "int valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "ds_write"
    if op.startswith("ds_"):
        return "ds other"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vmem load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
        return "vmem store"
    if op.startswith("global_atomic") or op.startswith("buffer_atomic") or op.startswith("flat_atomic"):
        return "vmem atomic"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


ORDER = ["f64 fma", "f64 mul", "f64 add", "f64 other", "mfma", "v_mov", "v_mov dpp", "v_cndmask", "v_cvt", "v_cmp", "int valu", "lane ops",
         "ds_read", "ds_write", "ds other", "vmem load", "vmem store", "vmem atomic", "scratch", "smem", "salu", "s_waitcnt",
         "s_barrier", "branch", "s_nop", "other"]
VALU = {"f64 fma", "f64 mul", "f64 add", "f64 other", "v_mov", "v_mov dpp", "v_cndmask", "v_cvt", "v_cmp", "int valu", "lane ops", "mfma"}


def census(path, wanted):
    lines = open(path).read().split("\n")
    heads = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    names = demangle([h[1] for h in heads])
    res = {}
    for (i, _), nm in zip(heads, names):
        for w in wanted:
            if w in nm and w not in res:
                c = Counter()
                meta = {}
                for l in lines[i + 1:]:
                    s = l.strip()
                    if s.startswith(".Lfunc_end"):
                        break
                    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
                        continue
                    c[classify(s.split()[0])] += 1
                # resource lines follow the body as comments
                for l in lines[i:i + 200000]:
                    m = re.match(r";\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|SGPRBlocks|NumSgprs):\s*(\d+)", l.strip())
                    if m and m.group(1) not in meta:
                        meta[m.group(1)] = int(m.group(2))
                    if len(meta) >= 6:
                        break
                res[w] = (nm, c, meta)
    return res


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    res = census(path, wanted)
    cols = [w for w in wanted if w in res]
    print("| class | " + " | ".join(cols) + " |")
    print("|---|" + "---|" * len(cols))
    for k in ORDER:
        if any(res[w][1].get(k) for w in cols):
            print(f"| {k} | " + " | ".join(str(res[w][1].get(k, 0)) for w in cols) + " |")
    print("| **VALU total** | " + " | ".join(str(sum(v for k, v in res[w][1].items() if k in VALU)) for w in cols) + " |")
    print("| **all** | " + " | ".join(str(sum(res[w][1].values())) for w in cols) + " |")
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        print(f"| {key} | " + " | ".join(str(res[w][2].get(key, "")) for w in cols) + " |")


if __name__ == "__main__":
    main()
