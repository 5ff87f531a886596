#!/bin/bash
# VGPR / SGPR / scratch / LDS of every device kernel (device-only assembly of csrc/kernels.cc).
# usage: tools/kernel_resources.sh [extra hipcc flags]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/rpde_kernels.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip --cuda-device-only -S "$@" \
  "$R/rustpde_mpi_amd/csrc/kernels.cc" -o "$OUT"
python3 - "$OUT" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
# amdhsa metadata: one YAML document at the end
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n\s+- \.|\namdhsa\.target)", txt, re.S):
    name, body = m.group(1), m.group(2)
    if ".vgpr_count" not in body:
        continue
    g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", body) or [None, "?"])[1]
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    print(f"vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} "
          f"lds {g('group_segment_fixed_size'):>6} spill_v {g('vgpr_spill_count'):>3}  {dem[:110]}")
PY
