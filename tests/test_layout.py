"""Repository layout rules: the oracle and the host emulation build are test infrastructure.

* the product package (rustpde_mpi_amd/, including its C++ sources) never imports, loads or names
  `oracle` or the emulation library;
* bench.py touches the oracle only inside its cpu_baseline / parity legs;
* __graft_entry__ touches it only inside smoke()."""
import ast
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources(d, exts):
    for base, _, files in os.walk(d):
        if "__pycache__" in base or os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith(exts):
                yield os.path.join(base, f)


def test_product_package_never_touches_the_oracle_or_the_emulation():
    pkg = os.path.join(ROOT, "rustpde_mpi_amd")
    for path in _sources(pkg, (".py", ".cc", ".h")):
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), path
        assert "librustpde_emu" not in src, path
        if path.endswith(".py"):
            assert not re.search(r"^\s*(from|import)\s+tests\b", src, re.M), path


def _functions_importing(path, module):
    tree = ast.parse(open(path).read())
    hits = []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            for sub in ast.walk(node):
                if isinstance(sub, ast.ImportFrom) and (sub.module or "").split(".")[0] == module:
                    hits.append(node.name)
                if isinstance(sub, ast.Import) and any(a.name.split(".")[0] == module for a in sub.names):
                    hits.append(node.name)
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and
           ((getattr(n, "module", None) or "").split(".")[0] == module or
            any(a.name.split(".")[0] == module for a in n.names))]
    return set(hits), top


def test_bench_uses_the_oracle_only_as_cpu_baseline_and_checker():
    fns, top = _functions_importing(os.path.join(ROOT, "bench.py"), "oracle")
    assert not top and fns <= {"cpu_baseline", "cpu_baseline_phases", "cpu_baseline_other_solver"}, (fns, top)


def test_entry_uses_the_oracle_only_in_smoke():
    fns, top = _functions_importing(os.path.join(ROOT, "__graft_entry__.py"), "oracle")
    assert not top and fns <= {"smoke"}, (fns, top)


# Every environment switch the product reads.  A/B switches exist only where a test exercises both sides (named next to them);
# INTEGRATION.md section 4 documents the same list.
ENV_SWITCHES = {
    "RPDE_LAPACK_LIB", "RPDE_RCCL_LIB",            # run-time libraries
    "RPDE_GRAPH",                                   # hipGraph replay on / off              (test_gpu_parity.test_graph_*)
    "RPDE_HC_BLOCKED",                              # "hc": blocked PdmaPlus2 column solve / one thread per column (tests/test_hc.py)
    "RPDE_OVERLAP",                                 # pencil-sharded: transposes of one field on a second stream under the next field's compute (test_sharded.test_overlap_*)
    "RPDE_SYNC_LAUNCHES",                           # diagnostics: every launch named and waited for
    "RPDE_ALLOC_LOG",                               # emulation build only: allocation trace for tools/fault_repro
    "RPDE_ARENA", "RPDE_ARENA_GUARD",               # device memory from slabs / one hipMalloc per buffer; guard granules (tests/test_arena.py)
    "RPDE_WHOLE_LINE", "RPDE_S1_LINE", "RPDE_S3_LINE", "RPDE_S5_LINE", "RPDE_S6_LINE", "RPDE_S8_LINE", "RPDE_S9_LINE", "RPDE_DCT_LINE", "RPDE_CONV_LINE",
                                                    # whole-line kernel / line program per stage (test_whole_line_stage_*, test_emu_parity)
    "RPDE_S1_PAIR", "RPDE_LINE_BATCH",              # S1 pair form, batched launches of 1025-point lines (test_whole_line_kernels_equal_line_programs_1025)
    "RPDE_COL_ONEPASS", "RPDE_COL1_W", "RPDE_COL1_FORCE",   # column scans: one pass / three kernels, blocks per workgroup, skip the residency test (test_column_scans_in_one_pass*)
    "RPDE_ADJOINT_FUSED",                           # Navier2DAdjoint: forward step on Navier2DEngine's fused schedule / generic operators (tests/test_adjoint.py test_emu_adjoint_fused_forward_step)
    "RPDE_FORK",                                    # the two independent chains behind G2 on two streams / graph branches (test_gpu_parity.test_round6_forked_tail_bit_identical)
    "RPDE_LNSE_FUSED",                              # Navier2DLnse::update / update_adjoint, Navier2DNonLin::update on Navier2DEngine's fused schedule / generic operators (tests/test_adjoint.py test_emu_*_on_the_fused_schedule)
    "RPDE_PER_ROWS",                                # periodic S5 / S8 / S9: element-wise kernels / line programs (test_periodic_elementwise_stages_equal_line_programs)
    "RPDE_GEMM_WAVES",                              # A/B of round 6: the GEMM's 128-tile by eight waves (test_gpu_parity.test_round6_gemm_eight_waves_bit_identical)
    "RPDE_S6_DERIVE",                               # A/B of round 6 (=1; not the default): S6 reads one factor row per line and derives the other three (test_emu_parity.test_s6_poisson_rows_as_one_kernel, test_gpu_parity.test_round6_s6_derived_factors)
    "RPDE_LIFT_STRUCT",                             # A/B of round 6: the step reads whole lift arrays / what analyse_lift found non-redundant in them (tests/test_emu_parity.test_lift_structure_*, test_gpu_parity.test_round6_lift_structure_bit_identical)
    "RPDE_GEMM_CTSWAP",                             # A/B of round 6: G2 accumulated transposed (128-byte stores) / the 32-byte transposed store (test_gpu_parity.test_round6_gemm_transposed_accumulation_bit_identical)
    "RPDE_GEMM_PERSIST",                            # A/B of round 6: both parity blocks of an eigen-transform by 512 persistent workgroups (test_gpu_parity.test_round6_gemm_persist_bit_identical)
    "RPDE_S1_SPLIT", "RPDE_GEMM_PEEL",                # A/B of round 5: S1 as two launches, the peeled GEMM loop (test_gpu_parity.test_round5_ab_switches)
    "RPDE_S6_KEEP",                                 # A/B of round 5: S6 with the back-substitution factors read twice (tests/test_emu_parity.test_s6_*, test_gpu_parity.test_round5_ab_switches)
    "RPDE_GEMM_LDS",                                # A/B of round 5: LDS layout of the GEMM's operand stages (test_gpu_parity.test_round5_ab_switches)
    "RPDE_EIG_CACHE",                               # directory that keeps the x eigen-decomposition between engines of one operator (tests/conftest.py sets it; test_eig_cache)
    "RPDE_DCT_DIRECT",                              # cross-check of the Bluestein lines: the O(n^2) cosine sum for n <= 500 (tests/test_general_lengths.py)
    "RPDE_COL_PAIR", "RPDE_GEMM_SWIZZLE",           # XCD pairing of the three-kernel correction-y, GEMM tile order (tests/test_gpu_parity)
}


def test_environment_switches_are_the_documented_list():
    found = set()
    for path in _sources(os.path.join(ROOT, "rustpde_mpi_amd"), (".cc", ".h", ".py")):
        src = open(path).read()
        found |= set(re.findall(r'getenv\("(RPDE_[A-Z0-9_]+)"\)', src))
        found |= set(re.findall(r'whole_line_on\("(RPDE_[A-Z0-9_]+)"', src))
        found |= set(re.findall(r'environ(?:\.get)?\(?\[?"(RPDE_[A-Z0-9_]+)"', src))
    assert found == ENV_SWITCHES, (sorted(found - ENV_SWITCHES), sorted(ENV_SWITCHES - found))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in ENV_SWITCHES:
        assert name in doc, f"{name} is not documented in INTEGRATION.md"


def test_committed_pmc_evidence_belongs_to_these_kernel_sources():
    """profiles/r04_pmc_traffic.json (rocprofv3 --pmc passes of the round's evidence run) carries a sha256 over
    rustpde_mpi_amd/csrc/*.{h,cc}; bench.py reports roofline.traffic_source.stale = true when the running sources hash
    differently.  The committed evidence of the latest round has to belong to the committed kernels."""
    import glob
    import hashlib
    import json
    root = os.path.join(ROOT, "rustpde_mpi_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(root)):
        if f.endswith((".h", ".cc")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))[-1]
    d = json.load(open(latest))
    assert d.get("csrc_sha256") == h.hexdigest(), f"{os.path.basename(latest)} was collected on other kernel sources"
