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
    assert not top and fns <= {"cpu_baseline", "cpu_baseline_phases"}, (fns, top)


def test_entry_uses_the_oracle_only_in_smoke():
    fns, top = _functions_importing(os.path.join(ROOT, "__graft_entry__.py"), "oracle")
    assert not top and fns <= {"smoke"}, (fns, top)
