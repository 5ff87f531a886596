"""bench.py harness self-test on CPU: the same script, `--dry-run-emu` (host emulation build of the
kernel sources), tiny grid.  Checks the JSON contract of the one line it prints -- not a number."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(emu_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-emu", "--nx", "33", "--ny", "33",
                          "--ra", "1e5", "--dt", "0.01", "--steps", "3", "--warmup", "1", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["dry_run"] is True and "NOT a measurement" in d["data"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity",
                "ms_per_step_update_plus_exit", "transform_pass"):
        assert key in d, key
    for key in ("seconds_per_step_by_phase", "single_thread"):
        assert key in d["cpu_baseline"], key
    for key in ():
        assert key in d, key
    assert d["dtype"] == "f64" and d["n_gpus"] == 1 and d["steps"] == 3 and "workload" in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in d["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["parity"]["ok"] and d["parity"]["steps"] in (3, 4) and max(d["parity"]["rel_l2"].values()) < 1e-10


def test_bench_line_contract_two_ranks(emu_lib):
    """The launch form the driver uses for N > 1 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W -- on the emulation build (gloo, host buffers): both ranks
    build their pencil of the engine, step it through the all-to-all callback, rank 0 prints ONE line with n_gpus = 2, the
    whole-job value, `exchange` and strong scaling; nothing is measured here, the harness is exercised."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--dry-run-emu", "--gpus", "2", "--nx", "33", "--ny", "33", "--ra", "1e5", "--dt", "0.01",
                          "--steps", "3", "--warmup", "1", "--profile-steps", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["scaling"] == "strong" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "pencil-sharded over 2 GPUs" in d["config"]["parallelism"] and "torch-gloo" in d["config"]["parallelism"]
    assert d["exchange"]["alltoalls_per_step"] > 0 and d["exchange"]["bytes_sent_per_gpu_per_step"] > 0
    per = d["exchange"]["per_exchange"]                # every exchange of a step with its time (the measured form of DESIGN.md 6's table)
    assert {"T1", "T2", "T4b", "T4c"} <= {r["tag"] for r in per} and all(r["ms"] > 0 for r in per) and d["exchange"]["ms_per_step_in_exchanges"] > 0
    assert "cpu_baseline" not in d                     # rank 0 at N = 1 only
