"""bench.py end to end on a small instance: the JSON line carries every field the bench contract
names, the roofline and cpu_baseline objects are populated, and the in-run GPU-vs-oracle gate ran."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_small_instance():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--num-vars", "18", "--steps", "12", "--warmup", "2",
                          "--cpu-num-vars", "16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["dtype"] == "u64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and d["config"]["num_vars"] == 18
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["steps_sampled"] >= 1 and r["launches_per_step"] >= 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    """the launch the driver uses for N > 1 (torch.distributed.run, one process per rank), with both
    ranks on GPU 0 and the host (gloo) transport - the only multi-rank configuration a one-GPU box can
    run; checks the rendezvous, the sharded proof's parity gate and the single JSON line of rank 0"""
    env = dict(os.environ, SC_BENCH_SINGLE_DEVICE="1", SC_BENCH_TRANSPORT="host")
    port = 29650 + (os.getpid() % 200)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--num-vars", "20", "--steps", "4", "--warmup", "1", "--cpu-num-vars", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "host" in d["config"]["transport"]
