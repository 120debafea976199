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
    assert 0 < r["frac"] <= 1.0, "roofline.frac must be a physical fraction"
    assert r["steps_sampled"] >= 1 and r["step"]["launches"] >= 1
    assert r["kernel"] and r["bytes_per_launch"] > 0 and r["avg_launch_us"] > 0
    # bytes moved come from the launches that ran: n = 18 -> first pass reads 2 x 2^18 x 8 B, ...
    st = r["step"]
    assert 0 < st["frac_of_kernel_time"] <= 1.0 and 0 < st["frac_of_wall_time"] <= st["frac_of_kernel_time"]
    assert st["bytes_moved"] >= 16 * 2**18 and st["bytes_moved"] <= 64 * 2**18
    assert abs(sum(k["bytes_per_launch"] * k["launches_per_step"] for k in r["kernels"]) - st["bytes_moved"]) < 1
    assert d["ms_per_step_median"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c


@pytest.mark.gpu
def test_bench_mle_workload():
    """--workload mle (BASELINE configs[1]): evaluate + fix_variables, own roofline object, parity gate"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mle", "--num-vars", "20", "--steps", "5",
                          "--warmup", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["config"]["num_vars"] == 20 and d["n_gpus"] == 1 and d["value"] > 0
    assert "bit-exact vs CPU oracle" in d["config"]["parity_gate"]
    r = d["roofline"]
    assert "evaluate_kernel" in r["kernel"] and r["bytes_per_launch"] == 8 * 2**20
    assert 0 < r["frac"] <= 1.0
    kinds = " ".join(k["kernel"] for k in r["kernels"])
    assert "fold_kernel" in kinds and "fix_low_kernel" in kinds and "evaluate_kernel" in kinds
    assert d["cpu_baseline"]["value"] > 0



@pytest.mark.gpu
@pytest.mark.parametrize("workload,size,cpu_size,kernel", [("gkr", 7, 6, "gkr_phase1_kernel"), ("gnew", 9, 8, "coldot_kernel"),
                                                           ("triangle", 6, 5, "")])
def test_bench_widened_workloads(workload, size, cpu_size, kernel):
    """--workload gkr | gnew | triangle: the same JSON contract (roofline from the launch log, cpu_baseline from the oracle
    on a bounded sample, parity gate inside the run) for the callers either side of the hot path"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--num-vars", str(size), "--steps", "4",
                          "--warmup", "1", "--cpu-num-vars", str(cpu_size)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and d["dtype"] == "u64"
    assert "bit-exact vs CPU oracle" in d["config"]["parity_gate"] and "workload" in d["config"]
    r = d["roofline"]
    # (the triangle workload's dominant launch group is either a streaming pass or, on small graphs, the int8 MFMA square)
    assert r["bound"] in (("hbm", "mfma") if workload == "triangle" else ("hbm",)) and 0 < r["frac"] <= 1.0 and kernel in r["kernel"]
    if workload == "triangle":
        assert r["matsq"]["bound"] == "mfma" and r["matsq"]["achieved"] > 0
    assert abs(sum(k["bytes_per_launch"] * k["launches_per_step"] for k in r["kernels"]) - r["step"]["bytes_moved"]) < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0


@pytest.mark.gpu
def test_bench_generic_field():
    """--field generic: p = 2^64-59 through the kernels of every non-Goldilocks modulus (the reference's Fp64<MontBackend<T,1>>),
    same contract, its own roofline object, bit-exact against the oracle inside the run"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--field", "generic", "--num-vars", "18", "--steps", "8", "--warmup", "2",
                          "--cpu-num-vars", "16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert "generic modulus" in d["metric"] and "18446744073709551557" in d["config"]["workload"]
    assert "MontGeneric" in d["roofline"]["kernel"] and 0 < d["roofline"]["frac"] <= 1.0
    assert "bit-exact vs CPU oracle at n=16 ok" in d["config"]["parity_gate"]
    assert d["cpu_baseline"]["value"] > 0


@pytest.mark.gpu
def test_bench_collects_its_own_pmc_traffic():
    """`roofline.traffic` first-hand: asked for at a small size (SC_BENCH_SELF_PMC=force; by default only the headline shape does it),
    bench.py starts two child runs of its own workload under `rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE` after the
    timed region and reports the dominant launch's HBM bytes as this box measured them, next to the launch log's"""
    env = dict(os.environ, SC_BENCH_SELF_PMC="force")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--num-vars", "24", "--steps", "6", "--warmup", "2", "--cpu-num-vars", "0"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    r = d["roofline"]
    assert r["traffic"] is not None and "this run's box" in r["traffic_source"], (r["traffic"], r["traffic_source"], r.get("traffic_record"))
    assert 0.9 < r["traffic"] / r["bytes_per_launch"] < 1.1, (r["traffic"], r["bytes_per_launch"])
    assert r["step"]["traffic_check"].startswith("ok")
