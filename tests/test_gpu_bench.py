"""bench.py end to end on a small instance: the JSON line carries every field the bench contract
names, the roofline and cpu_baseline objects are populated, and the in-run GPU-vs-oracle gate ran."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _quiet(text, n=3000):
    """stderr of a torchrun launch without the rendezvous chatter"""
    keep = [l for l in text.splitlines() if "[Gloo]" not in l and "socket.cpp" not in l and "amdgpu.ids" not in l]
    try:   # the whole text for a post-mortem (gpurun_out/ travels back from the GPU box)
        import os
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "failed_launch_%d.log" % os.getpid()), "a") as fh:
            fh.write("\n".join(keep) + "\n=====\n")
    except OSError:
        pass
    return "\n".join(keep)[-n:]

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
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "host" in d["config"]["transport"]
    # the default data plane: in-kernel exchange through peer-mapped inboxes (HIP IPC between the two processes)
    env = dict(os.environ, SC_BENCH_SINGLE_DEVICE="1")
    env.pop("SC_BENCH_TRANSPORT", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port + 1), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--num-vars", "20", "--steps", "4", "--warmup", "1", "--cpu-num-vars", "0"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["transport"] == "peer" and d["value"] > 0


@pytest.mark.gpu
def test_bench_eight_ranks_n28_on_one_device():
    """BASELINE config 4 at its stated shape - n = 28 over 8 ranks of 2^25-entry shards - through the driver's launch
    line and the default data plane (in-kernel exchange through HIP-IPC-mapped inboxes), the eight processes sharing
    GPU 0 (what a one-GPU box can run: everything but xGMI).  The run gates on the verifier identities of the n = 28
    transcript; its sharded schedule ends in the unsharded grid passes after the gather."""
    env = dict(os.environ, SC_BENCH_SINGLE_DEVICE="1")
    env.pop("SC_BENCH_TRANSPORT", None)
    port = 29750 + (os.getpid() % 90)
    # On a freshly started box one of eight ranks sharing the GPU can stall for tens of seconds (first run of the suite
    # on a box only; measured: the others wait for that rank's sums until peer_spin_ms and report SC_ERR_RCCL, naming
    # the rank).  bench.py bounds the wait at 60 s; should it still trip, the launch gets up to three attempts.
    for attempt in range(3):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                              "--master-addr", "127.0.0.1", "--master-port", str(port + attempt), os.path.join(ROOT, "bench.py"),
                              "--gpus", "8", "--steps", "6", "--warmup", "2", "--cpu-num-vars", "0"],
                             capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        if out.returncode == 0 or "did not arrive within" not in out.stderr:
            break
        _quiet(out.stderr)      # keep the failed attempt's text for a post-mortem
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    c = d["config"]
    assert c["num_vars"] == 28 and c["transport"] == "peer" and "verifier identities at n=28 ok" in c["parity_gate"]
    sched = c["schedule"]
    assert sched[0] == ["pass", 0, 3, 25] and sched[-1][0] == "grid_pass"       # 2^25-entry shards; the tail is unsharded
    assert sum(s[2] for s in sched) == 28


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
def test_bench_refuses_silent_transport_fallback():
    """a rank whose data-plane transport cannot be created (injected here) makes bench.py exit non-zero on
    every rank instead of quietly measuring the host transport"""
    env = dict(os.environ, SC_BENCH_SINGLE_DEVICE="1", SC_BENCH_FAIL_TRANSPORT_RANK="all")
    port = 29850 + (os.getpid() % 100)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--num-vars", "16", "--steps", "2", "--warmup", "0", "--cpu-num-vars", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
