"""Launches with several processes on ONE GPU: the driver's `torch.distributed.run` line for N > 1 ranks and the peer
transport between processes (HIP IPC), i.e. everything of a multi-GPU run but xGMI.

This file sorts first on purpose.  Eight ranks that wait for each other inside their kernels need the GPU to run all
of them side by side; a pytest process that already holds a few dozen contexts (streams = hardware queues) of earlier
tests oversubscribes the queues, the scheduler then time-slices whole processes, and an exchange that takes 4 s took
100 s and more (measured: the same tests at the end of the suite).  Here the parent has not created a context yet."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _quiet(text, n=3000):
    """stderr of a torchrun launch without the rendezvous chatter"""
    keep = [l for l in text.splitlines() if "[Gloo]" not in l and "socket.cpp" not in l and "amdgpu.ids" not in l]
    try:   # the whole text for a post-mortem (gpurun_out/ travels back from the GPU box)
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "failed_launch_%d.log" % os.getpid()), "a") as fh:
            fh.write("\n".join(keep) + "\n=====\n")
    except OSError:
        pass
    return "\n".join(keep)[-n:]

def _bench(args, env_extra=None, launcher=True, nproc=2, port=None, timeout=600):
    """bench.py for N ranks on GPU 0: through the driver's launch line, or plain `python bench.py --gpus N`"""
    env = dict(os.environ, SC_BENCH_SINGLE_DEVICE="1")
    env.pop("SC_BENCH_TRANSPORT", None)
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def _one_line(out):
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    """the launch the driver uses for N > 1 (torch.distributed.run, one process per rank), with both
    ranks on GPU 0 - the only multi-rank configuration a one-GPU box can run; checks the rendezvous, the sharded
    proof's parity gate and the single JSON line of rank 0, over the host (gloo) transport and over the default data
    planes"""
    port = 29650 + (os.getpid() % 200)
    args = ["--gpus", "2", "--num-vars", "20", "--steps", "4", "--warmup", "1", "--cpu-num-vars", "0"]
    d = _one_line(_bench(args, {"SC_BENCH_TRANSPORT": "host"}, port=port))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert "host" in d["config"]["transport"] and list(d["config"]["transports"]) == ["host"]
    # default: EVERY in-library data plane is attempted and timed.  In-kernel exchange through peer-mapped inboxes
    # (HIP IPC between the two processes) works on one device; RCCL refuses two ranks on one GPU ("Duplicate GPU
    # detected"), which the line must say instead of hiding - on a multi-GPU node both carry a time; and the one-process
    # plane: rank 0 alone over ONE handle of two devices (sc_ctx_create_multi), the other rank only at the barriers.
    # (with a CPU sample of the same size: an N > 1 line carries cpu_baseline too, and the timed sharded transcript is compared
    # with the oracle's bit for bit - VERDICT r05 next 1)
    d = _one_line(_bench(args[:-1] + ["20"], port=port + 1))
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0 and "rank 0's host cores" in d["cpu_baseline"]["sample"]
    assert d["cpu_baseline"]["reference_probe"]["cargo"] == "absent" or d["cpu_baseline"]["reference_probe"]["cargo"].startswith("/")
    assert "2-rank transcript is bit-exact vs the CPU oracle at n=20" in d["config"]["parity_gate"]
    tr = d["config"]["transports"]
    assert d["n_gpus"] == 2 and d["config"]["transport"].split(" ")[0].split("(")[0] in ("peer", "inproc") and d["value"] > 0
    assert set(tr) == {"peer", "rccl", "inproc"}
    assert tr["peer"]["ms_per_step"] > 0 and tr["peer"]["comm_nranks"] == 2
    assert tr["inproc"]["ms_per_step"] > 0 and tr["inproc"]["comm_nranks"] == 2
    assert tr["rccl"]["ms_per_step"] is None and tr["rccl"]["error"]
    assert "same transcript" in d["config"]["parity_gate"]
    assert d["roofline"]["per_gpu"] is True and 0 < d["roofline"]["frac"] <= 1


@pytest.mark.gpu
def test_bench_survives_a_collective_library_that_hangs():
    """an ncclCommInitRank that never returns (injected) must not take the peer plane's measurement with it: the line
    comes out with the RCCL plane marked unusable, exit code 0, no process left behind"""
    port = 29550 + (os.getpid() % 90)
    out = _bench(["--gpus", "2", "--num-vars", "18", "--steps", "3", "--warmup", "1", "--cpu-num-vars", "0"],
                 {"SC_BENCH_TEST_RCCL_HANG": "1", "SC_BENCH_RCCL_INIT_TIMEOUT": "3"}, port=port, timeout=300)
    d = _one_line(out)
    tr = d["config"]["transports"]
    assert tr["peer"]["ms_per_step"] > 0 and tr["rccl"]["ms_per_step"] is None and "time limit" in tr["rccl"]["error"]


@pytest.mark.gpu
def test_bench_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: bench.py starts the two rank processes itself (a
    child launcher, before this process touches the GPU) and relays rank 0's line and the exit code"""
    args = ["--gpus", "2", "--num-vars", "20", "--steps", "4", "--warmup", "1", "--cpu-num-vars", "0"]
    d = _one_line(_bench(args, launcher=False))
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["value"] > 0
    assert d["config"]["transports"]["peer"]["ms_per_step"] > 0
    # the exit code is relayed too
    out = _bench(args, {"SC_BENCH_FAIL_TRANSPORT_RANK": "all"}, launcher=False)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_eight_ranks_n28_on_one_device():
    """BASELINE config 4 at its stated shape - n = 28 over 8 ranks of 2^25-entry shards - through the driver's launch
    line and the default data plane (in-kernel exchange through HIP-IPC-mapped inboxes), the eight processes sharing
    GPU 0 (what a one-GPU box can run: everything but xGMI).  The run gates on the verifier identities of the n = 28
    transcript.  One attempt, with the library's default bound on the in-kernel waits."""
    port = 29750 + (os.getpid() % 90)
    d = _one_line(_bench(["--gpus", "8", "--steps", "6", "--warmup", "2", "--cpu-num-vars", "0"], {"SC_BENCH_TRANSPORT": "peer"},
                         nproc=8, port=port, timeout=900))
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["value"] > 0
    c = d["config"]
    assert c["num_vars"] == 28 and c["transport"] == "peer" and "verifier identities at n=28 ok" in c["parity_gate"]
    assert c["transports"]["peer"]["comm_nranks"] == 8
    sched = c["schedule"]
    assert sched[0] == ["gram_pass", 0, 4, 25] and sched[-1][0] == "grid_pass"  # 2^25-entry shards: the matrix-core first pass, five-round passes to the end
    assert sum(s[2] for s in sched if s[0] != "gram_finish") == 28      # (gram_finish: the second launch of the first pass)


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["all", "1"])
def test_bench_refuses_silent_transport_fallback(which):
    """a rank whose data-plane transports cannot be created (injected here, on every rank or on one) makes bench.py exit
    non-zero on every rank instead of quietly measuring the host transport"""
    port = 29850 + (os.getpid() % 100) + (7 if which == "1" else 0)
    out = _bench(["--gpus", "2", "--num-vars", "16", "--steps", "2", "--warmup", "0", "--cpu-num-vars", "0"],
                 {"SC_BENCH_FAIL_TRANSPORT_RANK": which}, port=port, timeout=300)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


RCCL_DOUBLE = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")


def _workers(nproc, mode, port, timeout=600, transport="peer", extra_env=None):
    env = dict(os.environ, SC_PEER_WORKER_MODE=mode, SC_WORKER_TRANSPORT=transport)
    if transport == "rccl":      # RCCL refuses two ranks on one GPU: the test double stands in for it (tests/rccl_double)
        assert os.path.exists(RCCL_DOUBLE), "build it: __graft_entry__.build()"
        env["SC_RCCL_LIBRARY"] = RCCL_DOUBLE
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "peer_worker.py")],
                          capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 8])
def test_peer_transport_processes_one_device(nproc):
    """2 / 8 PROCESSES on GPU 0 exchanging HIP IPC handles (the mapping a multi-GPU node uses, minus xGMI): sharded
    proofs, sharded evaluate and the degenerate paths, bit-exact against the oracle on every rank.  One attempt, with
    the library's default bound on the in-kernel waits."""
    out = _workers(nproc, "parity", 29950 + (os.getpid() % 40) + nproc)
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    assert out.stdout.count("PEER-OK") == nproc, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_peer_exchange_fault_injection(nproc):
    """adversarial timing of the in-kernel exchange: one rank launches every sharded pass 1..100 ms late (results must
    stay bit-exact: a late rank is only waited for), one rank falls out of step by an exchange tag (every rank must
    fail with SC_ERR_RCCL within the bound, none may hang or return a wrong transcript), interleaved sharded provers
    on one context (gathered tables must not be overwritten by a later gather), gathers longer than the arena"""
    out = _workers(nproc, "faults", 30050 + (os.getpid() % 40) + nproc)
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    assert out.stdout.count("FAULTS-OK") == nproc, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_config5_pieces_over_peer_processes(nproc):
    """sharded G::new, the sharded GKR W prover (sharded wiring, generic sums) and the sharded triangle prover between
    PROCESSES over the peer transport, with an arena so small that every vector all-reduce and gather goes in chunks"""
    out = _workers(nproc, "widened", 30150 + (os.getpid() % 40) + nproc)
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    assert out.stdout.count("WIDENED-OK") == nproc, out.stdout[-3000:]


# ---- the RCCL plane with N > 1 (VERDICT r04 missing 2): Transport::kRccl between processes on ONE device, librccl replaced by the
# test double of tests/rccl_double (selected through SC_RCCL_LIBRARY, as any other RCCL build would be).  What runs is the
# PRODUCT's control flow for N > 1 - fetch_limbs' all-reduce branch, the all-reduce + copy-out behind every five-round and
# matrix-core pass, gather_table at the tail, ncclCommCount - which no one-GPU box could reach before; what does not run is RCCL.

@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_rccl_plane_processes_one_device(nproc):
    """sharded proofs (n = 1 .. 22, both sharded schedules, the gather at the tail), sharded evaluate and the degenerate paths over
    Transport::kRccl with 2 / 4 / 8 ranks: bit-exact against the oracle on every rank, comm_nranks == world"""
    out = _workers(nproc, "parity", 30250 + (os.getpid() % 40) + nproc, transport="rccl")
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    assert out.stdout.count("RCCL-OK") == nproc, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_config5_pieces_over_rccl_processes(nproc):
    """sharded G::new (the f_a vector all-reduce), the sharded GKR W prover and the sharded triangle prover over Transport::kRccl"""
    out = _workers(nproc, "widened", 30350 + (os.getpid() % 40) + nproc, transport="rccl")
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    assert out.stdout.count("WIDENED-OK") == nproc, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 8])
def test_rccl_plane_a_rank_that_dies(nproc):
    """the last rank leaves between two proofs: every survivor's next proof fails with SC_ERR_RCCL inside the bound, none hangs"""
    out = _workers(nproc, "rccl_death", 30450 + (os.getpid() % 40) + nproc, timeout=300, transport="rccl",
                   extra_env={"SC_RCCL_DOUBLE_TIMEOUT_MS": "1500"})
    assert out.stdout.count("RCCL-DEATH-OK") == nproc - 1, (_quiet(out.stdout, 3000), _quiet(out.stderr))


@pytest.mark.gpu
@pytest.mark.parametrize("transport,nproc", [("peer", 2), ("peer", 8), ("rccl", 4)])
def test_headline_worker_on_one_device(transport, nproc):
    """the worker mode tests/test_gpu_multi_device.py runs at n = 20 and n = 28 on a real node (rank 0's oracle transcript handed to
    every rank, every rank's c_1 and round triples compared, launches == plan, cold and warm proof) - here at n = 16 and n = 22 with
    every rank on GPU 0, so that the first run on distinct devices is not the first run of this code"""
    out = _workers(nproc, "headline", 30650 + (os.getpid() % 40) + nproc, timeout=600, transport=transport,
                   extra_env={"SC_WORKER_NUM_VARS": "16,22"})
    assert out.returncode == 0, (_quiet(out.stdout, 1500), _quiet(out.stderr))
    assert out.stdout.count("HEADLINE-OK") == nproc, out.stdout[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 4])
def test_rccl_plane_a_collective_that_never_completes(nproc):
    """what a dead rank looks like under the REAL library: ncclAllReduce returns ncclSuccess and what it queued never completes
    (the double's SC_RCCL_DOUBLE_ASYNC_HANG mode: a host function blocks the stream until ncclCommAbort).  The LIBRARY's bound
    must end the wait: option "rccl_timeout_ms" -> ncclCommAbort, SC_ERR_RCCL naming the bound, a poisoned context, no hang"""
    out = _workers(nproc, "rccl_death", 30550 + (os.getpid() % 40) + nproc, timeout=300, transport="rccl",
                   extra_env={"SC_RCCL_DOUBLE_TIMEOUT_MS": "300", "SC_RCCL_DOUBLE_ASYNC_HANG": "1", "SC_WORKER_RCCL_TIMEOUT_MS": "1500"})
    assert out.stdout.count("RCCL-DEATH-OK") == nproc - 1, (_quiet(out.stdout, 3000), _quiet(out.stderr))


@pytest.mark.gpu
def test_bench_eight_ranks_all_three_planes():
    """`bench.py --gpus 8` through the driver's launch line with the RCCL double in place: ONE line with all three data planes
    timed - peer, rccl (comm_nranks 8 from the transport itself), inproc - and the same transcript on each"""
    port = 29760 + (os.getpid() % 90)
    d = _one_line(_bench(["--gpus", "8", "--num-vars", "24", "--steps", "4", "--warmup", "1", "--cpu-num-vars", "24"],
                         {"SC_RCCL_LIBRARY": RCCL_DOUBLE}, nproc=8, port=port, timeout=900))
    # the five keys of a measured line, at N = 8
    assert d["value"] > 0 and d["config"]["workload"] and d["roofline"]["frac"] > 0 and d["roofline"]["per_gpu"] is True
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert "8-rank transcript is bit-exact vs the CPU oracle at n=24" in d["config"]["parity_gate"]
    tr = d["config"]["transports"]
    assert set(tr) == {"peer", "rccl", "inproc"}
    for plane in ("peer", "rccl", "inproc"):
        assert tr[plane]["ms_per_step"] and tr[plane]["ms_per_step"] > 0 and tr[plane]["comm_nranks"] == 8, (plane, tr[plane])
    assert "same transcript" in d["config"]["parity_gate"]
    assert "double" in str(tr["rccl"].get("library", "")), tr["rccl"]      # the line says what stood in for librccl
