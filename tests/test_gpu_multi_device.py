"""FIRST CONTACT WITH A REAL MULTI-GPU NODE.  BASELINE configs[3] / configs[4] name 8 MI355X and one RCCL
all-reduce per round; every other test of this suite runs on ONE physical device (the pool's boxes have one GPU): N ranks / N
shards / N processes of device 0, RCCL at world 1 or replaced by tests/rccl_double.  The tests below are the same workloads with

  * DISTINCT devices      rank r (shard d) on GPU r (d): SC_BENCH_SINGLE_DEVICE unset, SC_WORKER_DISTINCT_DEVICES=1
  * the REAL librccl      SC_RCCL_LIBRARY unset: ncclCommInitRank over N devices, ncclAllReduce / ncclAllGather over xGMI
  * the peer plane        HIP IPC (dmabuf) between processes that own different devices: in-kernel exchange over xGMI

and they SKIP, with the reason below, wherever torch.cuda.device_count() < 2 - on this pool, always.  On a node that has the
GPUs they are what runs before the driver's SCALE bench does, so that bench is not the first execution of any of this:

  (a) process per GPU: sharded proofs n = 1 .. 22 (every sharded schedule), then n = 20 and n = 28 (BASELINE's shape) against
      the oracle's transcript, over RCCL and over the peer plane, comm_nranks == N as the transport itself reports it;
      config 5's pieces (sharded G::new, the W prover, the triangle prover) the same way
  (b) one process, one handle over N distinct devices: n = 28 against the oracle, G::new's hipMemcpyPeerAsync block exchange, the
      pinned-tail hand-over the host reads from every device (many proofs reusing one tail slot: the stale-read race of
      WgOut::host_out would show as a wrong transcript), the widened provers
  (c) a rank that dies mid-job under the real librccl: every survivor's next proof fails with SC_ERR_RCCL inside
      "rccl_timeout_ms" (the library aborts the communicator), none hangs
  (d) bench.py --gpus N through the driver's launch line, real devices: one line, every plane timed, five keys present

torch.cuda.device_count() does not initialise the GPU on this image (the task's own note), so importing this file is safe in the
parent of the worker processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_package
from util import GOLD, challenges, oracle, pyref

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


# REHEARSAL (SC_MULTI_DEVICE_REHEARSAL=1, what this pool can do): the same test bodies with every "device" = GPU 0 and the librccl
# stand-in of tests/rccl_double - it proves nothing about distinct devices, RCCL or xGMI (the one-device suite already covers that
# ground); it executes THIS FILE's code, so that a typo here is not what the first run on a real node finds.  Run once per round:
#   SC_MULTI_DEVICE_REHEARSAL=1 python -m pytest tests/test_gpu_multi_device.py -m gpu
REHEARSAL = os.environ.get("SC_MULTI_DEVICE_REHEARSAL") == "1"
RCCL_DOUBLE = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")
NDEV = 8 if REHEARSAL else _ndev()
NMAX = 1 << (min(NDEV, 8).bit_length() - 1) if NDEV >= 1 else 1          # the largest power of two of devices, up to 8
needs_two = pytest.mark.skipif(
    NDEV < 2, reason="needs >= 2 GPUs in one box (this one shows %d): distinct devices, the real librccl with N > 1 ranks and "
                     "xGMI cannot run here; the one-device forms of these tests are test_gpu_00_multiprocess.py, test_gpu_multi.py, "
                     "test_gpu_headline.py" % NDEV)
WORLDS = sorted({2, NMAX} - {1})


def _env(transport):
    env = dict(os.environ, SC_WORKER_DISTINCT_DEVICES="1", SC_WORKER_TRANSPORT=transport, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("SC_RCCL_LIBRARY", "SC_BENCH_SINGLE_DEVICE", "SC_RCCL_DOUBLE_ASYNC_HANG", "SC_BENCH_TRANSPORT", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)                                                 # the REAL librccl, one device per rank
    if REHEARSAL:
        env.update(SC_WORKER_DISTINCT_DEVICES="0", SC_RCCL_LIBRARY=RCCL_DOUBLE, SC_BENCH_SINGLE_DEVICE="1")
    return env


def _workers(nproc, mode, transport, port, timeout=900, extra=None):
    env = _env(transport)
    env["SC_PEER_WORKER_MODE"] = mode
    env.update(extra or {})
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "peer_worker.py")],
                          capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def _tail(out):
    keep = [l for l in (out.stdout + "\n" + out.stderr).splitlines() if "[Gloo]" not in l and "socket.cpp" not in l]
    return "\n".join(keep)[-4000:]


# ---- (a) one process per GPU ------------------------------------------------------------------------------------------------------

@needs_two
@pytest.mark.parametrize("transport", ["rccl", "peer"])
@pytest.mark.parametrize("nproc", WORLDS)
def test_sharded_proofs_one_process_per_gpu(transport, nproc):
    """every sharded schedule, n = 1 .. 22, two fields, against the oracle on every rank"""
    out = _workers(nproc, "parity", transport, 31050 + (os.getpid() % 40) + nproc + (20 if transport == "peer" else 0))
    assert out.returncode == 0, _tail(out)
    assert out.stdout.count("%s-OK" % transport.upper()) == nproc, _tail(out)


@needs_two
@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_n20_and_n28_over_all_devices_vs_oracle(transport):
    """BASELINE configs[3]: the n = 28 hypercube over the box's GPUs (8 where it has them), one RCCL all-reduce per sharded
    pass - and the same over the in-kernel exchange - c_1 and all 28 round triples equal to the CPU oracle's on every rank"""
    out = _workers(NMAX, "headline", transport, 31150 + (os.getpid() % 40) + (20 if transport == "peer" else 0), timeout=1800,
                   extra={"SC_WORKER_NUM_VARS": "20,28"})
    assert out.returncode == 0, _tail(out)
    assert out.stdout.count("HEADLINE-OK") == NMAX, _tail(out)


@needs_two
@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_config5_pieces_one_process_per_gpu(transport):
    """BASELINE configs[4]'s pieces between devices: sharded G::new (the f_a vector all-reduce over xGMI), the sharded W prover
    with sharded wiring, the sharded triangle prover"""
    out = _workers(min(NMAX, 4), "widened", transport, 31250 + (os.getpid() % 40) + (20 if transport == "peer" else 0))
    assert out.returncode == 0, _tail(out)
    assert out.stdout.count("WIDENED-OK") == min(NMAX, 4), _tail(out)


@needs_two
def test_peer_plane_fault_injection_between_devices():
    """late ranks, a rank out of step, interleaved provers, gathers longer than the arena - across devices"""
    out = _workers(min(NMAX, 4), "faults", "peer", 31350 + (os.getpid() % 40))
    assert out.returncode == 0, _tail(out)
    assert out.stdout.count("FAULTS-OK") == min(NMAX, 4), _tail(out)


# ---- (c) a rank that dies under the real librccl -------------------------------------------------------------------------------

@needs_two
@pytest.mark.parametrize("nproc", WORLDS)
def test_a_rank_dies_under_the_real_rccl(nproc):
    """the last rank leaves between two proofs: the survivors' collective never completes on its own (RCCL's kernel waits for the
    peer), so the library's bound ends it - ncclCommAbort, SC_ERR_RCCL within rccl_timeout_ms, a poisoned context, no hang"""
    out = _workers(nproc, "rccl_death", "rccl", 31450 + (os.getpid() % 40) + nproc, timeout=600,
                   extra=({"SC_WORKER_RCCL_TIMEOUT_MS": "1500", "SC_RCCL_DOUBLE_ASYNC_HANG": "1", "SC_RCCL_DOUBLE_TIMEOUT_MS": "500"} if REHEARSAL
                          else {"SC_WORKER_RCCL_TIMEOUT_MS": "5000"}))
    assert out.stdout.count("RCCL-DEATH-OK") == nproc - 1, _tail(out)


# ---- (b) one process, one handle over distinct devices -------------------------------------------------------------------------

def _handle(pkg, p, n_dev, **opts):
    ctx = pkg.Context(pkg.Field(p), devices=[0] * n_dev if REHEARSAL else list(range(n_dev)))
    assert ctx.get_option("n_devices") == n_dev and ctx.get_option("transport") == 4
    for k, v in opts.items():
        ctx.set_option(k, v)
    return ctx


@needs_two
@pytest.mark.parametrize("n", [20, 28])
def test_handle_over_distinct_devices_vs_oracle(n):
    pkg = load_package()
    o = oracle(GOLD)
    oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ch_ref = challenges(o, n)
    c1_ref, ev_ref = o.prover_run_mt(oa, ob, ch_ref)
    del oa, ob
    for n_dev in WORLDS:
        ctx = _handle(pkg, GOLD, n_dev)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        G = pkg.matrix_multiplication.G(a, b)
        for rep in range(3):
            c1, evals, ch = pkg.matrix_multiplication.prove(ctx, G, pyref.SEED_R)
            assert c1 == c1_ref and np.array_equal(ch, ch_ref), (n, n_dev, rep)
            bad = [j for j in range(n) if not np.array_equal(evals[j], ev_ref[j])]
            assert not bad, (n, n_dev, rep, "rounds that differ from the oracle", bad)
        del G, a, b
        ctx.close()


@needs_two
def test_handle_tail_handover_stress_across_devices():
    """the hand-over the host reads from EVERY device: 300 proofs of alternating instances reuse one tail slot
    per device; a stale read of a device's pinned tail (a store still in some XCD's L2 when the host looks) gives a wrong
    transcript.  n = 16 .. 19 over all devices: the last device launch hands 2^9 .. 2^11-entry tables to the host"""
    pkg = load_package()
    o = oracle(GOLD)
    ctx = _handle(pkg, GOLD, NMAX)
    insts = []
    for n, (sa, sb) in [(16, (21, 22)), (17, (23, 24)), (19, (25, 26)), (16, (27, 28))]:
        ch = challenges(o, n)
        ref = o.prove(o.generate(sa, n), o.generate(sb, n), ch)
        a = pkg.DenseMultilinearExtension.generate(ctx, sa, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, sb, n)
        insts.append((pkg.matrix_multiplication.G(a, b), ref))
    for it in range(300):
        G, ref = insts[it % len(insts)]
        c1, evals, _ = pkg.matrix_multiplication.prove(ctx, G, pyref.SEED_R)
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]), it
    del insts
    ctx.close()


@needs_two
def test_handle_g_new_block_exchange_across_devices():
    """G::new behind a handle: f_a needs every device's row block of A (hipMemcpyPeerAsync over xGMI); against the oracle's G::new
    (matrix-multiplication/src/lib.rs:77-92) and through the proof"""
    pkg = load_package()
    o = oracle(GOLD)
    for n in (6, 10):
        A, B = o.generate(11, 2 * n), o.generate(12, 2 * n)
        pt = np.array([o.challenge(pyref.SEED_PT, j) for j in range(2 * n)], dtype=np.uint64)
        fa, fb = o.g_new(n, A, B, pt)
        ch = challenges(o, n)
        ref = o.prove(fa, fb, ch)
        for n_dev in WORLDS:
            ctx = _handle(pkg, GOLD, n_dev)
            G = pkg.matrix_multiplication.G.new(ctx, n, A, B, [int(x) for x in pt])
            assert np.array_equal(G.f_a.to_evaluations(), fa) and np.array_equal(G.f_b.to_evaluations(), fb), (n, n_dev)
            c1, evals, _ = pkg.matrix_multiplication.prove(ctx, G, pyref.SEED_R)
            assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]), (n, n_dev)
            del G
            ctx.close()


@needs_two
def test_the_one_device_multi_handle_suite_spreads_over_the_devices():
    """tests/test_gpu_multi.py names device d % device_count for shard d: on this box its whole suite (W, triangle, GKR end to end,
    interleaved provers, more provers than tail slots ...) already ran over distinct devices - make that visible"""
    from test_gpu_multi import device_list
    assert REHEARSAL or len(set(device_list(NMAX))) == NMAX


# ---- (d) the bench line of a real N > 1 run --------------------------------------------------------------------------------------

@needs_two
def test_bench_line_on_real_devices():
    """python -m torch.distributed.run ... bench.py --gpus N exactly as the driver launches it, nothing overridden: one JSON line,
    every data plane timed on real devices (the RCCL plane by the real library), roofline and cpu_baseline present"""
    env = _env("peer")
    env.pop("SC_WORKER_DISTINCT_DEVICES")
    env.pop("SC_WORKER_TRANSPORT")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(NMAX), "--master-addr", "127.0.0.1",
           "--master-port", str(31550 + (os.getpid() % 40)), os.path.join(ROOT, "bench.py"), "--gpus", str(NMAX), "--steps", "5", "--warmup", "2",
           "--num-vars", "24", "--cpu-num-vars", "24"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert out.returncode == 0, _tail(out)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    tr = d["config"]["transports"]
    assert d["n_gpus"] == NMAX and set(tr) == {"peer", "rccl", "inproc"}
    for plane in ("peer", "rccl", "inproc"):
        assert tr[plane]["ms_per_step"] and tr[plane]["comm_nranks"] == NMAX, (plane, tr[plane])
    assert REHEARSAL or "double" not in str(tr["rccl"].get("library", ""))
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1
    assert "%d-rank transcript is bit-exact vs the CPU oracle at n=24" % NMAX in d["config"]["parity_gate"]
