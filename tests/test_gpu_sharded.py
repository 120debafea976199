"""The sharded (one rank per GPU) prover on ONE GPU: N virtual ranks = N threads, each with
its own context/stream on device 0 and its own shard, joined by in-process host collectives.
The control flow per rank is the one the 8-GPU run uses; only the transport differs.  Also
the RCCL transport itself with world = 1."""
import threading

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, challenges, oracle, pid, pyref

pytestmark = pytest.mark.gpu



class Loopback:
    """sum / concatenate across `world` threads"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.n_allreduce = 0
        self.n_allgather = 0

    def collectives(self, rank):
        def allreduce(arr):
            self.slots[rank] = arr.copy()
            self.barrier.wait()
            total = np.zeros_like(arr)
            for s in self.slots:
                total += s
            self.barrier.wait()
            arr[:] = total
            if rank == 0:
                self.n_allreduce += 1

        def allgather(send):
            self.slots[rank] = send.copy()
            self.barrier.wait()
            out = np.concatenate(self.slots)
            self.barrier.wait()
            if rank == 0:
                self.n_allgather += 1
            return out

        return allreduce, allgather


def run_virtual_ranks(pkg, p, n, world, tail_log, vpp, what="prove", transport="host", extra=None):
    lb = Loopback(world)
    results = [None] * world
    errors = []
    ctxs = [None] * world

    def body(rank):
        try:
            ctx = pkg.Context(pkg.Field(p))
            ctx.set_option("tail_log", tail_log)
            ctx.set_option("vars_per_pass", min(vpp, 2))
            ctx.set_option("first_pass_vars", vpp if vpp >= 3 else min(vpp, 2))    # (4: the matrix-core first pass on the shards)
            ctx.set_option("grid_pass", 1 if vpp >= 3 else 0)
            for k, v in (extra or {}).items():
                ctx.set_option(k, v)
            if transport == "peer":
                # in-kernel exchange through peer-mapped inboxes; threads of one process share the address space
                ctx.set_option("peer_spin_ms", 20000)
                ctx.comm_peer_export(rank, world)
                ctxs[rank] = ctx
                lb.barrier.wait()
                ctx.comm_peer_connect_local(ctxs)
                lb.barrier.wait()
            else:
                ar, ag = lb.collectives(rank)
                ctx.comm_init_host(rank, world, ar, ag)
            start, length = pkg.distributed.shard_range(n, rank, world)
            nl = length.bit_length() - 1
            a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, nl, start=start)
            b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, nl, start=start)
            g = pkg.matrix_multiplication.G(a, b)
            assert g.num_vars() == n
            c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
            final = g.evaluate([int(x) for x in ch])
            e0 = g.round_evals() if nl >= 1 else None
            s0 = g.hypercube_sum()
            results[rank] = (c1, evals, ch, final, e0, s0)
            if transport == "peer":
                lb.barrier.wait()      # nobody unmaps a region a peer's kernel may still write
            ctx.close()
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errors.append(e)
            lb.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    return results, lb


@pytest.mark.parametrize("vpp", [1, 2, 3])
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_virtual_ranks_match_oracle(p, world, vpp):
    pkg = load_package()
    import thaler_study_amd.distributed  # noqa: F401
    o = oracle(p)
    g = world.bit_length() - 1
    for n, tail_log in [(g, 0), (g + 1, 0), (g + 2, 0), (g + 5, 0), (12, 0), (12, 5), (15, 12), (16, 3)]:
        oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
        ch = challenges(o, n)
        ref = o.prove(oa, ob, ch)
        results, lb = run_virtual_ranks(pkg, p, n, world, tail_log, vpp)
        for rank, (c1, evals, chn, final, e0, s0) in enumerate(results):
            assert c1 == ref["c_1"], (n, tail_log, rank)
            assert np.array_equal(evals, ref["evals"]), (n, tail_log, rank)
            assert final == ref["final_eval"], (n, tail_log, rank)
            assert s0 == ref["c_1"]
            if e0 is not None:
                assert e0 == [int(x) for x in ref["evals"][0]]
        # with tail_log 0 and enough local variables the run really used per-pass all-reduces
        if tail_log == 0 and n - g >= 4:
            assert lb.n_allreduce >= 2
        assert lb.n_allgather >= 2  # the tail gather of both tables


WFOLD_ON_SMALL_SHARDS = {"wfold_min_log": 12, "wfold_always": 1, "wfold5_min_log": 12}


@pytest.mark.parametrize("transport,world", [("host", 2), ("host", 8), ("peer", 2)])
@pytest.mark.parametrize("p", [GOLD, 2**64 - 59], ids=pid)
def test_virtual_ranks_wfold_pass_on_the_shards(p, transport, world):
    """wfold_pass_kernel on the shards of the rank transports (its cells cross the ranks as a grid pass's: split limbs through the
    collective / the in-kernel exchange): matrix-core first pass, (4, 5) fold, and the (5, ks) form behind it, on shards of
    2^17 .. 2^19 entries - every rank's transcript is the oracle's"""
    pkg = load_package()
    o = oracle(p)
    g = world.bit_length() - 1
    for n in (20, 21):
        plan = pkg.schedule.plan_proof(n, world, transport, first_pass_vars=4, tail_log=0, **WFOLD_ON_SMALL_SHARDS)     # (tail_log 0: no early gather)
        assert [s["action"] for s in plan][:3] == ["gram_pass", "wfold_pass", "wfold_pass"] and all(s["sharded"] for s in plan[:3]), plan
        oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
        ref = o.prove(oa, ob, challenges(o, n))
        results, _ = run_virtual_ranks(pkg, p, n, world, 0, 4, transport=transport, extra=WFOLD_ON_SMALL_SHARDS)
        for rank, (c1, evals, chn, final, e0, s0) in enumerate(results):
            assert c1 == ref["c_1"], (n, rank)
            assert np.array_equal(evals, ref["evals"]), (n, rank)
            assert final == ref["final_eval"] and s0 == ref["c_1"]


def test_rccl_world1():
    """the in-library RCCL transport (ncclCommInitRank / AllReduce / AllGather) with one rank"""
    pkg = load_package()
    import thaler_study_amd.distributed as D
    o = oracle(GOLD)
    for n, tail_log, first in [(14, 4, 0), (14, 4, 3), (10, 12, 0), (3, 0, 3), (3, 0, 2)]:
        ctx = pkg.Context(pkg.Field(GOLD))
        ctx.set_option("tail_log", tail_log)
        ctx.set_option("first_pass_vars", first)   # 3: the 54-limb all-reduce of the 27-cell first pass
        D.attach_rccl(ctx, 0, 1)
        assert ctx.rank_world() == (0, 1)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        ref = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), ch)
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
        assert g.evaluate([int(x) for x in ch]) == ref["final_eval"]
        del a, b, g
        ctx.close()


@pytest.mark.parametrize("world", [2, 8])
def test_virtual_ranks_g_new(world):
    """sharded G::new (row-block shards of A and B) + proof on the result, against
    the oracle's G::new + prover on the whole matrices (BASELINE config 5 shape)"""
    pkg = load_package()
    p = GOLD
    o = oracle(p)
    for n in (3, 5, 7):
        A = o.generate(11, 2 * n)
        B = o.generate(12, 2 * n)
        pt = np.array([o.challenge(pyref.SEED_PT, j) for j in range(2 * n)], dtype=np.uint64)
        fa, fb = o.g_new(n, A, B, pt)
        ch = challenges(o, n)
        ref = o.prove(fa, fb, ch)
        lb = Loopback(world)
        results, errors = [None] * world, []

        def body(rank):
            try:
                ctx = pkg.Context(pkg.Field(p))
                ctx.set_option("tail_log", 0)
                ar, ag = lb.collectives(rank)
                ctx.comm_init_host(rank, world, ar, ag)
                start, length = pkg.distributed.shard_range(2 * n, rank, world)
                nl = length.bit_length() - 1
                At = pkg.DenseMultilinearExtension.generate(ctx, 11, nl, start=start)
                Bt = pkg.DenseMultilinearExtension.generate(ctx, 12, nl, start=start)
                g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, At, Bt, [int(x) for x in pt])
                assert g.num_vars() == n
                s0, l0 = pkg.distributed.shard_range(n, rank, world)
                assert np.array_equal(g.f_a.to_evaluations(), fa[s0:s0 + l0])
                assert np.array_equal(g.f_b.to_evaluations(), fb[s0:s0 + l0])
                c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
                results[rank] = (c1, evals)
                ctx.close()
            except Exception as e:  # pragma: no cover
                import traceback
                traceback.print_exc()
                errors.append(e)
                lb.barrier.abort()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        for c1, evals in results:
            assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])


@pytest.mark.parametrize("world", [2])
@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_virtual_ranks_peer_transport(p, world):
    """the in-kernel exchange (no collective launch): every sharded pass's last workgroup writes its limbs into
    every rank's inbox and sums what the peers wrote; the tail gather goes through the peer-mapped arenas.
    Two ranks only when the ranks are threads of ONE process: a workgroup that waits for a peer needs the
    peer's kernel to run beside it, and one process's streams share a handful of hardware queues (with four
    contexts a waiting kernel can sit in front of the kernel it waits for: measured, bounded by peer_spin_ms).
    One process per rank - the deployment - has no such coupling; tests/test_gpu_00_multiprocess.py runs 2 and 8 processes."""
    pkg = load_package()
    o = oracle(p)
    g = world.bit_length() - 1
    for n, tail_log, vpp in [(g, 0, 2), (g + 1, 0, 3), (g + 3, 0, 2), (12, 0, 3), (12, 5, 2), (16, 3, 3), (18, 12, 3), (20, 16, 3)]:
        oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
        ch = challenges(o, n)
        ref = o.prove(oa, ob, ch)
        results, _ = run_virtual_ranks(pkg, p, n, world, tail_log, vpp, transport="peer")
        for rank, (c1, evals, chn, final, e0, s0) in enumerate(results):
            assert c1 == ref["c_1"], (n, tail_log, rank)
            assert np.array_equal(evals, ref["evals"]), (n, tail_log, rank)
            assert final == ref["final_eval"], (n, tail_log, rank)
            assert s0 == ref["c_1"]
            if e0 is not None:
                assert e0 == [int(x) for x in ref["evals"][0]]

