"""BASELINE's headline sizes under ORACLE parity, in the suite.

Until round 5 the -m gpu tests compared n = 26 / 28 / 30 proofs through verifier identities and schedule-vs-schedule only; the
bit-exact comparison with the CPU oracle at n = 28 lived in bench.py's gate.  Here configs[2] (n = 26, one GPU) and configs[3]
(n = 28: one GPU, and the hypercube over 8 shards) are compared bit for bit - c_1 and every round triple - with
oracle/sc_oracle.c's reference-shaped prover (its all-cores form: same passes, same outputs, seconds instead of a minute), for
Goldilocks and for the generic-modulus kernels (p = 2^64 - 59).  These are the sizes at which gram_pass_kernel runs its full
grid of 256 blocks x 1 024 steps (and, at n = 28, every block's int32 accumulators reach their 2^16 rows), pass_kernel<4,2>
its LDS-DMA form on nontemporal loads and stores, and the last launch hands its tables to the host.

The oracle's tables and transcript of one (p, n) are built once per session (2 x 2 GiB of host memory at n = 28, freed after
the transcript is taken)."""
import functools

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, challenges, oracle, pid, pyref, verifier_identities

pytestmark = pytest.mark.gpu

P59 = 2**64 - 59


@functools.lru_cache(maxsize=None)
def oracle_transcript(p, n):
    """(c_1, evals[n, 3], challenges[n]) of the synthetic instance (SEED_A, SEED_B, SEED_R) by the CPU oracle"""
    o = oracle(p)
    oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ch = challenges(o, n)
    c1, ev = o.prover_run_mt(oa, ob, ch)
    if n <= 20:   # the all-cores form is the reference-shaped one with its loops split: pinned against it where that is cheap
        c1s, evs = o.prover_run(oa, ob, ch)
        assert c1s == c1 and np.array_equal(evs, ev)
    del oa, ob
    return c1, ev, ch


def test_oracle_all_cores_form_is_the_reference_shaped_one():
    for p in (GOLD, P59, 389):
        oracle_transcript(p, 16)


@pytest.mark.parametrize("p,n", [(GOLD, 26), (GOLD, 28), (P59, 26), (P59, 28), (GOLD, 25), (GOLD, 27), (P59, 27)], ids=lambda v: pid(v) if v > 64 else "n%d" % v)
def test_default_schedule_vs_oracle(p, n):
    """configs[2] / configs[3] on one GPU, default options: the transcript is the oracle's, the launches are the plan's"""
    pkg = load_package()
    c1_ref, ev_ref, ch_ref = oracle_transcript(p, n)
    ctx = pkg.Context(pkg.Field(p))
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    log = [(r["kind"], r["kf"], r["ks"], r["log_in"]) for r in ctx.launch_log(reset=True)]
    ctx.set_option("time_kernels", 0)
    assert np.array_equal(ch, ch_ref)
    assert c1 == c1_ref, (p, n)
    bad = [j for j in range(n) if not np.array_equal(evals[j], ev_ref[j])]
    assert not bad, (p, n, "rounds that differ from the oracle", bad)
    plan = pkg.schedule.plan_proof(n)
    assert log == [(s["action"], s["kf"], s["ks"], s["log_in"]) for s in plan if s["action"] != "host_tail"]
    # (behind the matrix-core pass: wfold_pass_kernel - it saves a launch at every one of these sizes; n = 26 ends with a hand-over
    # of 2^12-entry tables, the others of 2^11)
    assert plan[0]["action"] == "gram_pass" and plan[1]["action"] == "wfold_pass" and plan[1]["kf"] == 4
    assert plan[-1]["action"] == "host_tail" and len(plan) == (5 if n in (25, 26) else 6) and plan[-1]["log_in"] == (12 if n == 26 else 11)
    # the verifier's last check against the tables themselves (sum-check-protocol/src/lib.rs:302-307)
    assert verifier_identities(ctx.field, c1, evals, ch, g.evaluate([int(x) for x in ch])) is None
    # a second proof on the same context (warm pool, recycled buffers, the gram ticket back at rest): the same transcript
    c1b, evals_b, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert c1b == c1_ref and np.array_equal(evals_b, ev_ref)
    del a, b, g
    ctx.close()


@pytest.mark.parametrize("opts", [{"host_tail_log": 0}, {"gram_log": 0}, {"fold_dma": 0}, {"vars_per_pass": 1}],
                         ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()))
def test_n28_other_schedules_vs_oracle(opts):
    """the same n = 28 instance through the schedules the default replaced: the device alone to the last round, the 27-cell first
    pass, the register form of the four-variable fold, the reference's one round per pass"""
    pkg = load_package()
    n = 28
    c1_ref, ev_ref, _ = oracle_transcript(GOLD, n)
    ctx = pkg.Context(pkg.Field(GOLD))
    for k, v in opts.items():
        ctx.set_option(k, v)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    c1, evals, _ = pkg.matrix_multiplication.prove(ctx, pkg.matrix_multiplication.G(a, b), pyref.SEED_R)
    assert c1 == c1_ref and np.array_equal(evals, ev_ref), opts
    del a, b
    ctx.close()


@pytest.mark.parametrize("n_dev", [8, 2])
def test_n28_sharded_on_a_handle_vs_oracle(n_dev):
    """configs[3]'s shape - the n = 28 hypercube over 8 (and 2) shards behind one multi-device handle - against the ORACLE's
    transcript, not against the one-device GPU run (tests/test_gpu_multi.py::test_multi_n28_equals_one_device)"""
    from test_gpu_multi import multi_ctx, tables
    pkg = load_package()
    n = 28
    c1_ref, ev_ref, ch_ref = oracle_transcript(GOLD, n)
    ctx = multi_ctx(pkg, GOLD, n_dev)
    a, b = tables(pkg, ctx, n)
    G = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, G, pyref.SEED_R)
    assert c1 == c1_ref and np.array_equal(ch, ch_ref)
    bad = [j for j in range(n) if not np.array_equal(evals[j], ev_ref[j])]
    assert not bad, (n_dev, "rounds that differ from the oracle", bad)
    del G, a, b
    ctx.close()


def test_n28_eight_virtual_ranks_vs_oracle():
    """the same shape as eight RANKS (one context each, host-callback transport: the sums cross the ranks as split limbs, the
    shards are gathered when they are down to their pending challenges) - every rank's transcript is the oracle's"""
    import threading
    from test_gpu_sharded import Loopback
    pkg = load_package()
    n, world = 28, 8
    c1_ref, ev_ref, _ = oracle_transcript(GOLD, n)
    lb = Loopback(world)
    got, errors = [None] * world, []

    def body(rank):
        try:
            ctx = pkg.Context(pkg.Field(GOLD))
            ar, ag = lb.collectives(rank)
            ctx.comm_init_host(rank, world, ar, ag)
            start, length = pkg.distributed.shard_range(n, rank, world)
            a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, length.bit_length() - 1, start=start)
            b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, length.bit_length() - 1, start=start)
            got[rank] = pkg.matrix_multiplication.prove(ctx, pkg.matrix_multiplication.G(a, b), pyref.SEED_R)
            del a, b
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
        t.join(timeout=600)
    assert not errors, errors
    for rank in range(world):
        c1, evals, _ = got[rank]
        assert c1 == c1_ref and np.array_equal(evals, ev_ref), rank
