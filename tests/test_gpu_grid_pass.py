"""The passes on the smaller tables (wgrid_pass_kernel: up to five pending challenges folded, up to five
rounds served per launch, folded tables of <= 2^20 entries): every schedule the planner can produce gives the reference's transcript
(sum-check-protocol/src/lib.rs:105-112 one round at a time), bit for bit against the C oracle."""
import numpy as np
import pytest

from conftest import load_package
from util import GOLD, challenges, oracle, pid, pyref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


def prove_and_check(pkg, ctx, o, n, seed_shift=0):
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A + seed_shift, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B + seed_shift, n)
    g = pkg.matrix_multiplication.G(a, b)
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    log = ctx.launch_log(reset=True)
    ctx.set_option("time_kernels", 0)
    ref = o.prove(o.generate(pyref.SEED_A + seed_shift, n), o.generate(pyref.SEED_B + seed_shift, n), ch)
    assert ref["status"] == 0
    assert c1 == ref["c_1"], n
    assert np.array_equal(evals, ref["evals"]), n
    assert g.evaluate([int(x) for x in ch]) == ref["final_eval"], n
    return log


@pytest.mark.parametrize("p", [GOLD, 389, 2**64 - 59], ids=pid)
def test_default_schedule_uses_grid_passes(pkg, p):
    ctx = pkg.Context(pkg.Field(p))
    o = oracle(p)
    for n in list(range(1, 19)) + [20, 21, 22]:
        # the default: the launches are the plan's, the host serves what the last launch leaves (option "host_tail_log")
        ctx.set_option("host_tail_log", 12)
        log = prove_and_check(pkg, ctx, o, n)
        plan = pkg.schedule.plan_proof(n)
        assert [(r["kind"], r["kf"], r["ks"]) for r in log] == [(s["action"], s["kf"], s["ks"]) for s in plan if s["action"] != "host_tail"], (n, log)
        assert sum(r["ks"] for r in log) == n - (plan[-1]["ks"] if plan[-1]["action"] == "host_tail" else 0)
        assert (plan[-1]["action"] == "host_tail") == (n > 10), (n, plan)
        # the device alone (round 4's schedule)
        ctx.set_option("host_tail_log", 0)
        log = prove_and_check(pkg, ctx, o, n)
        kinds = [r["kind"] for r in log]
        # tables of <= 2^20 entries are proved by grid passes alone; larger ones end with them
        assert kinds[-1] == "grid_pass", (n, kinds)
        if n <= 20:
            assert set(kinds) == {"grid_pass"}, (n, kinds)
            assert len(log) == (n + 4) // 5, (n, log)                        # five rounds per launch
        assert sum(r["ks"] for r in log) == n, (n, log)   # every round served exactly once
        assert all(r["kf"] <= 5 and 1 <= r["ks"] <= 5 for r in log)
    ctx.close()


@pytest.mark.parametrize("opts", [
    {"grid_max_vars": 1}, {"grid_max_vars": 2}, {"grid_max_vars": 3}, {"grid_max_vars": 4},
    {"grid_log": 3}, {"grid_log": 8}, {"grid_log": 14, "grid_max_vars": 4}, {"grid_log": 26},
    {"grid_blocks": 1}, {"grid_blocks": 3}, {"grid_blocks": 33}, {"grid_blocks": 70, "grid_max_vars": 3},
    {"grid_pass": 0}, {"grid_pass": 0, "first_pass_vars": 2},
    # the device alone down to the last round (host_tail_log 0), and host tails from other sizes
    {"host_tail_log": 0}, {"host_tail_log": 0, "grid_max_vars": 3}, {"host_tail_log": 0, "grid_blocks": 3}, {"host_tail_log": 0, "grid_pass": 0},
    {"host_tail_log": 6}, {"host_tail_log": 2, "grid_max_vars": 2}, {"host_tail_log": 10, "grid_blocks": 2},
], ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()))
def test_every_grid_schedule_matches_the_oracle(pkg, opts):
    for p in (GOLD, 1572869):
        ctx = pkg.Context(pkg.Field(p))
        for k, v in opts.items():
            ctx.set_option(k, v)
            assert ctx.get_option(k) == v
        o = oracle(p)
        for n in (1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 13, 14, 15, 17, 19, 21):
            log = prove_and_check(pkg, ctx, o, n, seed_shift=n)
            if opts.get("grid_pass", 1) == 0:
                assert all(r["kind"] != "grid_pass" for r in log)
            if "grid_max_vars" in opts:
                assert all(r["ks"] <= opts["grid_max_vars"] for r in log if r["kind"] == "grid_pass")
            if "grid_log" in opts:
                assert all(r["log_in"] - r["kf"] <= opts["grid_log"] for r in log if r["kind"] == "grid_pass")
        ctx.close()


def test_round_by_round_api_over_grid_passes(pkg):
    """Prover::round one round at a time (the Rust shim's call sequence): answers come from the cached grid of a
    five-round pass; challenges arrive one by one"""
    scp = pkg.sum_check_protocol
    for p in (GOLD, 389):
        ctx = pkg.Context(pkg.Field(p))
        o = oracle(p)
        for n in (5, 6, 10, 11, 12, 16):
            a = pkg.DenseMultilinearExtension.generate(ctx, 71 + n, n)
            b = pkg.DenseMultilinearExtension.generate(ctx, 72 + n, n)
            g = pkg.matrix_multiplication.G(a, b)
            ch = challenges(o, n)
            ref = o.prove(o.generate(71 + n, n), o.generate(72 + n, n), ch)
            prover = scp.Prover.new(g.clone())
            assert prover.c_1() == ref["c_1"]
            r_prev = ctx.field.one
            for j in range(n):
                poly = prover.round(r_prev, j)
                F = ctx.field
                e = [poly.evaluate(x) for x in (0, F.one, F.add(F.one, F.one))]
                assert e == [int(x) for x in ref["evals"][j]], (p, n, j)
                r_prev = int(ch[j])
        ctx.close()


def test_two_provers_interleaved_on_one_context(pkg):
    """two provers of different sizes on one context, their rounds interleaved: each keeps its own copy of the cells
    of its last pass (the wide mailbox and the ticket counters are shared and reused by every launch)"""
    scp = pkg.sum_check_protocol
    p = GOLD
    ctx = pkg.Context(pkg.Field(p))
    o = oracle(p)
    F = ctx.field
    pts = (0, F.one, F.add(F.one, F.one))
    sizes = (13, 17)
    gs, refs, chs, provers = [], [], [], []
    for n in sizes:
        a = pkg.DenseMultilinearExtension.generate(ctx, 31 + n, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, 32 + n, n)
        gs.append(pkg.matrix_multiplication.G(a, b))
        chs.append(challenges(o, n))
        refs.append(o.prove(o.generate(31 + n, n), o.generate(32 + n, n), chs[-1]))
        provers.append(scp.Prover.new(gs[-1].clone()))
    for k, pr in enumerate(provers):
        assert pr.c_1() == refs[k]["c_1"]
    for j in range(max(sizes)):
        for k, pr in enumerate(provers):
            if j < sizes[k]:
                poly = pr.round(int(chs[k][j - 1]) if j else F.one, j)
                assert [poly.evaluate(x) for x in pts] == [int(x) for x in refs[k]["evals"][j]], (k, j)
    ctx.close()


def test_extreme_words_through_grid_passes(pkg):
    """largest and smallest Montgomery words, challenges p-1, p-2, p-3: the five-challenge fold weights and the
    243-cell lazy sums at their extremes"""
    p = GOLD
    ctx = pkg.Context(pkg.Field(p))
    o = oracle(p)
    for n, pattern in [(5, [p - 1]), (10, [p - 1, p - 2]), (13, [p - 1, 0, 1, p - 1, p - 2, 0, 0xFFFFFFFF, 1 << 32]),
                       (14, [p - 1, p - 1, p - 1, 0]), (16, [p - 1]), (20, [p - 1, p - 2, 0])]:
        words = np.array([pattern[i % len(pattern)] for i in range(1 << n)], dtype=np.uint64)
        other = words[::-1].copy()
        a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, words)
        b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, other)
        g = pkg.matrix_multiplication.G(a, b)
        ch = np.array([p - 1 - (j % 3) for j in range(n)], dtype=np.uint64)
        it = iter(ch)
        c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, 0, draw=lambda _u, _j, _e: int(next(it)))
        ref = o.prove(words, other, ch)
        assert ref["status"] == 0 and c1 == ref["c_1"], n
        assert np.array_equal(evals, ref["evals"]), n
    ctx.close()


def test_random_schedules(pkg):
    """seeded fuzz over the schedule options (all of them legal): whatever mix of pass widths, grid limits and block
    counts the planner is given, the transcript is the oracle's"""
    import random
    rng = random.Random(20260)
    for it in range(48):
        p = rng.choice([GOLD, GOLD, 389, 2**64 - 59])
        ctx = pkg.Context(pkg.Field(p))
        opts = {"grid_pass": rng.choice([1, 1, 1, 0]), "grid_log": rng.randrange(0, 23), "grid_max_vars": rng.randrange(1, 6),
                "grid_blocks": rng.choice([0, 0, 1, 2, 5, 31, 32, 33, 64, 100, 1024]), "first_pass_vars": rng.choice([0, 0, 1, 2, 3]),
                "max_blocks": rng.choice([7, 64, 768])}
        for k, v in opts.items():
            ctx.set_option(k, v)
        o = oracle(p)
        for n in rng.sample(range(1, 23), 4):
            try:
                prove_and_check(pkg, ctx, o, n, seed_shift=it)
            except AssertionError as e:
                raise AssertionError("options %r, p=%d, n=%d: %s" % (opts, p, n, e))
        ctx.close()


def test_option_ranges(pkg):
    ctx = pkg.Context(pkg.Field(GOLD))
    for k, bad in [("grid_log", 27), ("grid_max_vars", 0), ("grid_max_vars", 6),
                   ("grid_blocks", -1), ("grid_blocks", 1025)]:
        with pytest.raises(Exception):
            ctx.set_option(k, bad)
    ctx.close()
