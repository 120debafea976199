"""wfold_pass_kernel (kernels/grid_pass.hpp): the fold behind the matrix-core first pass that serves FIVE rounds - four pending
challenges folded, the 243 cells of the next five rounds left like a grid pass's.  Transcripts bit for bit against the CPU oracle
for every kind of modulus, table sizes from one tile (2^12 entries) up, block caps that make one block walk many tiles and many
blocks share few, the pass on shards (multi-device handle), round-by-round use, and the planner's choice of it."""
import numpy as np
import pytest

from conftest import load_package
from util import GOLD, challenges, oracle, pid, pyref

pytestmark = pytest.mark.gpu

P59 = 2**64 - 59


def plan_str(plan):
    return " ".join("%s(%d,%d)@%d" % (s["action"], s["kf"], s["ks"], s["log_in"]) for s in plan)


def run_vs_oracle(pkg, p, n, opts, expect_wfold=True, devices=None):
    o = oracle(p)
    ha, hb = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ref = o.prove(ha, hb, challenges(o, n))
    assert ref["status"] == 0
    ctx = pkg.Context(pkg.Field(p), devices=devices) if devices else pkg.Context(pkg.Field(p))
    for k, v in opts.items():
        ctx.set_option(k, v)
    a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, ha)
    b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, hb)
    g = pkg.matrix_multiplication.G(a, b)
    if not devices:
        ctx.set_option("time_kernels", 1)
        ctx.launch_log(reset=True)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    if not devices:
        kinds = [r["kind"] for r in ctx.launch_log(reset=True)]
        ctx.set_option("time_kernels", 0)
        assert ("wfold_pass" in kinds) == expect_wfold, (kinds, n, opts)
    assert c1 == ref["c_1"], (p, n, opts)
    assert np.array_equal(evals, ref["evals"]), (p, n, opts)
    assert g.evaluate([int(x) for x in ch]) == ref["final_eval"]
    # round by round (Prover::round): rounds 4..8 come out of the wfold pass's cache
    pr = g.native_prover()
    assert pr.c1() == ref["c_1"]
    for j in range(min(n, 11)):
        assert pr.round_evals(int(ch[j - 1]) if j else ctx.field.one, j) == [int(x) for x in ref["evals"][j]], (p, n, opts, j)
    del pr, g, a, b
    ctx.close()


@pytest.mark.parametrize("p", [GOLD, P59, 389, 5, 2**61 - 1], ids=pid)
@pytest.mark.parametrize("n,opts", [(16, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1}),      # one block, 16 tiles
                                    # the (5, ks) form behind it on the small tables: (5, 4) on ONE tile, (5, 5) on 4 / 8 tiles
                                    (16, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1, "host_tail_log": 0, "wfold5_min_log": 12}),
                                    (19, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1, "host_tail_log": 0, "wfold5_min_log": 12, "max_blocks": 3}),
                                    # the streaming form where NO grid pass could take the five challenges (n >= 30 by default; here grid_log 8)
                                    (20, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold5_min_log": 12, "grid_log": 8}),
                                    (21, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold5_min_log": 12, "grid_log": 6, "host_tail_log": 0}),
                                    (18, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1, "max_blocks": 3}),
                                    (18, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1, "host_tail_log": 0}),   # the device serves every round
                                    (20, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1}),
                                    (21, {}), (22, {"max_blocks": 5}), (22, {"nt_load_log": 12, "nt_store_log": 12}), (23, {"max_blocks": 64})])
def test_wfold_pass_vs_oracle(p, n, opts):
    pkg = load_package()
    popts = {k: v for k, v in opts.items() if k in ("first_pass_vars", "wfold_min_log", "wfold_always", "host_tail_log", "wfold5_min_log", "grid_log")}
    plan = pkg.schedule.plan_proof(n, **popts)
    assert plan[1] == {"action": "wfold_pass", "kf": 4, "ks": 5, "log_in": n, "sharded": False}, plan_str(plan)
    if "wfold5_min_log" in opts:
        assert plan[2]["action"] == "wfold_pass" and plan[2]["kf"] == 5 and plan[2]["ks"] in (3, 4, 5), plan_str(plan)
    run_vs_oracle(pkg, p, n, opts)


@pytest.mark.parametrize("p", [GOLD, P59], ids=pid)
def test_wfold_off_is_the_two_round_fold(p):
    """wfold_log = 0: the schedule of round 4 (pass_kernel<4,2> or a grid pass behind the gram pass), same transcript"""
    pkg = load_package()
    assert "wfold_pass" not in plan_str(pkg.schedule.plan_proof(22, wfold_log=0))
    run_vs_oracle(pkg, p, 22, {"wfold_log": 0}, expect_wfold=False)


@pytest.mark.parametrize("p", [GOLD, P59], ids=pid)
@pytest.mark.parametrize("n,devs", [(24, 8), (23, 2), (24, 4)])
def test_wfold_on_the_shards_of_a_handle(p, n, devs):
    """shards of 2^21 / 2^22 entries on a multi-device handle (entries of device 0): gram pass + wfold pass on every shard, the
    host adds the shards' cells"""
    pkg = load_package()
    plan = pkg.schedule.plan_proof(n, devs, "local")
    assert plan[1]["action"] == "wfold_pass" and plan[1]["sharded"], plan_str(plan)
    run_vs_oracle(pkg, p, n, {}, devices=[0] * devs)


def test_planner_takes_wfold_where_it_pays():
    """(the planner's rules as pure host logic: tests/test_schedule_cpu.py; here the default launches of real proofs)"""
    pkg = load_package()
    pp = pkg.schedule.plan_proof
    assert plan_str(pp(25)) == "gram_pass(0,4)@25 wfold_pass(4,5)@25 grid_pass(5,5)@21 grid_pass(5,4)@16 host_tail(4,7)@11"
    assert plan_str(pp(28)) == "gram_pass(0,4)@28 wfold_pass(4,5)@28 wfold_pass(5,3)@24 grid_pass(3,5)@19 grid_pass(5,4)@16 host_tail(4,7)@11"
    assert plan_str(pp(26)) == "gram_pass(0,4)@26 wfold_pass(4,5)@26 grid_pass(5,5)@22 grid_pass(5,4)@17 host_tail(4,8)@12"
    assert pp(26, host_tail_log=11)[1] == {"action": "pass", "kf": 4, "ks": 2, "log_in": 26, "sharded": False}
    for n in range(1, 21):
        assert "wfold_pass" not in plan_str(pp(n)), n
    ctx = pkg.Context(pkg.Field(GOLD))
    ctx.set_option("time_kernels", 1)
    for n in (24, 25, 26):
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        ctx.launch_log(reset=True)
        pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        log = [(r["kind"], r["kf"], r["ks"], r["log_in"]) for r in ctx.launch_log(reset=True)]
        assert log == [(s["action"], s["kf"], s["ks"], s["log_in"]) for s in pp(n) if s["action"] != "host_tail"], (n, log)
        del g, a, b
    ctx.close()
