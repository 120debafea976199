"""The four-round first pass on the int8 matrix cores (kernels/gram.hpp: gram_pass_kernel + gram_finish_kernel) and the
four-variable fold pass behind it (pass_kernel<4, 2>): transcripts bit for bit against the oracle for every kind of modulus
(the kernel never sees p), adversarial byte patterns (the signed-byte correction), partial counts, the schedules that lead
through pass_kernel<4, 2>, and the default schedule at the sizes where it is chosen by itself."""
import numpy as np
import pytest

from conftest import load_package
from util import GOLD, challenges, oracle, pid, pyref, verifier_identities

pytestmark = pytest.mark.gpu

P59 = 2**64 - 59


def prove_vs_oracle(pkg, ctx, p, n, ha, hb, tag):
    o = oracle(p)
    a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, ha)
    b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, hb)
    g = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    ref = o.prove(ha, hb, ch)
    assert ref["status"] == 0
    assert c1 == ref["c_1"], tag
    assert np.array_equal(evals, ref["evals"]), tag
    assert g.evaluate([int(x) for x in ch]) == ref["final_eval"], tag
    # round by round as well (Prover::round): the cache of the gram pass serves rounds 0..3
    pr = g.native_prover()
    assert pr.c1() == ref["c_1"]
    for j in range(min(n, 7)):
        assert pr.round_evals(int(ch[j - 1]) if j else ctx.field.one, j) == [int(x) for x in ref["evals"][j]], (tag, j)


@pytest.mark.parametrize("p", [GOLD, P59, 389, 5, 2**61 - 1], ids=pid)
@pytest.mark.parametrize("n,opts", [(14, {}), (15, {}), (17, {"max_blocks": 3}), (18, {"grid_log": 10}), (20, {"grid_log": 12}),
                                    (20, {"max_blocks": 64}), (21, {"grid_log": 9, "max_blocks": 7}),
                                    # several partials per block (an int32 accumulator takes 2^16 rows: normally from n = 29 up): the block
                                    # reduces each on its own and adds the entries mod p - one block walking two, two blocks walking two each
                                    (21, {"max_blocks": 1}), (22, {"max_blocks": 2})])
def test_gram_first_pass_vs_oracle(p, n, opts):
    """first_pass_vars = 4 asks for the matrix-core pass at any size; a small grid_log sends the pass behind it - four pending
    challenges - to pass_kernel<4, 2> instead of wgrid_pass_kernel"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    ctx.set_option("first_pass_vars", 4)
    for k, v in opts.items():
        ctx.set_option(k, v)
    plan = pkg.schedule.plan_proof(n, first_pass_vars=4, **{k: v for k, v in opts.items() if k != "max_blocks"})
    assert plan[0]["action"] == "gram_pass" and plan[0]["ks"] == 4
    if "grid_log" in opts:
        assert plan[1] == {"action": "pass", "kf": 4, "ks": 2, "log_in": n, "sharded": False}
    o = oracle(p)
    ha, hb = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    prove_vs_oracle(pkg, ctx, p, n, ha, hb, (p, n, opts))
    ctx.close()


@pytest.mark.parametrize("p", [GOLD, P59], ids=pid)
@pytest.mark.parametrize("kind", ["ff", "zero_a", "x80", "x7f", "spike", "alternating"])
def test_gram_byte_patterns(p, kind):
    """the signed-byte correction at its corners: bytes 0xFF / 0x80 / 0x7F / 0x00 everywhere (entries reduced mod p first where
    they have to be), one non-zero entry, and 0x00 / 0xFF alternating by entry"""
    pkg = load_package()
    n = 16
    size = 1 << n
    rng = np.random.default_rng(5)
    if kind == "ff":
        ha = np.full(size, p - 1, dtype=np.uint64)
        hb = np.full(size, p - 1, dtype=np.uint64)
    elif kind == "zero_a":
        ha = np.zeros(size, dtype=np.uint64)
        hb = rng.integers(0, 2**63, size=size, dtype=np.uint64) % np.uint64(p)
    elif kind == "x80":
        ha = np.full(size, 0x8080808080808080 % p, dtype=np.uint64)
        hb = np.full(size, 0x8080808080808080 % p, dtype=np.uint64)
    elif kind == "x7f":
        ha = np.full(size, 0x7F7F7F7F7F7F7F7F, dtype=np.uint64)
        hb = np.full(size, 0x7F7F7F7F7F7F7F7F, dtype=np.uint64)
    elif kind == "spike":
        ha = np.zeros(size, dtype=np.uint64)
        hb = np.zeros(size, dtype=np.uint64)
        ha[12345] = p - 2
        hb[12345] = p - 3
        hb[12344] = 77
    else:
        ha = np.where(np.arange(size) % 2 == 0, 0, p - 1).astype(np.uint64)
        hb = np.where(np.arange(size) % 3 == 0, p - 1, 0).astype(np.uint64)
    ctx = pkg.Context(pkg.Field(p))
    ctx.set_option("first_pass_vars", 4)
    ctx.set_option("grid_log", 8)
    prove_vs_oracle(pkg, ctx, p, n, np.ascontiguousarray(ha), np.ascontiguousarray(hb), (p, kind))
    ctx.close()


def test_gram_is_the_default_from_2_21_and_agrees_with_the_three_round_schedule():
    """n = 28: the default schedule (tables and shards of >= 2^21 entries, profiles/r04_gram_vs_27cell.txt) opens with the gram pass;
    its transcript equals the one of the 27-cell first pass (gram_log = 0) bit for bit, and the verifier's identities hold"""
    pkg = load_package()
    F = pkg.Field(GOLD)
    n = 28
    assert pkg.schedule.plan_proof(n)[0]["action"] == "gram_pass" and pkg.schedule.plan_proof(21)[0]["action"] == "gram_pass"
    assert pkg.schedule.plan_proof(n, gram_log=0)[0] == {"action": "pass", "kf": 0, "ks": 3, "log_in": n, "sharded": False}
    assert pkg.schedule.plan_proof(20)[0]["action"] == "grid_pass"
    assert [s["action"] for s in pkg.schedule.plan_proof(28, 8, "peer")][:2] == ["gram_pass", "pass"]      # 2^25-entry shards
    out = []
    for gram_log in (21, 0):
        ctx = pkg.Context(F)
        ctx.set_option("gram_log", gram_log)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        ctx.set_option("time_kernels", 1)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        kinds = [r["kind"] for r in ctx.launch_log()]
        assert ("gram_pass" in kinds) == (gram_log != 0) and "gram_finish" not in kinds, kinds      # (one launch since round 5)
        final = g.evaluate([int(x) for x in ch])
        assert verifier_identities(F, c1, evals, ch, final) is None
        out.append((c1, evals, ch, final))
        del a, b, g
        ctx.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and out[0][3] == out[1][3]


def test_gram_on_a_one_device_handle_and_generic_field_n24():
    """a handle over one device takes the plain schedule (gram pass included); the generic modulus at n = 24 against the oracle"""
    pkg = load_package()
    n = 24
    o = oracle(P59)
    ha, hb = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ref = o.prove(ha, hb, challenges(o, n))
    for devices in (None, [0]):
        ctx = pkg.Context(pkg.Field(P59), devices=devices) if devices else pkg.Context(pkg.Field(P59))
        ctx.set_option("gram_log", 24)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]), devices
        del a, b, g
        ctx.close()


@pytest.mark.parametrize("p", [GOLD, P59, 389], ids=pid)
@pytest.mark.parametrize("n", [13, 16, 20])
def test_pipelined_three_variable_fold(p, n):
    """pass_kernel<3,2> in its pipelined whole-tile form (option pipe32, default on from 2^20-entry tables; pipe32_log = 11 brings it
    down to the smallest table it takes) against the staged form and the oracle"""
    pkg = load_package()
    o = oracle(p)
    ha, hb = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ref = o.prove(ha, hb, challenges(o, n))
    for pipe in (1, 0):
        ctx = pkg.Context(pkg.Field(p))
        for k, v in (("gram_log", 0), ("first_pass_vars", 3), ("grid_log", 7), ("pipe32", pipe), ("pipe32_log", 11)):
            ctx.set_option(k, v)
        plan = pkg.schedule.plan_proof(n, gram_log=0, first_pass_vars=3, grid_log=7)
        assert plan[1] == {"action": "pass", "kf": 3, "ks": 2, "log_in": n, "sharded": False}
        a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, ha)
        b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, hb)
        g = pkg.matrix_multiplication.G(a, b)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]), (p, n, pipe)
        ctx.close()


@pytest.mark.parametrize("world,transport", [(2, "host"), (4, "host"), (8, "host"), (2, "peer")])
@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_gram_pass_on_shards(p, world, transport):
    """a sharded prover's shards take the matrix-core first pass too (its cells travel as a grid pass's cells do: summed by the
    host callbacks, or exchanged inside gram_finish_kernel on the peer transport): every rank's transcript equals the oracle's"""
    from test_gpu_sharded import run_virtual_ranks
    pkg = load_package()
    n = 14 + world.bit_length() - 1 + 3            # shards of 2^17 entries
    o = oracle(p)
    oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ref = o.prove(oa, ob, challenges(o, n))
    plan = pkg.schedule.plan_proof(n, world, transport, first_pass_vars=4)
    assert plan[0]["action"] == "gram_pass" and plan[0]["sharded"] and plan[0]["log_in"] == 17
    results, _ = run_virtual_ranks(pkg, p, n, world, tail_log=5, vpp=4, transport=transport)
    for rank, (c1, evals, ch, final, e0, s0) in enumerate(results):
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]) and final == ref["final_eval"], (rank, world, transport)


@pytest.mark.parametrize("n_dev", [2, 8])
def test_gram_pass_on_a_multi_device_handle(n_dev):
    pkg = load_package()
    n = 19
    o = oracle(P59)
    ha, hb = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ctx = pkg.Context(pkg.Field(P59), devices=[0] * n_dev)
    ctx.set_option("first_pass_vars", 4)
    assert pkg.schedule.plan_proof(n, n_dev, "local", first_pass_vars=4)[0]["action"] == "gram_pass"
    prove_vs_oracle(pkg, ctx, P59, n, ha, hb, ("handle", n_dev))
    ctx.close()


@pytest.mark.parametrize("n", [14, 17, 20])
def test_four_variable_fold_in_both_forms(n):
    """pass_kernel<4,2>: the LDS-DMA form (option fold_dma, default on for Goldilocks) and the register-staged pipelined form
    against the oracle"""
    pkg = load_package()
    o = oracle(GOLD)
    ha, hb = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    ref = o.prove(ha, hb, challenges(o, n))
    for dma in (1, 0):
        ctx = pkg.Context(pkg.Field(GOLD))
        for k, v in (("first_pass_vars", 4), ("grid_log", 8), ("fold_dma", dma), ("max_blocks", 256 if dma else 5)):
            ctx.set_option(k, v)
        a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, ha)
        b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, hb)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, pkg.matrix_multiplication.G(a, b), pyref.SEED_R)
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]), (n, dma)
        ctx.close()
