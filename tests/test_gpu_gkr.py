"""GKR round polynomial W on the GPU against the oracle restatement of
gkr-protocol/src/round_polynomial.rs and the reference's circuit known answers."""
import random

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, oracle, pid, pyref

pytestmark = pytest.mark.gpu

BOOK = [[("mul", 0, 1), ("mul", 2, 3)], [("mul", 0, 0), ("mul", 1, 1), ("mul", 1, 2), ("mul", 3, 3)]]


def make_circuit(pkg, layers, num_inputs):
    gp = pkg.gkr_protocol
    return gp.Circuit([gp.CircuitLayer([gp.Gate(t, [a, b]) for (t, a, b) in layer]) for layer in layers], num_inputs)


def random_circuit(rng, ks):
    """layers[i] has 2^ks[i] gates reading layer i+1 (2^ks[i+1] values); last entry = inputs"""
    layers = []
    for i in range(len(ks) - 1):
        n_next = 1 << ks[i + 1]
        layers.append([(rng.choice(["add", "mul"]), rng.randrange(n_next), rng.randrange(n_next))
                       for _ in range(1 << ks[i])])
    return layers


def test_circuit_from_book():
    """gkr-protocol/src/circuit.rs:259-284"""
    pkg = load_package()
    F = pkg.Field(389)
    c = make_circuit(pkg, BOOK, 4)
    layers = c.evaluate(F, F.from_ints([3, 2, 3, 1]).tolist())
    assert [F.to_ints(l) for l in layers] == [[36, 6], [9, 4, 6, 1], [3, 2, 3, 1]]
    for a in range(4):
        for b in range(4):
            for cc in range(4):
                expected = ((a in (0, 1)) and a == b and a == cc) or (a == 2 and b == 1 and cc == 2) or (a == b == cc == 3)
                assert c.mul_i(1, a, b, cc) == expected
    assert c.num_vars_at(0) == 1 and c.num_vars_at(1) == 2 and c.num_vars_at(2) == 2 and c.num_vars_at(3) is None


@pytest.mark.parametrize("p", [389, GOLD], ids=pid)
def test_w_book_circuit_layers(p):
    """every layer of the book circuit: wiring tables, c_1 = W_i(r_i), all rounds, final evaluation"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    o = oracle(p)
    scp, gp = pkg.sum_check_protocol, pkg.gkr_protocol
    circuit = make_circuit(pkg, BOOK, 4)
    evaluation = circuit.evaluate(F, F.from_ints([3, 2, 3, 1]).tolist())
    rng = random.Random(5)
    for i in (0, 1):
        k_i, k_next = circuit.num_vars_at(i), circuit.num_vars_at(i + 1)
        r_i = [F.from_int(rng.randrange(p)) for _ in range(k_i)]
        w = gp.start_round_w(ctx, circuit, evaluation, i, r_i)
        oadd, omul = o.wiring_fixed(BOOK[i], k_next, r_i)
        assert np.array_equal(w.add_i.to_evaluations(), oadd) and np.array_equal(w.mul_i.to_evaluations(), omul)
        ow = np.array(evaluation[i + 1], dtype=np.uint64)
        ch = [F.from_int(rng.randrange(p)) for _ in range(2 * k_next)]
        ref = o.w_prove(oadd, omul, ow, ow, ch)
        assert ref["status"] == 0
        # c_1 of the layer sumcheck is W_i~(r_i) (Thaler, GKR)
        wi = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, k_i, np.array(evaluation[i], dtype=np.uint64))
        assert ref["c_1"] == wi.evaluate(r_i)
        # trait methods
        assert w.num_vars() == 2 * k_next
        assert np.array_equal(w.to_evaluations(), o.w_to_evaluations(oadd, omul, ow, ow))
        assert w.hypercube_sum(F) == ref["c_1"]
        assert w.round_evals() == [int(x) for x in ref["evals"][0]]
        assert w.evaluate(ch) == ref["final_eval"] and w.evaluate(ch[:-1]) is None
        # Prover / Verifier loop (the verifier has oracle access here, unlike full GKR)
        class Scripted(scp.RngF):
            def __init__(self, vals):
                self.vals = list(vals)

            def draw(self):
                return self.vals.pop(0)

        for use_engine in (True, False):
            g = w.clone()
            if not use_engine:
                g.native_prover = lambda: None     # generic fix_variables -> to_univariate path
            prover = scp.Prover.new(g)
            assert prover.c_1() == ref["c_1"]
            verifier = scp.Verifier.new(w.num_vars(), w)
            verifier.set_c_1(prover.c_1())
            rr = Scripted(ch)
            r_j, final = F.one, None
            for j in range(w.num_vars()):
                g_j = prover.round(r_j, j)
                co = o.interpolate(ref["evals"][j])
                dense = [0, 0, 0]
                for d, cf in g_j.coeffs:
                    dense[d] = cf
                assert dense == [int(x) for x in co], (i, j, use_engine)
                res = verifier.round(g_j, rr)
                if res.is_final():
                    final = res.value
                else:
                    r_j = res.value
            assert final is True


@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_w_random_circuits_vs_oracle(p):
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    o = oracle(p)
    gp = pkg.gkr_protocol
    rng = random.Random(p % 97)
    for ks in ([1, 1], [2, 3], [3, 2], [4, 4], [6, 5], [5, 6], [8, 7]):
        layers = random_circuit(rng, ks)
        circuit = make_circuit(pkg, layers, 1 << ks[-1])
        inputs = [F.from_int(rng.randrange(p)) for _ in range(1 << ks[-1])]
        evaluation = circuit.evaluate(F, inputs)
        i = 0
        k_i, k_next = ks[0], ks[1]
        r_i = [F.from_int(rng.randrange(p)) for _ in range(k_i)]
        w = gp.start_round_w(ctx, circuit, evaluation, i, r_i)
        oadd, omul = o.wiring_fixed(layers[0], k_next, r_i)
        assert np.array_equal(w.add_i.to_evaluations(), oadd) and np.array_equal(w.mul_i.to_evaluations(), omul)
        ow = np.array(evaluation[1], dtype=np.uint64)
        ch = [F.from_int(rng.randrange(p)) for _ in range(2 * k_next)]
        # the sparse (per-gate) prover gives the dense prover's round polynomials, bit for bit
        dense_eng, sparse_eng = w.native_prover(), gp.SparseLayerProver(ctx, circuit, evaluation, i, r_i)
        assert dense_eng.c1() == sparse_eng.c1()
        for j in range(2 * k_next):
            rp = ch[j - 1] if j else F.one
            assert dense_eng.round_evals(rp, j) == sparse_eng.round_evals(rp, j), (ks, j)
        if 2 * k_next <= 12:
            ref = o.w_prove(oadd, omul, ow, ow, ch)
            assert ref["status"] == 0
            eng = w.native_prover()
            assert eng.c1() == ref["c_1"]
            for j in range(2 * k_next):
                assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (ks, j)
            assert w.evaluate(ch) == ref["final_eval"]
            # fix_variables across the b/c boundary
            for k in (1, k_next, k_next + 1, 2 * k_next):
                w2 = w.fix_variables(ch[:k])
                assert w2.num_vars() == 2 * k_next - k
                if k < 2 * k_next:
                    assert w2.evaluate(ch[k:]) == ref["final_eval"]
        else:
            # larger: identities only (the oracle is reference-shaped and slow)
            eng = w.native_prover()
            wi = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, k_i, np.array(evaluation[0], dtype=np.uint64))
            assert eng.c1() == wi.evaluate(r_i)
            claim = eng.c1()
            for j in range(2 * k_next):
                e = eng.round_evals(ch[j - 1] if j else F.one, j)
                assert F.add(e[0], e[1]) == claim
                c = o.interpolate(np.array(e, dtype=np.uint64))
                claim = o.poly2_eval(c, ch[j])
            assert claim == w.evaluate(ch)


def test_restrict_poly():
    """gkr-protocol/src/lib.rs:507-548 ([32, 385, 383] over F_389) and random lines vs pyref"""
    pkg = load_package()
    gp = pkg.gkr_protocol
    kat = __import__("util").load_golden("reference_kats.json")["restrict_poly_389"]
    p = kat["p"]
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    mle = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2, F.from_ints(kat["evals"]))
    poly = gp.restrict_poly(F.from_ints(kat["b"]).tolist(), F.from_ints(kat["c"]).tolist(), mle)
    dense = [0] * 3
    for d, cf in poly.coeffs:
        dense[d] = F.to_int(cf)
    assert dense == kat["expected_coeffs"]                      # -6t^2 - 4t + 32
    ln = gp.line(F, F.from_ints(kat["b"]).tolist(), F.from_ints(kat["c"]).tolist())
    assert [F.to_int(l.evaluate(F.zero)) for l in ln] == kat["b"] and [F.to_int(l.evaluate(F.one)) for l in ln] == kat["c"]
    rng = random.Random(8)
    for q in (389, GOLD):
        ctx = pkg.Context(pkg.Field(q))
        F = ctx.field
        for k in (1, 3, 6, 9):
            ev = [rng.randrange(q) for _ in range(1 << k)]
            b = [rng.randrange(q) for _ in range(k)]
            c = [rng.randrange(q) for _ in range(k)]
            mle = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, k, F.from_ints(ev))
            poly = gp.restrict_poly(F.from_ints(b).tolist(), F.from_ints(c).tolist(), mle)
            expect = pyref.restrict_poly(b, c, ev, q)
            got = [0] * (k + 1)
            for d, cf in poly.coeffs:
                got[d] = F.to_int(cf)
            while len(got) > 1 and got[-1] == 0:
                got.pop()
            assert got == expect, (q, k)


def test_wiring_collisions_and_errors():
    """many gates wired to the same (b, c): the scatter's compare-and-swap path must sum them;
    malformed gate lists and inconsistent tables are rejected with SC_ERR_ARG"""
    pkg = load_package()
    p = GOLD
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    o = oracle(p)
    gp = pkg.gkr_protocol
    rng = random.Random(21)
    k_i, k_next = 7, 2
    layers = [[(("add" if a % 3 else "mul"), a % 2, 1) for a in range(1 << k_i)]]   # 2 distinct targets
    circuit = make_circuit(pkg, layers, 1 << k_next)
    r_i = [F.from_int(rng.randrange(p)) for _ in range(k_i)]
    add_d, mul_d = gp.wiring(ctx, circuit, 0, r_i)
    oadd, omul = o.wiring_fixed(layers[0], k_next, r_i)
    assert np.array_equal(add_d.to_evaluations(), oadd) and np.array_equal(mul_d.to_evaluations(), omul)
    assert int(np.count_nonzero(oadd)) <= 2 and int(np.count_nonzero(omul)) <= 2
    inputs = [F.from_int(rng.randrange(p)) for _ in range(1 << k_next)]
    evaluation = circuit.evaluate(F, inputs)
    dense, sparse = gp.start_round_w(ctx, circuit, evaluation, 0, r_i).native_prover(), gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i)
    ch = [F.from_int(rng.randrange(p)) for _ in range(2 * k_next)]
    ref = o.w_prove(oadd, omul, np.array(evaluation[1], dtype=np.uint64), np.array(evaluation[1], dtype=np.uint64), ch)
    assert dense.c1() == sparse.c1() == ref["c_1"]
    for j in range(2 * k_next):
        rp = ch[j - 1] if j else F.one
        e = [int(x) for x in ref["evals"][j]]
        assert dense.round_evals(rp, j) == e and sparse.round_evals(rp, j) == e
    # errors
    bad = make_circuit(pkg, [[("add", 0, 9)] * 4], 4)           # input index out of range
    with pytest.raises(pkg.SumcheckHipError) as ei:
        gp.wiring(ctx, bad, 0, [F.one, F.one])
    assert ei.value.code == 1
    t4 = pkg.DenseMultilinearExtension.generate(ctx, 1, 4)
    t3 = pkg.DenseMultilinearExtension.generate(ctx, 2, 3)
    with pytest.raises(pkg.SumcheckHipError) as ei:
        gp.W(t4, t4, t3, t3).round_evals()                       # 4 != 3 + 3 variables
    assert ei.value.code == 1
    with pytest.raises(pkg.SumcheckHipError) as ei:
        pkg.triangle_counting.G(t4, t3, t4, 2).round_evals()     # inconsistent copies
    assert ei.value.code == 1
    eng = gp.W(t4, t4, pkg.DenseMultilinearExtension.generate(ctx, 3, 2), pkg.DenseMultilinearExtension.generate(ctx, 4, 2)).native_prover()
    with pytest.raises(pkg.SumcheckHipError) as ei:
        eng.round_evals(F.one, 2)                                # out of order
    assert ei.value.code == 5


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("transport", ["host", "peer"])
def test_w_prover_sharded(world, transport):
    """the dense W prover on a sharded context (BASELINE config 5 names gkr on 8 GPUs): every rank holds its rows of c
    of add_i / mul_i (top index bits = rank) and the whole W tables; P and L are summed across ranks, add(r_b,.) /
    mul(r_b,.) gathered; round polynomials bit-exact vs the oracle on every rank.  The sparse prover is replicated."""
    import threading
    from test_gpu_sharded import Loopback
    if transport == "peer" and world > 2:
        pytest.skip("threads of one process: see test_virtual_ranks_peer_transport")
    pkg = load_package()
    p = GOLD
    o = oracle(p)
    F0 = pkg.Field(p)
    gp = pkg.gkr_protocol
    rng = random.Random(77)
    for ks in ([3, 3], [5, 4], [6, 7], [4, 9]):
        layers = random_circuit(rng, ks)
        circuit = make_circuit(pkg, layers, 1 << ks[-1])
        inputs = [F0.from_int(rng.randrange(p)) for _ in range(1 << ks[-1])]
        evaluation = circuit.evaluate(F0, inputs)
        k_i, k_next = ks[0], ks[1]
        r_i = [F0.from_int(rng.randrange(p)) for _ in range(k_i)]
        oadd, omul = o.wiring_fixed(layers[0], k_next, r_i)
        ow = np.array(evaluation[1], dtype=np.uint64)
        ch = [F0.from_int(rng.randrange(p)) for _ in range(2 * k_next)]
        ref = o.w_prove(oadd, omul, ow, ow, ch)
        assert ref["status"] == 0
        lb = Loopback(world)
        ctxs, errors, results = [None] * world, [], [None] * world

        def body(rank):
            try:
                ctx = pkg.Context(pkg.Field(p))
                if transport == "peer":
                    ctx.set_option("peer_spin_ms", 20000)
                    ctx.comm_peer_export(rank, world)
                    ctxs[rank] = ctx
                    lb.barrier.wait()
                    ctx.comm_peer_connect_local(ctxs)
                    lb.barrier.wait()
                else:
                    ar, ag = lb.collectives(rank)
                    ctx.comm_init_host(rank, world, ar, ag)
                n_loc = oadd.size // world
                DM = pkg.DenseMultilinearExtension
                add_t = DM.from_evaluations_vec(ctx, 2 * k_next - (world.bit_length() - 1), oadd[rank * n_loc:(rank + 1) * n_loc])
                mul_t = DM.from_evaluations_vec(ctx, 2 * k_next - (world.bit_length() - 1), omul[rank * n_loc:(rank + 1) * n_loc])
                w_t = DM.from_evaluations_vec(ctx, k_next, ow)
                eng = gp.W(add_t, mul_t, w_t, w_t).native_prover()
                seng = gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i)
                got = [(eng.c1(), seng.c1())]
                for j in range(2 * k_next):
                    rp = ch[j - 1] if j else F0.one
                    got.append((eng.round_evals(rp, j), seng.round_evals(rp, j)))
                results[rank] = got
                if transport == "peer":
                    lb.barrier.wait()
                del eng, seng, add_t, mul_t, w_t
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
        for got in results:
            assert got[0] == (ref["c_1"], ref["c_1"]), ks
            for j in range(2 * k_next):
                e = [int(x) for x in ref["evals"][j]]
                assert got[1 + j] == (e, e), (ks, j)


@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_w_prover_general_shapes(p):
    """the two-phase W prover on DENSE random add/mul tables (not wiring-shaped) with any split of the variables
    between b and c - including none on one side, and a W that already had variables fixed - against the
    independent big-integer restatement of W (oracle/pyref.py w_transcript)"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    gp = pkg.gkr_protocol
    DM = pkg.DenseMultilinearExtension
    rng = random.Random(p % 1013)
    for kb, kc in [(1, 1), (3, 5), (5, 2), (0, 4), (4, 0), (1, 6), (6, 1), (2, 2)]:
        n = kb + kc
        add = [rng.randrange(p) for _ in range(1 << n)]
        mul = [rng.randrange(p) for _ in range(1 << n)]
        wb = [rng.randrange(p) for _ in range(1 << kb)]
        wc = [rng.randrange(p) for _ in range(1 << kc)]
        ch = [rng.randrange(p) for _ in range(n)]
        ref = pyref.w_transcript(add, mul, wb, wc, ch, p)
        w = gp.W(DM.from_evaluations_vec(ctx, n, F.from_ints(add)), DM.from_evaluations_vec(ctx, n, F.from_ints(mul)),
                 DM.from_evaluations_vec(ctx, kb, F.from_ints(wb)), DM.from_evaluations_vec(ctx, kc, F.from_ints(wc)))
        eng = w.native_prover()
        assert F.to_int(eng.c1()) == ref["c_1"], (kb, kc)
        for j in range(n):
            e = eng.round_evals(F.from_int(ch[j - 1]) if j else F.one, j)
            assert [F.to_int(x) for x in e] == ref["evals"][j], (kb, kc, j)
        assert F.to_int(w.evaluate(F.from_ints(ch).tolist())) == ref["final_eval"]
        # a W with its first t variables already fixed (t inside b, at the boundary, inside c): the engine
        # starts from that state and its rounds are the tail of the full transcript
        for t in sorted({1, kb, min(kb + 1, n - 1)} - {0, n}):
            if t >= n:
                continue
            w2 = w.fix_variables(F.from_ints(ch[:t]).tolist())
            eng2 = w2.native_prover()
            for j in range(n - t):
                e = eng2.round_evals(F.from_int(ch[t + j - 1]) if j else F.one, j)
                assert [F.to_int(x) for x in e] == ref["evals"][t + j], (kb, kc, t, j)
