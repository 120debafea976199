"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against
  - the committed golden fixtures (tests/golden/*.json),
  - the CPU oracle on the same seeded inputs,
  - the reference's own known-answer tests replayed through the Python mirror of its API.
Everything here is exact integer arithmetic: the bar is bit-for-bit equality."""
import random

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, TOY_MODULI, challenges, load_golden, oracle, pid, pyref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return load_package()


_ctxs = {}


@pytest.fixture(scope="module", autouse=True)
def _close_cached_contexts():
    """every cached context holds a stream (a hardware queue) and pinned buffers: give them back when the module is done"""
    yield
    for c in _ctxs.values():
        try:
            c.close()
        except Exception:
            pass
    _ctxs.clear()


def schedule_options(vpp):
    """test shorthand for the prover schedule: 1 = one round per pass; 2 = two rounds per pass;
    3 = two rounds per pass and three from the first pass (what the library does by default on a
    large instance, forced here at every size).  1 and 2 also switch the five-round passes of the
    small tables off: the literal schedules, which the default one must reproduce bit for bit"""
    return {"vars_per_pass": min(vpp, 2), "first_pass_vars": 3 if vpp == 3 else min(vpp, 2),
            "grid_pass": 1 if vpp == 3 else 0}


def ctx_for(pkg, p, **opts):
    key = (p, tuple(sorted(opts.items())))
    if key not in _ctxs:
        c = pkg.Context(pkg.Field(p))
        if "vars_per_pass" in opts:
            opts = dict(opts, **schedule_options(opts["vars_per_pass"]))
        for k, v in opts.items():
            c.set_option(k, v)
        _ctxs[key] = c
    return _ctxs[key]


def run_python_protocol(pkg, ctx, g, ch):
    """the loop of matrix-multiplication/src/lib.rs:354-370 with scripted verifier draws"""
    scp = pkg.sum_check_protocol

    class Scripted(scp.RngF):
        def __init__(self, vals):
            self.vals = list(vals)

        def draw(self):
            return int(self.vals.pop(0))

    prover = scp.Prover.new(g.clone())
    c_1 = prover.c_1()
    num_vars = g.num_vars()
    verifier = scp.Verifier.new(num_vars, g)
    verifier.set_c_1(c_1)
    rng = Scripted(ch)
    r_j = ctx.field.one
    polys, final = [], None
    for j in range(num_vars):
        g_j = prover.round(r_j, j)
        polys.append(g_j)
        res = verifier.round(g_j, rng)
        if res.is_final():
            final = res.value
        else:
            r_j = res.value
    return c_1, polys, final


# ---- golden transcripts -------------------------------------------------------------------

@pytest.mark.parametrize("vpp", [1, 2, 3])
@pytest.mark.parametrize("entry", load_golden("transcripts.json"), ids=lambda e: "%s-n%d" % (pid(e["p"]), e["n"]))
def test_golden_transcripts(pkg, entry, vpp):
    p, n = entry["p"], entry["n"]
    ctx = ctx_for(pkg, p, vars_per_pass=vpp)
    F = ctx.field
    a = pkg.DenseMultilinearExtension.generate(ctx, entry["seed_a"], n)
    b = pkg.DenseMultilinearExtension.generate(ctx, entry["seed_b"], n)
    if "a" in entry:
        assert F.to_ints(a.to_evaluations()) == entry["a"]
        assert F.to_ints(b.to_evaluations()) == entry["b"]
    g = pkg.matrix_multiplication.G(a, b)
    ch = [F.from_int(x) for x in entry["challenges"]]
    c_1, polys, final = run_python_protocol(pkg, ctx, g, ch)
    assert F.to_int(c_1) == entry["c_1"]
    for j, poly in enumerate(polys):
        dense = [0, 0, 0]
        for d, c in poly.coeffs:
            dense[d] = F.to_int(c)
        assert dense == entry["coeffs"][j], "round %d coefficients" % j
    if n >= 2:
        assert final is True
    assert F.to_int(g.evaluate(ch)) == entry["final_eval"]
    # native whole-run entry point gives the same sums
    c1n, evals, chn = pkg.matrix_multiplication.prove(ctx, g, entry["seed_r"])
    assert F.to_int(c1n) == entry["c_1"]
    assert [F.to_ints(row) for row in evals] == entry["evals"]
    assert F.to_ints(chn) == entry["challenges"]


# ---- GPU vs C oracle on seeded inputs -----------------------------------------------------

@pytest.mark.parametrize("vpp", [1, 2, 3])
@pytest.mark.parametrize("p", [GOLD] + TOY_MODULI + [2**64 - 59], ids=pid)
def test_prover_vs_oracle_sizes(pkg, p, vpp):
    ctx = ctx_for(pkg, p, vars_per_pass=vpp)
    o = oracle(p)
    sizes = range(1, 21) if p == GOLD else (1, 2, 3, 4, 7, 10, 13, 16)
    for n in sizes:
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A + n, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B + n, n)
        oa, ob = o.generate(pyref.SEED_A + n, n), o.generate(pyref.SEED_B + n, n)
        assert np.array_equal(a.to_evaluations(), oa)
        g = pkg.matrix_multiplication.G(a, b)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        assert np.array_equal(ch, challenges(o, n))
        ref = o.prove(oa, ob, ch)
        assert ref["status"] == 0
        assert c1 == ref["c_1"], "n=%d c_1" % n
        assert np.array_equal(evals, ref["evals"]), "n=%d round sums" % n
        assert g.evaluate([int(x) for x in ch]) == ref["final_eval"], "n=%d final evaluation" % n


@pytest.mark.parametrize("first,tail", [(2, 1), (3, 0), (1, 1), (0, 1), (0, 0)])
def test_prover_mixed_pass_widths(pkg, first, tail):
    """first-pass width and the five-round passes of the small tables (tail = grid_pass) are independent
    options: every combination gives the reference's transcript (first = 0 is the size rule)"""
    ctx = ctx_for(pkg, GOLD, first_pass_vars=first, grid_pass=tail)
    o = oracle(GOLD)
    for n in (3, 4, 5, 6, 8, 9, 10, 11, 14, 17, 21):
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A + n, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B + n, n)
        g = pkg.matrix_multiplication.G(a, b)
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        ref = o.prove(o.generate(pyref.SEED_A + n, n), o.generate(pyref.SEED_B + n, n), ch)
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]), (first, tail, n)
        assert g.evaluate([int(x) for x in ch]) == ref["final_eval"]


def test_random_tables_uploaded(pkg):
    """arbitrary (not generator-made) inputs incl. 0 and p-1, through upload"""
    rng = random.Random(99)
    for p in (GOLD, 5, 389):
        ctx = ctx_for(pkg, p)
        o = oracle(p)
        for n in (1, 2, 5, 9, 12):
            vals = [rng.choice([0, 1, p - 1, rng.randrange(p)]) for _ in range(2 << n)]
            oa, ob = o.to_mont(vals[: 1 << n]), o.to_mont(vals[1 << n:])
            a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, oa)
            b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, ob)
            g = pkg.matrix_multiplication.G(a, b)
            ch = o.to_mont([rng.randrange(p) for _ in range(n)])
            it = iter(ch)
            import ctypes
            c1, evals, chn = pkg.matrix_multiplication.prove(
                ctx, g, 0, draw=lambda _u, _j, _e: int(next(it)))
            ref = o.prove(oa, ob, ch)
            assert ref["status"] == 0 and c1 == ref["c_1"]
            assert np.array_equal(evals, ref["evals"])
            assert np.array_equal(chn, ch)


@pytest.mark.parametrize("vpp", [2, 3])
def test_extreme_words(pkg, vpp):
    """tables made only of the largest and smallest Montgomery words (p-1, p-2, 0, 1) and
    challenges p-1: every 128-bit product is as large as it gets, which is what the lazy signed
    accumulators and the carry chains of the hand-scheduled arithmetic have to survive"""
    p = GOLD
    ctx = ctx_for(pkg, p, vars_per_pass=vpp)
    o = oracle(p)
    for n, pattern in [(16, [p - 1]), (18, [p - 1, p - 2]), (17, [p - 1, 0, 1, p - 1, p - 2, 0, 0xFFFFFFFF, 1 << 32]),
                       (20, [p - 1, p - 1, p - 1, 0])]:
        words = np.array([pattern[i % len(pattern)] for i in range(1 << n)], dtype=np.uint64)   # raw words
        other = words[::-1].copy()
        a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, words)
        b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, other)
        g = pkg.matrix_multiplication.G(a, b)
        ch = np.array([p - 1 - (j % 3) for j in range(n)], dtype=np.uint64)
        it = iter(ch)
        c1, evals, chn = pkg.matrix_multiplication.prove(ctx, g, 0, draw=lambda _u, _j, _e: int(next(it)))
        ref = o.prove(words, other, ch)
        assert ref["status"] == 0 and c1 == ref["c_1"], n
        assert np.array_equal(evals, ref["evals"]), n
        assert g.evaluate([int(x) for x in ch]) == ref["final_eval"], n


# ---- trait methods one by one ------------------------------

@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_trait_methods_vs_oracle(pkg, p):
    ctx = ctx_for(pkg, p)
    o = oracle(p)
    F = ctx.field
    for n in (1, 2, 3, 6, 11, 15):
        oa, ob = o.generate(1, n), o.generate(2, n)
        a = pkg.DenseMultilinearExtension.generate(ctx, 1, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, 2, n)
        g = pkg.matrix_multiplication.G(a, b)
        assert g.num_vars() == n
        assert np.array_equal(g.to_evaluations(), o.to_evaluations(oa, ob))            # a8
        assert g.hypercube_sum() == o.c1(oa, ob)                                       # a2
        e = o.round_evals(oa, ob)
        assert g.round_evals() == [int(x) for x in e]                                  # a6
        poly = g.to_univariate()
        co = o.interpolate(e)
        dense = [0, 0, 0]
        for d, c in poly.coeffs:
            dense[d] = c
        assert dense == [int(x) for x in co]                                           # a7
        # the reference-shaped interpolation (three divisions) agrees with the C-ABI one
        pts = [(F.zero, int(e[0])), (F.one, int(e[1])), (F.two, int(e[2]))]
        assert pkg.matrix_multiplication.interpolate_quadratic_poly(F, pts) == poly
        r = o.challenge(7, n)
        g2 = g.fix_variables([r])                                                       # a5
        assert np.array_equal(g2.f_a.to_evaluations(), o.fix_variables(oa, [r]))
        assert np.array_equal(g2.f_b.to_evaluations(), o.fix_variables(ob, [r]))
        if n >= 2:
            g3, poly3 = g.fold_and_univariate(r)                                       # fused a5+a6
            assert np.array_equal(g3.f_a.to_evaluations(), g2.f_a.to_evaluations())
            assert poly3 == g2.to_univariate()
        pt = [o.challenge(8, j) for j in range(n)]
        assert g.evaluate(pt) == o.g_evaluate(oa, ob, pt)                               # a9
        assert g.evaluate(pt[:-1]) is None
        # inputs were never modified
        assert np.array_equal(a.to_evaluations(), oa)


# ---- dense MLE: fix_variables / evaluate, LE and BE (BASELINE config 2 shapes) --------------

@pytest.mark.parametrize("entry", load_golden("mle_vectors.json"), ids=lambda e: "%s-n%d" % (pid(e["p"]), e["n"]))
def test_golden_mle_vectors(pkg, entry):
    p, n = entry["p"], entry["n"]
    ctx = ctx_for(pkg, p)
    F = ctx.field
    t = pkg.DenseMultilinearExtension.generate(ctx, entry["seed"], n)
    pt = [F.from_int(x) for x in entry["point"]]
    assert F.to_int(t.evaluate(pt)) == entry["evaluate_le"]
    assert F.to_int(t.evaluate(pt, order=pkg.ORDER_BE)) == entry["evaluate_be"]
    me = pkg.multilinear_extensions
    assert F.to_int(me.vsbw_multilinear_from_evaluations(ctx, t, pt)) == entry["evaluate_be"]
    assert F.to_int(me.cti_multilinear_from_evaluations(ctx, t.to_evaluations(), pt)) == entry["evaluate_be"]
    k = entry["k"]
    if entry["fix_le"] is not None:
        assert F.to_ints(t.fix_variables(pt[:k]).to_evaluations()) == entry["fix_le"]
        assert F.to_ints(t.fix_variables(pt[:k], order=pkg.ORDER_BE).to_evaluations()) == entry["fix_be"]


@pytest.mark.parametrize("p", [GOLD, 5], ids=pid)
def test_fix_variables_all_k(pkg, p):
    ctx = ctx_for(pkg, p)
    o = oracle(p)
    for n in (1, 2, 3, 4, 5, 9, 14):
        ot = o.generate(3, n)
        t = pkg.DenseMultilinearExtension.generate(ctx, 3, n)
        pt = [o.challenge(4, j) for j in range(n)]
        for k in range(0, n + 1):
            for order in (pkg.ORDER_LE, pkg.ORDER_BE):
                got = t.fix_variables(pt[:k], order=order).to_evaluations()
                assert np.array_equal(got, o.fix_variables(ot, pt[:k], order)), (n, k, order)
        assert t.evaluate(pt) == o.evaluate(ot, pt)
        assert t.evaluate(pt, order=pkg.ORDER_BE) == o.vsbw(ot, pt)
        assert np.array_equal(t.to_evaluations(), ot)
    # many variables at once: the one-pass segment dot (8..17 variables per pass) and its chaining
    n = 20
    ot = o.generate(5, n)
    t = pkg.DenseMultilinearExtension.generate(ctx, 5, n)
    pt = [o.challenge(6, j) for j in range(n)]
    for k in (8, 11, 16, 17, 18, 20):
        got = t.fix_variables(pt[:k]).to_evaluations()
        assert np.array_equal(got, o.fix_variables(ot, pt[:k], pkg.ORDER_LE)), (n, k)


@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
@pytest.mark.parametrize("max_blocks", [1, 3, 8])
def test_block_per_cu_streamers_vs_oracle(pkg, p, max_blocks):
    """large tables launch evaluate / fix_low / fold with ONE block per CU whose waves draw their work from an LDS counter
    (kernels.hpp: evaluate_kernel, fix_low_kernel, fold_kernel); the launch shape follows `max_blocks`, so a small cap
    takes a 2^18..2^20-entry table - sizes the oracle does in a second - down the same path, with odd grids and with
    fewer units of work than waves"""
    ctx = ctx_for(pkg, p, max_blocks=max_blocks)
    o = oracle(p)
    for n in (18, 20):
        ot = o.generate(7, n)
        t = pkg.DenseMultilinearExtension.generate(ctx, 7, n)
        pt = [o.challenge(8, j) for j in range(n)]
        assert t.evaluate(pt) == o.evaluate(ot, pt), (n, "evaluate")
        assert t.evaluate(pt, order=pkg.ORDER_BE) == o.vsbw(ot, pt), (n, "evaluate BE")
        for k in (1, 2, 3, 8, 12, 17):
            got = t.fix_variables(pt[:k]).to_evaluations()
            assert np.array_equal(got, o.fix_variables(ot, pt[:k], pkg.ORDER_LE)), (n, k)


def test_relabel_and_clone(pkg):
    ctx = ctx_for(pkg, 389)
    o = oracle(389)
    for nv, a, b, k in [(4, 0, 2, 2), (6, 0, 3, 3), (6, 1, 4, 2), (8, 0, 4, 4), (5, 0, 3, 1)]:
        ot = o.generate(5, nv)
        t = pkg.DenseMultilinearExtension.generate(ctx, 5, nv)
        assert np.array_equal(t.relabel(a, b, k).to_evaluations(), o.relabel(ot, a, b, k))
        assert np.array_equal(t.clone().to_evaluations(), ot)


# ---- the reference's own tests, replayed through the mirrored API ---------------------------

KATS = load_golden("reference_kats.json")


def test_ref_example_from_book_mle(pkg):
    """multilinear-extensions/src/lib.rs:76-120"""
    k = KATS["mle_f5_grid"]
    ctx = ctx_for(pkg, 5)
    F = ctx.field
    me = pkg.multilinear_extensions
    evals = F.from_ints(k["evals"])
    for i in range(5):
        line = [F.to_int(me.cti_multilinear_from_evaluations(ctx, evals, [F.from_int(i), F.from_int(j)]))
                for j in range(5)]
        assert line == k["expected_grid"][i], "at line %d" % i
        line = [F.to_int(me.vsbw_multilinear_from_evaluations(ctx, evals, [F.from_int(i), F.from_int(j)]))
                for j in range(5)]
        assert line == k["expected_grid"][i]


def _bits(field, v, nbits):
    return [field.one if (v >> i) & 1 else field.zero for i in range(nbits)]


def test_ref_matmul_example_from_book(pkg):
    """matrix-multiplication/src/lib.rs:245-303 (+ :340's identity)"""
    k = KATS["matmul_book"]
    ctx = ctx_for(pkg, 5)
    F = ctx.field
    scp, mm = pkg.sum_check_protocol, pkg.matrix_multiplication
    a, b = F.from_ints(sum(k["A"], [])), F.from_ints(sum(k["B"], []))
    rng = scp.FieldRng(F, random.Random(3))
    for i in range(2):
        for j in range(2):
            point = _bits(F, i, 1) + _bits(F, j, 1)
            g = mm.G.new(ctx, 1, a, b, point)
            prover = scp.Prover.new(g.clone())
            c_1 = prover.c_1()
            assert F.to_int(c_1) == k["C"][i][j]
            num_vars = g.num_vars()
            r_j = F.one
            verifier = scp.Verifier.new(num_vars, g)
            verifier.set_c_1(c_1)
            for kk in range(num_vars):
                g_j = prover.round(r_j, kk)
                res = verifier.round(g_j, rng)
                if res.is_final():
                    assert res.value
                else:
                    r_j = res.value


def test_ref_matmul_randomized(pkg):
    """matrix-multiplication/src/lib.rs:315-374 on the committed matrices"""
    k = KATS["matmul_randomized_f5"]
    ctx = ctx_for(pkg, 5)
    F = ctx.field
    scp, mm = pkg.sum_check_protocol, pkg.matrix_multiplication
    rng = scp.FieldRng(F, random.Random(11))
    for case in k["cases"]:
        p_ = case["logn"]
        n = 1 << p_
        a, b = F.from_ints(sum(case["A"], [])), F.from_ints(sum(case["B"], []))
        for i in range(n):
            for j in range(n):
                point = _bits(F, i, p_) + _bits(F, j, p_)
                g = mm.G.new(ctx, p_, a, b, point)
                prover = scp.Prover.new(g.clone())
                c_1 = prover.c_1()
                assert F.to_int(c_1) == case["C"][i][j]                                # :340
                resu = F.zero
                for z in range(n):
                    resu = F.add(resu, g.evaluate(_bits(F, z, p_)))                    # :346-350
                assert c_1 == resu                                                     # :352
                num_vars = g.num_vars()
                r_j = F.one
                verifier = scp.Verifier.new(num_vars, g)
                verifier.set_c_1(c_1)
                seen_final = False
                for jj in range(num_vars):
                    g_j = prover.round(r_j, jj)
                    res = verifier.round(g_j, rng)
                    if res.is_final():
                        assert res.value
                        seen_final = True
                    else:
                        r_j = res.value
                assert seen_final


def test_ref_restrict_poly_le_order(pkg):
    """gkr-protocol/src/lib.rs:507-548: the [32,385,383] polynomial is the LE evaluate of
    the table along the line b + t(c-b)"""
    k = KATS["restrict_poly_389"]
    p = k["p"]
    ctx = ctx_for(pkg, p)
    F = ctx.field
    t = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2, F.from_ints(k["evals"]))
    for t0 in range(9):
        pt = [F.from_int(bi + t0 * (ci - bi)) for bi, ci in zip(k["b"], k["c"])]
        assert F.to_int(t.evaluate(pt)) == pyref.poly_eval(k["expected_coeffs"], t0, p)


def test_verifier_rejects_wrong_claim(pkg):
    """sum-check-protocol/src/lib.rs:286-291 and :318-323"""
    ctx = ctx_for(pkg, 389)
    F = ctx.field
    scp, mm = pkg.sum_check_protocol, pkg.matrix_multiplication
    a = pkg.DenseMultilinearExtension.generate(ctx, 1, 4)
    b = pkg.DenseMultilinearExtension.generate(ctx, 2, 4)
    g = mm.G(a, b)
    prover = scp.Prover.new(g.clone())
    verifier = scp.Verifier.new(4, g)
    verifier.set_c_1(F.add(prover.c_1(), F.one))
    rng = scp.FieldRng(F, random.Random(5))
    with pytest.raises(scp.ProverClaimMismatch):
        verifier.round(prover.round(F.one, 0), rng)
    v2 = scp.Verifier.new(4, None, F)
    v2.set_c_1(prover.c_1())
    p2 = scp.Prover.new(g.clone())
    r = F.one
    with pytest.raises(scp.NoPolySet):
        for j in range(4):
            res = v2.round(p2.round(r, j), rng)
            r = res.value


# ---- error behaviour of the C ABI ------------------------------------------------------------

def test_abi_errors(pkg):
    import ctypes
    ctx = ctx_for(pkg, GOLD)
    lib = ctx.lib
    h = ctypes.c_void_p()
    host = np.zeros(3, dtype=np.uint64)
    rc = lib.sc_table_upload(ctx.h, host.ctypes.data_as(pkg._lib.u64p), 3, ctypes.byref(h))
    assert rc == 1 and b"power of two" in lib.sc_last_error(ctx.h)
    a = pkg.DenseMultilinearExtension.generate(ctx, 1, 5)
    b = pkg.DenseMultilinearExtension.generate(ctx, 2, 4)
    with pytest.raises(pkg.SumcheckHipError) as ei:
        pkg.matrix_multiplication.G(a, b).round_evals()
    assert ei.value.code == 1
    b = pkg.DenseMultilinearExtension.generate(ctx, 2, 5)
    eng = pkg.matrix_multiplication.G(a, b).native_prover()
    with pytest.raises(pkg.SumcheckHipError) as ei:
        eng.round_evals(ctx.field.one, 1)          # round 1 before round 0
    assert ei.value.code == 5
    eng.round_evals(ctx.field.one, 0)
    with pytest.raises(pkg.SumcheckHipError) as ei:
        eng.round_evals(GOLD, 1)                   # unreduced challenge
    assert ei.value.code == 1
    with pytest.raises(pkg.SumcheckHipError):
        a.fix_variables([1] * 6)
    with pytest.raises(pkg.SumcheckHipError):
        a.evaluate([1] * 4)
    bad = pkg._lib.ScField(GOLD, 1, 2, 3)
    rc = lib.sc_ctx_create(ctypes.byref(bad), 0, ctypes.byref(h))
    assert rc == 1
    with pytest.raises(pkg.SumcheckHipError):
        ctx.set_option("vars_per_pass", 3)


# ---- BASELINE.json sizes through size-independent properties ---------------------------------

def _check_round_identities(F, c1, evals, ch, final_eval):
    """g_1(0)+g_1(1) = c_1; g_j(0)+g_j(1) = g_{j-1}(r_{j-1}); g_n(r_n) = g(r)
    (sum-check-protocol/src/lib.rs:286, :316-318, :303)"""
    def at(e, r):
        # Lagrange on {0,1,2}
        inv2 = F.inv(F.two)
        l0 = F.mul(F.mul(F.sub(r, F.one), F.sub(r, F.two)), inv2)
        l1 = F.neg(F.mul(r, F.sub(r, F.two)))
        l2 = F.mul(F.mul(r, F.sub(r, F.one)), inv2)
        return F.add(F.add(F.mul(l0, int(e[0])), F.mul(l1, int(e[1]))), F.mul(l2, int(e[2])))
    claim = c1
    for j in range(len(evals)):
        assert F.add(int(evals[j][0]), int(evals[j][1])) == claim, "round %d" % j
        claim = at(evals[j], int(ch[j]))
    assert claim == final_eval


@pytest.mark.parametrize("n,vpp", [(24, 2), (24, 3), (26, 2), (26, 1), (28, 2), (28, 3), (28, 1), (30, 3)])
def test_full_size_identities(pkg, n, vpp):
    ctx = ctx_for(pkg, GOLD, vars_per_pass=vpp)
    F = ctx.field
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    final = g.evaluate([int(x) for x in ch])
    _check_round_identities(F, c1, evals, ch, final)
    # the two schedules agree bit for bit
    other = ctx_for(pkg, GOLD, vars_per_pass=1 if vpp > 1 else 3)
    if n <= 26:
        a2 = pkg.DenseMultilinearExtension.generate(other, pyref.SEED_A, n)
        b2 = pkg.DenseMultilinearExtension.generate(other, pyref.SEED_B, n)
        c1b, evals_b, _ = pkg.matrix_multiplication.prove(other, pkg.matrix_multiplication.G(a2, b2), pyref.SEED_R)
        assert c1b == c1 and np.array_equal(evals_b, evals)
        del a2, b2
    del a, b, g


def test_full_size_vs_oracle_n24(pkg):
    """config 2/3 size the oracle still finishes in seconds"""
    n = 24
    ctx = ctx_for(pkg, GOLD)
    o = oracle(GOLD)
    oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    ref = o.prove(oa, ob, ch)
    assert ref["status"] == 0 and c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
    pt = [o.challenge(pyref.SEED_PT, j) for j in range(n)]
    assert a.evaluate(pt) == o.evaluate(oa, pt)
    assert a.evaluate(pt, order=pkg.ORDER_BE) == o.vsbw(oa, pt)
    assert np.array_equal(a.fix_variables(pt[:1]).to_evaluations(), o.fix_variables(oa, pt[:1]))


def test_stress_inkernel_handoff(pkg):
    """the ticket hand-off of finish_pass under back-to-back launches whose partials differ:
    a stale partial (or mailbox word) from the previous launch would change a round sum"""
    ctx = ctx_for(pkg, GOLD)
    o = oracle(GOLD)
    mm = pkg.matrix_multiplication
    for n in (13, 16, 19):
        gs, refs = [], []
        for k in range(3):
            a = pkg.DenseMultilinearExtension.generate(ctx, 100 + 2 * k, n)
            b = pkg.DenseMultilinearExtension.generate(ctx, 101 + 2 * k, n)
            gs.append(mm.G(a, b))
            ch = challenges(o, n)
            refs.append(o.prove(o.generate(100 + 2 * k, n), o.generate(101 + 2 * k, n), ch))
        for it in range(150):
            k = it % 3
            c1, evals, _ = mm.prove(ctx, gs[k], pyref.SEED_R)
            assert c1 == refs[k]["c_1"], (n, it)
            assert np.array_equal(evals, refs[k]["evals"]), (n, it)


def test_cpp_host_mirror_runs_reference_tests():
    """thaler-study_amd/host/*.hpp (C++ mirror of the Rust API) replaying the reference's own
    unit tests against the C ABI: tests/cpp/test_reference_tests.cpp"""
    import subprocess
    import __graft_entry__ as ge
    exe = ge.build_cpp_host_tests()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout


def test_zero_variable_polynomial(pkg):
    """tables of a single entry: no variable, c_1 is the product, no rounds
    (Prover::new on a 0-variable G, sum-check-protocol/src/lib.rs:88-97)"""
    for p in (GOLD, 5):
        ctx = ctx_for(pkg, p)
        F = ctx.field
        a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 0, F.from_ints([3]))
        b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 0, F.from_ints([4]))
        g = pkg.matrix_multiplication.G(a, b)
        assert g.num_vars() == 0
        assert F.to_int(g.hypercube_sum()) == 12 % p
        assert F.to_ints(g.to_evaluations()) == [12 % p]
        assert F.to_int(g.evaluate([])) == 12 % p
        prover = pkg.sum_check_protocol.Prover.new(g.clone())
        assert F.to_int(prover.c_1()) == 12 % p and prover.num_vars() == 0
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        assert F.to_int(c1) == 12 % p and len(evals) == 0 and len(ch) == 0
        assert F.to_ints(a.fix_variables([]).to_evaluations()) == [3]
