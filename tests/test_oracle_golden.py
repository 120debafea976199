"""Pins the CPU oracle (oracle/sc_oracle.c) against
  (1) every known-answer vector the reference's own tests hold for this path
      (tests/golden/reference_kats.json, transcribed with file:line sources),
  (2) the independent big-integer restatement oracle/pyref.py,
  (3) the committed transcripts of the synthetic instance.
No GPU needed."""
import json
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref  # noqa: E402
from oracle import Oracle  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLD = pyref.GOLDILOCKS


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


KATS = load("reference_kats.json")


def test_kat_mle_f5_grid():
    """multilinear-extensions/src/lib.rs:76-120"""
    k = KATS["mle_f5_grid"]
    o = Oracle(k["p"])
    ev = o.to_mont(k["evals"])
    for i in range(5):
        for j in range(5):
            r = o.to_mont([i, j])
            assert o.from_mont1(o.cti(ev, r)) == k["expected_grid"][i][j]
            assert o.from_mont1(o.vsbw(ev, r)) == k["expected_grid"][i][j]
            # BE evaluate == LE evaluate at the reversed point
            assert o.from_mont1(o.evaluate(ev, r[::-1].copy())) == k["expected_grid"][i][j]


def test_kat_matmul_book():
    """matrix-multiplication/src/lib.rs:203-303"""
    k = KATS["matmul_book"]
    o = Oracle(k["p"])
    A = o.to_mont(sum(k["A"], []))
    B = o.to_mont(sum(k["B"], []))
    for i in range(2):
        for j in range(2):
            pt = o.to_mont(pyref.bits_le(i, 1) + pyref.bits_le(j, 1))
            fa, fb = o.g_new(1, A, B, pt)
            assert o.from_mont1(o.c1(fa, fb)) == k["C"][i][j]
            ch = o.to_mont([3])
            res = o.prove(fa, fb, ch)
            assert res["status"] == 0 and o.from_mont1(res["c_1"]) == k["C"][i][j]


def test_kat_matmul_randomized_identities():
    """matrix-multiplication/src/lib.rs:316-374: c_1 == (A*B)[i][j], sum_z g(z) == c_1,
    and the verifier accepts every round."""
    k = KATS["matmul_randomized_f5"]
    p = k["p"]
    o = Oracle(p)
    rng = random.Random(1)
    for case in k["cases"]:
        logn = case["logn"]
        n = 1 << logn
        A = o.to_mont(sum(case["A"], []))
        B = o.to_mont(sum(case["B"], []))
        for i in range(n):
            for j in range(n):
                pt = o.to_mont(pyref.bits_le(i, logn) + pyref.bits_le(j, logn))
                fa, fb = o.g_new(logn, A, B, pt)
                c1 = o.from_mont1(o.c1(fa, fb))
                assert c1 == case["C"][i][j]
                tot = sum(o.from_mont1(o.g_evaluate(fa, fb, o.to_mont(pyref.bits_le(z, logn))))
                          for z in range(n)) % p
                assert tot == c1
                ch = o.to_mont([rng.randrange(p) for _ in range(logn)])
                assert o.prove(fa, fb, ch)["status"] == 0


def test_kat_restrict_poly_pins_le_order():
    """gkr-protocol/src/lib.rs:507-548: [32,385,383] over F_389.  restrict_poly(t) at a
    point t0 equals the LE evaluate at b + t0*(c-b); checked through the C oracle."""
    k = KATS["restrict_poly_389"]
    p = k["p"]
    o = Oracle(p)
    assert pyref.restrict_poly(k["b"], k["c"], k["evals"], p) == k["expected_coeffs"]
    ev = o.to_mont(k["evals"])
    for t0 in range(7):
        pt = [(bi + t0 * (ci - bi)) % p for bi, ci in zip(k["b"], k["c"])]
        expect = pyref.poly_eval(k["expected_coeffs"], t0, p)
        assert o.from_mont1(o.evaluate(ev, o.to_mont(pt))) == expect
    # the BE reading of the same table gives a different polynomial: the KAT does pin LE
    assert any(
        pyref.vsbw(k["evals"], [(bi + t0 * (ci - bi)) % p for bi, ci in zip(k["b"], k["c"])], p)
        != pyref.poly_eval(k["expected_coeffs"], t0, p) for t0 in range(7))


def test_kat_triangle_index_convention():
    """triangle-counting/src/lib.rs:232-266, :296-300"""
    k = KATS["triangle_simple_389"]
    assert pyref.triangle_c1(sum(k["adjacency"], []), k["k"], k["p"]) == k["expected_c_1"]


@pytest.mark.parametrize("entry", load("transcripts.json"),
                         ids=lambda e: "p%d-n%d" % (e["p"] if e["p"] < 2**40 else 0, e["n"]))
def test_transcripts_c_oracle_vs_fixture(entry):
    p, n = entry["p"], entry["n"]
    o = Oracle(p)
    a = o.generate(entry["seed_a"], n)
    b = o.generate(entry["seed_b"], n)
    if "a" in entry:
        assert o.from_mont(a) == entry["a"] and o.from_mont(b) == entry["b"]
    ch = np.array([o.challenge(entry["seed_r"], j + 1) for j in range(n)], dtype=np.uint64)
    assert o.from_mont(ch) == entry["challenges"]
    res = o.prove(a, b, ch)
    assert res["status"] == 0
    assert o.from_mont1(res["c_1"]) == entry["c_1"]
    assert [o.from_mont(row) for row in res["evals"]] == entry["evals"]
    assert [o.from_mont(row) for row in res["coeffs"]] == entry["coeffs"]
    assert o.from_mont1(res["final_eval"]) == entry["final_eval"]


@pytest.mark.parametrize("entry", load("mle_vectors.json"),
                         ids=lambda e: "p%d-n%d" % (e["p"] if e["p"] < 2**40 else 0, e["n"]))
def test_mle_vectors(entry):
    p, n = entry["p"], entry["n"]
    o = Oracle(p)
    t = o.generate(entry["seed"], n)
    pt = o.to_mont(entry["point"])
    assert o.from_mont1(o.evaluate(t, pt)) == entry["evaluate_le"]
    assert o.from_mont1(o.vsbw(t, pt)) == entry["evaluate_be"]
    if n <= 8:
        assert o.from_mont1(o.cti(t, pt)) == entry["evaluate_be"]
    k = entry["k"]
    if entry["fix_le"] is not None:
        assert o.from_mont(o.fix_variables(t, pt[:k], 0)) == entry["fix_le"]
        assert o.from_mont(o.fix_variables(t, pt[:k], 1)) == entry["fix_be"]


@pytest.mark.parametrize("p", [5, 389, 1572869, GOLD, 2**64 - 59])
def test_c_oracle_vs_pyref_random(p):
    rng = random.Random(p % 1000)
    o = Oracle(p)
    for n in (1, 2, 3, 4, 7):
        a_c = [rng.randrange(p) for _ in range(1 << n)]
        b_c = [rng.randrange(p) for _ in range(1 << n)]
        ch_c = [rng.randrange(p) for _ in range(n)]
        t = pyref.transcript(a_c, b_c, ch_c, p)
        a, b, ch = o.to_mont(a_c), o.to_mont(b_c), o.to_mont(ch_c)
        res = o.prove(a, b, ch)
        assert res["status"] == 0
        assert o.from_mont1(res["c_1"]) == t["c_1"]
        assert [o.from_mont(r) for r in res["evals"]] == t["evals"]
        assert [o.from_mont(r) for r in res["coeffs"]] == t["coeffs"]
        assert o.from_mont1(res["final_eval"]) == t["final_eval"]
        if n >= 2:
            S = pyref.g_grid_sums(a_c, b_c, p)
            assert o.from_mont(o.grid_sums(a, b)) == sum(S, [])
        # relabel + g_new against pyref
    n = 2
    A = [rng.randrange(p) for _ in range(1 << (2 * n))]
    B = [rng.randrange(p) for _ in range(1 << (2 * n))]
    pt = [rng.randrange(p) for _ in range(2 * n)]
    fa, fb = pyref.g_new(n, A, B, pt, p)
    ga, gb = o.g_new(n, o.to_mont(A), o.to_mont(B), o.to_mont(pt))
    assert o.from_mont(ga) == fa and o.from_mont(gb) == fb


def test_prover_rejects_tampering():
    """the verifier restatement does reject a wrong claim (sum-check-protocol/src/lib.rs:286-291)"""
    p = 389
    a_c = [3, 1, 4, 1, 5, 9, 2, 6]
    b_c = [2, 7, 1, 8, 2, 8, 1, 8]
    v = pyref.VerifierRef(3, lambda pt: pyref.g_evaluate(a_c, b_c, pt, p), p)
    pr = pyref.ProverRef(a_c, b_c, p)
    v.set_c_1((pr.c_1 + 1) % p)
    with pytest.raises(ValueError):
        v.round(pr.round(1, 0), 5)


def test_kat_gkr_circuit_and_w():
    """gkr-protocol/src/circuit.rs:259-284 (layer values, mul_1), and the W polynomial of
    gkr-protocol/src/round_polynomial.rs: C oracle == pyref, c_1 == W_i~(r_i), and the reference's
    4-point roots-of-unity interpolation (:78-90) gives the coefficients of the direct form"""
    k = KATS["gkr_circuit_book"]
    p = k["p"]
    layers = [[tuple(g) for g in layer] for layer in k["layers"]]
    vals = pyref.circuit_evaluate(layers, k["inputs"], p)
    assert vals == k["expected_layers"]
    _, mul_t = pyref.wiring_tables(layers[1], 2, p)
    truth = sorted([a, b, c] for a in range(4) for b in range(4) for c in range(4) if mul_t[(((c << 2) | b) << 2) | a])
    assert truth == k["mul_1_true"]
    rng = random.Random(9)
    for q in (p, GOLD):
        o = Oracle(q)
        for li in (0, 1):
            layer = layers[li]
            k_i = (len(layer) - 1).bit_length()
            k_next = (len(vals[li + 1]) - 1).bit_length()
            r_i = [rng.randrange(q) for _ in range(k_i)]
            add, mul = pyref.wiring_fixed(layer, k_next, r_i, q)
            oa, om = o.wiring_fixed(layer, k_next, o.to_mont(r_i))
            assert o.from_mont(oa) == add and o.from_mont(om) == mul
            w_c = [v % q for v in vals[li + 1]]
            ch = [rng.randrange(q) for _ in range(2 * k_next)]
            t = pyref.w_transcript(add, mul, w_c, w_c, ch, q)
            assert t["c_1"] == pyref.mle_evaluate([v % q for v in vals[li]], r_i, q)
            ow = o.to_mont(w_c)
            res = o.w_prove(oa, om, ow, ow, o.to_mont(ch))
            assert res["status"] == 0 and o.from_mont1(res["c_1"]) == t["c_1"]
            assert [o.from_mont(r) for r in res["evals"]] == t["evals"]
            assert o.from_mont1(res["final_eval"]) == t["final_eval"]
            cur = (add, mul, w_c, w_c)
            for j in range(2 * k_next):
                if j:
                    cur = pyref.w_fix_variables(*cur, [ch[j - 1]], q)
                dom = pyref.w_to_univariate_domain(*cur, q)
                mine = list(t["coeffs"][j])
                while mine and mine[-1] == 0:
                    mine.pop()
                assert dom == mine


def test_triangle_restatements():
    """triangle-counting/src/lib.rs: c_1 == 6 * triangles (:296-300), C oracle == pyref on the whole
    transcript, and the 4-point-domain interpolation of :120-132 equals the direct form"""
    rng = random.Random(4)
    for p in (389, 1572869, GOLD):
        o = Oracle(p)
        for k in (1, 2, 3):
            n = 1 << k
            adj = [[0] * n for _ in range(n)]
            for i in range(n):
                for j in range(i + 1, n):
                    adj[i][j] = adj[j][i] = rng.randrange(2)
            flat = sum(adj, [])
            tri = sum(adj[x][y] & adj[y][z] & adj[x][z] for x in range(n) for y in range(n) for z in range(n)) // 6
            ch = [rng.randrange(p) for _ in range(3 * k)]
            t = pyref.tri_transcript(flat, k, ch, p)
            assert t["c_1"] == 6 * tri % p
            res = o.tri_prove(o.to_mont(flat), k, o.to_mont(ch))
            assert res["status"] == 0 and o.from_mont1(res["c_1"]) == t["c_1"]
            assert [o.from_mont(r) for r in res["evals"]] == t["evals"]
            assert o.from_mont1(res["final_eval"]) == t["final_eval"]
            if p == 389 and k <= 2:
                cur = (flat, flat, flat)
                for j in range(3 * k):
                    if j:
                        cur = pyref.tri_fix_variables(*cur, k, [ch[j - 1]], p)
                    dom = pyref.tri_to_univariate_domain(*cur, k, p)
                    e = t["evals"][j]
                    mine = pyref.interpolate_quadratic([(0, e[0]), (1, e[1]), (2, e[2])], p)
                    while mine and mine[-1] == 0:
                        mine.pop()
                    assert dom == mine
    k = KATS["triangle_simple_389"]
    flat = sum(k["adjacency"], [])
    t = pyref.tri_transcript(flat, k["k"], [3, 5, 7, 11, 13, 17], k["p"])
    assert t["c_1"] == k["expected_c_1"]


def test_gridk_matches_grid_and_round_evals():
    """the general k-round grid (test infrastructure for the three-round first pass) agrees with
    the two pinned special cases: k = 2 is sco_g_grid_sums, k = 1 is (H(0), H(1), H(inf))"""
    for p in (GOLD, 389, 5):
        o = Oracle(p)
        for n in (3, 4, 7):
            a, b = o.generate(21, n), o.generate(22, n)
            assert np.array_equal(o.gridk_sums(a, b, 2), o.grid_sums(a, b))
            e = o.round_evals(a, b)          # H(0), H(1), H(2)
            g1 = [int(x) for x in o.gridk_sums(a, b, 1)]
            F = o.lib
            t = F.sco_add(o.fp, g1[1], g1[2])
            assert [g1[0], g1[1], F.sco_sub(o.fp, F.sco_add(o.fp, t, t), g1[0])] == [int(x) for x in e]
            # k = 3: summing the last axis over {0,1} gives the k = 2 grid of the same tables
            g3 = [int(x) for x in o.gridk_sums(a, b, 3)]
            g2 = [int(x) for x in o.grid_sums(a, b)]
            # S3[(3u+v)*3+w] summed over w in {0,1} is NOT S2[3u+v] (different blocks), but the
            # total H(u) = sum_{v,w in {0,1}} S3[u][v][w] equals sum_{v in {0,1}} S2[u][v]
            for u in range(3):
                h3 = 0
                for v in range(2):
                    for w in range(2):
                        h3 = F.sco_add(o.fp, h3, g3[(3 * u + v) * 3 + w])
                assert h3 == F.sco_add(o.fp, g2[3 * u], g2[3 * u + 1])


def test_gridk_four_and_five_rounds_vs_bigint():
    """sco_g_gridk_sums for k = 4, 5 (the grids of the five-round passes; the gloo prover model of tests/test_dist_cpu.py
    runs on them) against a direct big-integer evaluation of the definition:
    S[c] = sum over blocks of 2^k entries of a~(c) b~(c), c_j in {0, 1, inf}, inf = leading coefficient t1 - t0,
    variable j = index bit j, cell index = sum_j c_j 3^(k-1-j)"""
    def grid(a, b, k, p):
        cells = 3 ** k
        S = [0] * cells

        def extend(v):                       # v: 2^k values indexed by entry bits -> 3^k extension values
            ext = {}
            for c in range(cells):
                digits = [(c // 3 ** (k - 1 - j)) % 3 for j in range(k)]
                # sum over the entries with sign: inf on variable j -> (bit_j = 1) - (bit_j = 0); 0 / 1 -> that bit fixed
                total = 0
                for e in range(1 << k):
                    sign = 1
                    for j, d in enumerate(digits):
                        bit = (e >> j) & 1
                        if d == 2:
                            sign *= 1 if bit else -1
                        elif d != bit:
                            sign = 0
                            break
                    total += sign * v[e]
                ext[c] = total % p
            return ext
        for blk in range(len(a) >> k):
            ea = extend([int(x) for x in a[blk << k:(blk + 1) << k]])
            eb = extend([int(x) for x in b[blk << k:(blk + 1) << k]])
            for c in range(cells):
                S[c] = (S[c] + ea[c] * eb[c]) % p
        return S
    for p in (389, GOLD):
        o = Oracle(p)
        # the oracle's words are Montgomery words for GOLD: compare in the canonical domain through its own conversions
        for k, n in ((4, 4), (4, 6), (5, 5), (5, 6)):
            a, b = o.generate(31, n), o.generate(32, n)
            got = [int(o.from_mont1(int(x))) for x in o.gridk_sums(a, b, k)]
            ca = [int(o.from_mont1(int(x))) for x in a]
            cb = [int(o.from_mont1(int(x))) for x in b]
            assert got == grid(ca, cb, k, p), (p, k, n)
