"""SURVEY.md section 8f row 3 on the CPU: the oracle-side restatement of the non-interactive transform
(oracle/fs_ref.py, following fiat-shamir/src/lib.rs:44-98, :123-171) against the committed byte fixtures, against RFC 9380,
and against the product's host mirror (thaler-study_amd/fiat_shamir.py + sum_check_protocol.SparsePolynomial) - two
implementations that share no code.  The GPU half is tests/test_gpu_fiat_shamir.py."""
import random

import pytest

from conftest import load_package
from util import GOLD, load_golden, pyref

import fs_ref as FS  # noqa: E402  (oracle/ is on sys.path through util)

FIX = load_golden("fs_transcripts.json")


def _prover_and_oracle(c):
    p = c["p"]
    if c["kind"] == "matmul":
        a, b = pyref.synth_table(c["seed_a"], c["n"], p), pyref.synth_table(c["seed_b"], c["n"], p)
        return FS.MatMulProver(a, b, p), c["n"], lambda pt: pyref.g_evaluate(a, b, pt, p)
    if c["kind"] == "triangle":
        adj, k = c["adjacency"], c["var_len"]
        return FS.TriangleProver(adj, k, p), 3 * k, lambda pt: pyref.tri_evaluate(adj, adj, adj, k, pt, p)
    add, mul, w = c["add"], c["mul"], c["w"]
    return FS.WProver(add, mul, w, w, p), (len(add) - 1).bit_length(), lambda pt: pyref.w_evaluate(add, mul, w, w, pt, p)


def _id(c):
    return "%s-p%s-%s" % (c["kind"], "gold" if c["p"] == GOLD else c["p"], c.get("n", c.get("var_len", c.get("layer"))))


@pytest.mark.parametrize("c", FIX["cases"], ids=_id)
def test_fixture_is_what_the_oracle_produces_and_verifies(c):
    prover, n, evaluate = _prover_and_oracle(c)
    dst = c["dst"].encode()
    g, rs = FS.generate_transcript(prover, dst)
    assert [m.hex() for m in g] == c["messages"] and rs == c["challenges"] and len(g) == n
    assert FS.verify_transcript(g, n, evaluate, c["p"], dst)
    # every message matters: flip the last coefficient byte of each in turn (a reduced value stays reduced for these p)
    for j in range(n if c["p"] > 5 else n - 1):      # (the last round's only check is g_n(r_n) = g(r): 1 in 5 over F_5)
        bad = list(g)
        last = bad[j][-1]
        bad[j] = bad[j][:-1] + bytes([last ^ 1])
        try:
            ok = FS.verify_transcript(bad, n, evaluate, c["p"], dst)
        except ValueError:                                                  # not a reduced field element
            ok = False
        assert not ok, "tampered message %d accepted" % j
    # a verifier with another domain-separation tag derives other challenges
    if n > 1 and c["p"] > 5:
        assert not FS.verify_transcript(g, n, evaluate, c["p"], dst + b"x")


def test_wprover_fixture_tables_follow_from_the_circuit():
    """the W cases' add / mul / w tables are what the reference's start_round builds (gkr-protocol/src/lib.rs:388-416)"""
    for c in FIX["cases"]:
        if c["kind"] != "w":
            continue
        layers = [[tuple(g) for g in layer] for layer in c["layers"]]
        vals = pyref.circuit_evaluate(layers, c["inputs"], c["p"])
        k_next = (len(vals[c["layer"] + 1]) - 1).bit_length()
        add, mul = pyref.wiring_fixed(layers[c["layer"]], k_next, c["r_i"], c["p"])
        assert add == c["add"] and mul == c["mul"] and vals[c["layer"] + 1] == c["w"]


def test_expander_against_rfc9380():
    """appendix K.1 (expand_message_xmd, SHA-256): the oracle's expander in its 64-byte Z_pad mode"""
    k = load_golden("rfc9380_k1_xmd_sha256.json")
    dst = k["DST"].encode()
    for v in k["vectors"]:
        out = FS.expand_message_xmd_sha256(v["msg"].encode(), dst, v["len_in_bytes"], 64)
        assert out.hex() == v["uniform_bytes"]


def test_oversize_dst_and_limits():
    assert len(FS.expand_message_xmd_sha256(b"", b"x" * 300, 17, 17)) == 17
    with pytest.raises(ValueError):
        FS.expand_message_xmd_sha256(b"", b"", 255 * 32 + 1, 64)


def test_g_round_polynomial_canonical_forms():
    """matrix-multiplication/src/lib.rs:17-60 adds three SparsePolynomials; arkworks' `+` returns the other operand AS IS
    when one is zero, so a Lagrange term's explicit zero constant survives when H(0) = 0 and exactly one of H(1), H(2) is
    non-zero.  The fixture lists the term lists for all of F_5^3; the product's host mirror must produce the same."""
    pkg = load_package()
    F = pkg.Field(5)
    mm = pkg.matrix_multiplication
    seen_explicit_zero = 0
    for t in FIX["g_round_poly_forms_f5"]:
        e = t["e"]
        terms = [tuple(x) for x in t["terms"]]
        assert FS.lagrange_quadratic([(0, e[0]), (1, e[1]), (2, e[2])], 5) == terms
        poly = mm.interpolate_quadratic_poly(F, [(F.zero, F.from_int(e[0])), (F.one, F.from_int(e[1])), (F.two, F.from_int(e[2]))])
        assert [(d, F.to_int(c)) for d, c in poly.coeffs] == terms
        for x in range(3):                                                  # whatever the form, the values are H's
            assert FS.sparse_eval(terms, x, 5) == e[x]
        explicit = [d for d, c in terms if c == 0]
        if explicit:
            assert explicit == [0] and e[0] == 0 and (e[1] == 0) != (e[2] == 0)
            seen_explicit_zero += 1
        else:
            assert not (e[0] == 0 and (e[1] == 0) != (e[2] == 0))
    assert seen_explicit_zero == 8


@pytest.mark.parametrize("p", [5, 389, 1572869, GOLD], ids=lambda p: "gold" if p == GOLD else "p%d" % p)
def test_host_mirror_and_oracle_agree_on_forms_bytes_and_challenges(p):
    """two restatements, no shared code: canonical forms of sums, the serializer, the field hasher"""
    pkg = load_package()
    F = pkg.Field(p)
    fs, SP = pkg.fiat_shamir, pkg.sum_check_protocol.SparsePolynomial
    rng = random.Random(p)
    small = lambda: rng.choice([0, 0, 1, rng.randrange(p)])     # noqa: E731  (zeros on purpose)
    for _ in range(300):
        ta = [(d, small()) for d in rng.sample(range(5), rng.randrange(0, 4))]
        tb = [(d, small()) for d in rng.sample(range(5), rng.randrange(0, 4))]
        try:
            oa, ob = FS.sparse_from_coefficients_vec(ta), FS.sparse_from_coefficients_vec(tb)
        except AssertionError:          # an unsorted vector whose highest term is zero: arkworks' own assert! fires
            with pytest.raises(AssertionError):
                SP.from_coefficients_vec(F, [(d, F.from_int(c)) for d, c in ta])
                SP.from_coefficients_vec(F, [(d, F.from_int(c)) for d, c in tb])
            continue
        ma = SP.from_coefficients_vec(F, [(d, F.from_int(c)) for d, c in ta])
        mb = SP.from_coefficients_vec(F, [(d, F.from_int(c)) for d, c in tb])
        assert [(d, F.to_int(c)) for d, c in ma.coeffs] == oa
        osum, msum = FS.sparse_add(oa, ob, p), ma + mb
        assert [(d, F.to_int(c)) for d, c in msum.coeffs] == osum
        wire = FS.ser_sparse(osum, p)
        assert fs.serialize_poly(msum) == wire
        back, end = fs.deserialize_poly(F, wire)
        assert back == msum and end == len(wire) and FS.de_sparse(wire, 0, p) == (osum, len(wire))
        dense = [small() for _ in range(4)]
        assert [(d, F.to_int(c)) for d, c in SP.from_dense(F, [F.from_int(c) for c in dense]).coeffs] == FS.sparse_from_dense(dense)
    for dst in (b"", b"thaler-study", b"y" * 256):
        h = fs.Sha256FieldHasher(F, dst=dst)
        for ln in (0, 1, 31, 32, 33, 200):
            msg = bytes(rng.randrange(256) for _ in range(ln))
            assert F.to_int(h.hash_to_field(msg, 1)[0]) == FS.hash_to_field_1(msg, p, dst)
        assert h._expand(b"abc", 100) == FS.expand_message_xmd_sha256(b"abc", dst, 100, h.z_pad)
    x = rng.randrange(p)
    assert fs.serialize_field(F, F.from_int(x)) == FS.ser_field(x, p) and len(FS.ser_field(x, p)) == (p.bit_length() + 7) // 8
