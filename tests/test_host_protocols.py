"""Host-side protocol logic that needs no GPU: the sumcheck verifier's strict mode (forged c_1),
BooleanHypercube, the Fiat-Shamir hasher against RFC 9380's public vectors and the
SparsePolynomial canonical form, and the oracle's GKR message loop against the reference's
protocol known answers."""
import random

import pytest

from conftest import load_package
from util import load_golden, pyref

BOOK = [[("mul", 0, 1), ("mul", 2, 3)], [("mul", 0, 0), ("mul", 1, 1), ("mul", 1, 2), ("mul", 3, 3)]]
THREE = [[("add", 0, 1), ("add", 2, 3)], [("add", 0, 1), ("add", 2, 3), ("add", 4, 5), ("add", 6, 7)]]


def gkr_draw_count(layers, num_inputs):
    k = [(len(l) - 1).bit_length() for l in layers] + [(num_inputs - 1).bit_length()]
    return k[0] + sum(2 * k[i + 1] + 1 for i in range(len(layers)))


@pytest.mark.parametrize("layers,num_inputs,inputs,outputs", [
    (BOOK, 4, [3, 2, 3, 1], [36, 6]),              # gkr-protocol/src/lib.rs:550-624 (protocol_test_from_book)
    (THREE, 8, [0, 1] * 4, [2, 2]),                # :626-702 (three_layer_protocol_test)
])
def test_oracle_gkr_protocol_known_answers(layers, num_inputs, inputs, outputs):
    """the oracle's restatement of the GKR message loop accepts on the reference's two protocol
    tests and reproduces their asserted outputs, for many draws of the verifier's randomness"""
    p = 389
    for seed in range(20):
        rng = random.Random(seed)
        draws = [rng.randrange(p) for _ in range(gkr_draw_count(layers, num_inputs))]
        t = pyref.gkr_transcript(layers, num_inputs, inputs, draws, p)
        assert t["circuit_outputs"] == outputs
        assert t["check_input"]
        for i, layer in enumerate(t["layers"]):
            assert len(layer["q"]) <= (len(layers[i + 1]) if i + 1 < len(layers) else num_inputs).bit_length()
    # a wrong input fails the last check (the claim chain ends in W_d(r_d) = m_d)
    bad = list(inputs)
    bad[0] = (bad[0] + 1) % p
    t = pyref.gkr_transcript(layers, num_inputs, inputs, draws, p)
    assert pyref.mle_evaluate(bad, t["layers"][-1]["r_next"], p) != t["layers"][-1]["m_next"]


class PyProduct:
    """a tiny SumCheckPolynomial over pyref (canonical ints inside, Montgomery words outside)"""

    def __init__(self, pkg, F, a, b):
        self.pkg, self.field, self.a, self.b = pkg, F, list(a), list(b)

    def evaluate(self, point):
        F, p = self.field, self.field.p
        if len(point) != self.num_vars():
            return None
        pt = [F.to_int(x) for x in point]
        return F.from_int(pyref.g_evaluate(self.a, self.b, pt, p))

    def fix_variables(self, partial):
        F, p = self.field, self.field.p
        pt = [F.to_int(x) for x in partial]
        return PyProduct(self.pkg, F, pyref.mle_fix_variables(self.a, pt, p), pyref.mle_fix_variables(self.b, pt, p))

    def to_univariate(self):
        F = self.field
        c = pyref.g_to_univariate(self.a, self.b, F.p)
        return self.pkg.sum_check_protocol.SparsePolynomial.from_coefficients_vec(F, [(d, F.from_int(v)) for d, v in enumerate(c)])

    def num_vars(self):
        return (len(self.a) - 1).bit_length()

    def to_evaluations(self):
        return [self.field.from_int(v) for v in pyref.g_to_evaluations(self.a, self.b, self.field.p)]

    def hypercube_sum(self, field):
        acc = 0
        for v in self.to_evaluations():
            acc = field.add(acc, v)
        return acc

    def native_prover(self):
        return None


def forged_run(pkg, F, g, delta, strict):
    """a prover that claims c_1 + delta and stays self-consistent until the last round, where it
    sends the honest g_n (the attack the reference's final branch does not catch)"""
    scp = pkg.sum_check_protocol
    n = g.num_vars()
    honest = scp.Prover.new(g, F)
    verifier = scp.Verifier.new(n, g, F, strict=strict)
    verifier.set_c_1(F.add(honest.c_1(), delta))
    rng = scp.FieldRng(F, random.Random(3))
    r_j, res = F.one, None
    for j in range(n):
        g_j = honest.round(r_j, j)
        if j < n - 1 and delta:
            # shift the constant term so that g_j(0) + g_j(1) equals the (false) running claim:
            # the claim error e_j propagates as e_{j+1} = e_j / 2
            half = F.div(delta, F.from_int(1 << (j + 1)))
            g_j = g_j + scp.SparsePolynomial.from_coefficients_vec(F, [(0, half)])
        res = verifier.round(g_j, rng)
        r_j = res.value
    return res


@pytest.mark.parametrize("p", [389, pyref.GOLDILOCKS])
def test_verifier_strict_rejects_forged_claim(p):
    pkg = load_package()
    scp = pkg.sum_check_protocol
    F = pkg.Field(p)
    rng = random.Random(11)
    n = 4
    g = PyProduct(pkg, F, [rng.randrange(p) for _ in range(1 << n)], [rng.randrange(p) for _ in range(1 << n)])
    ok = forged_run(pkg, F, g, 0, strict=True)
    assert ok.is_final() and ok.value is True
    # reference behaviour (strict=False): the forged transcript is ACCEPTED - sum-check-protocol/src/lib.rs:298-310
    res = forged_run(pkg, F, g, F.from_int(7), strict=False)
    assert res.is_final() and res.value is True
    # strict: the final round also chains g_n(0)+g_n(1) to the previous claim
    with pytest.raises(scp.ProverClaimMismatch):
        forged_run(pkg, F, g, F.from_int(7), strict=True)
    # a wrong final polynomial: reference panics (assert_eq!), strict reports FinalRound(false)
    honest = scp.Prover.new(g, F)
    for strict in (True, False):
        v = scp.Verifier.new(n, g, F, strict=strict)
        v.set_c_1(honest.c_1())
        pr = scp.Prover.new(g, F)
        rr = scp.FieldRng(F, random.Random(9))
        r_j = F.one
        for j in range(n - 1):
            r_j = v.round(pr.round(r_j, j), rr).value
        g_n = pr.round(r_j, n - 1)
        bad = g_n + scp.SparsePolynomial.from_coefficients_vec(F, [(1, F.one), (2, F.neg(F.one))])   # same g(0)+g(1)
        if strict:
            out = v.round(bad, rr)
            assert out.is_final() and out.value is False
        else:
            with pytest.raises(AssertionError):
                v.round(bad, rr)


def test_boolean_hypercube():
    """sum-check-protocol/src/lib.rs:34-70: 2^n points, index bit 0 first; the reference's use is
    `BooleanHypercube::new(n).map(|p| g.evaluate(&p)).sum()` (:192, :220)"""
    pkg = load_package()
    scp = pkg.sum_check_protocol
    F = pkg.Field(389)
    assert list(scp.BooleanHypercube(F, 0)) == [[]]
    pts = list(scp.BooleanHypercube(F, 3))
    assert len(pts) == 8
    for v, pt in enumerate(pts):
        assert pt == [F.one if (v >> i) & 1 else F.zero for i in range(3)]
    rng = random.Random(2)
    n = 5
    a, b = [rng.randrange(389) for _ in range(1 << n)], [rng.randrange(389) for _ in range(1 << n)]
    g = PyProduct(pkg, F, a, b)
    total = 0
    for idx, pt in enumerate(scp.BooleanHypercube(F, n)):
        val = g.evaluate(pt)
        assert F.to_int(val) == a[idx] * b[idx] % 389          # point idx <-> table entry idx (LE)
        total = F.add(total, val)
    assert total == g.hypercube_sum(F)
    it = scp.BooleanHypercube(F, 1)
    assert next(it) == [F.zero] and next(it) == [F.one]
    with pytest.raises(StopIteration):
        next(it)


def test_expand_message_xmd_rfc9380_vectors():
    """fiat-shamir/src/lib.rs:75-98 hashes with DefaultFieldHasher<Sha256,128>; its expander is RFC 9380's
    expand_message_xmd.  Pinned against appendix K.1 in "rfc9380" mode (64-byte Z_pad)."""
    pkg = load_package()
    fs = pkg.fiat_shamir
    kat = load_golden("rfc9380_k1_xmd_sha256.json")
    h = fs.Sha256FieldHasher(pkg.Field(pyref.GOLDILOCKS), dst=kat["DST"].encode(), z_pad="rfc9380")
    for v in kat["vectors"]:
        assert h._expand(v["msg"].encode(), v["len_in_bytes"]).hex() == v["uniform_bytes"], v["msg"][:8]
    # arkworks mode: Z_pad = len_per_base_elem = ceil((bits + 128) / 8)
    for p, lpe in ((5, 17), (389, 18), (pyref.GOLDILOCKS, 24)):
        ha = fs.Sha256FieldHasher(pkg.Field(p))
        assert ha.len_per_elem == lpe and ha.z_pad == lpe
        hr = fs.Sha256FieldHasher(pkg.Field(p), z_pad="rfc9380")
        assert hr.z_pad == 64
        assert ha._expand(b"abc", lpe) != hr._expand(b"abc", lpe)
        # from_be_bytes_mod_order of len_per_elem bytes
        out = ha.hash_to_field(b"abc", 2)
        data = ha._expand(b"abc", 2 * lpe)
        F = pkg.Field(p)
        assert [F.to_int(x) for x in out] == [int.from_bytes(data[:lpe], "big") % p, int.from_bytes(data[lpe:], "big") % p]
    # DSTs longer than 255 bytes are replaced by H("H2C-OVERSIZE-DST-" || DST) (RFC 9380 5.3.3)
    long = fs.Sha256FieldHasher(pkg.Field(5), dst=b"x" * 300)
    assert len(long._expand(b"", 17)) == 17


def test_sparse_polynomial_canonical_form_and_wire_format():
    """ark_poly SparsePolynomial::from_coefficients_vec: terms sorted by degree, zero coefficients
    dropped, so the zero polynomial serialises as an empty Vec (u64 length 0); `Add` merges equal
    degrees (matrix-multiplication/src/lib.rs:55-59 builds the round polynomial that way)."""
    pkg = load_package()
    scp, fs = pkg.sum_check_protocol, pkg.fiat_shamir
    F = pkg.Field(389)
    SP = scp.SparsePolynomial
    z = SP.from_coefficients_vec(F, [(0, 0), (2, 0)])
    assert z.coeffs == [] and z.degree() == 0 and fs.serialize_poly(z) == (0).to_bytes(8, "little")
    c = SP.from_coefficients_vec(F, [(0, F.from_int(7))])
    assert c.coeffs == [(0, F.from_int(7))] and c.degree() == 0
    assert fs.serialize_poly(c) == (1).to_bytes(8, "little") + (0).to_bytes(8, "little") + (7).to_bytes(2, "little")
    u = SP.from_coefficients_vec(F, [(2, F.from_int(3)), (0, F.from_int(1)), (1, 0)])
    assert [d for d, _ in u.coeffs] == [0, 2]
    s = u + SP.from_coefficients_vec(F, [(2, F.from_int(386)), (1, F.from_int(5))])        # 3 + 386 = 0 mod 389
    assert s.coeffs == [(0, F.from_int(1)), (1, F.from_int(5))]
    back, end = fs.deserialize_poly(F, fs.serialize_poly(s))
    assert back == s and end == len(fs.serialize_poly(s))
    # field element = canonical integer, little-endian, ceil(bits / 8) bytes
    assert fs.serialize_field(pkg.Field(5), pkg.Field(5).from_int(3)) == b"\x03"
    G = pkg.Field(pyref.GOLDILOCKS)
    assert fs.serialize_field(G, G.from_int(pyref.GOLDILOCKS - 1)) == (pyref.GOLDILOCKS - 1).to_bytes(8, "little")
    with pytest.raises(fs.SerializationError):
        fs.deserialize_field(G, (pyref.GOLDILOCKS).to_bytes(8, "little"), 0)               # not reduced
    with pytest.raises(fs.SerializationError):
        fs.deserialize_poly(F, (2).to_bytes(8, "little") + b"\x00" * 10)                   # truncated
