"""fiat-shamir/src/lib.rs:216-236 (`it_works`) with the GPU-backed polynomials: a transcript produced
by generate_transcript verifies; a tampered one does not."""
import random

import pytest

from conftest import load_package
from util import GOLD, pid, pyref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p", [5, GOLD], ids=pid)
def test_it_works(p):
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    scp, fs, mm = pkg.sum_check_protocol, pkg.fiat_shamir, pkg.matrix_multiplication
    rng = random.Random(1)
    for n in range(2, 10):
        a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, F.from_ints([rng.randrange(p) for _ in range(1 << n)]))
        b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, F.from_ints([rng.randrange(p) for _ in range(1 << n)]))
        g = mm.G(a, b)
        hasher = fs.Sha256FieldHasher(F)
        transcript = fs.generate_transcript(scp.Prover.new(g.clone()), hasher)
        assert len(transcript.g) == n
        assert fs.verify_transcript(transcript, scp.Verifier.new(n, g), hasher)
        # round-trip of the wire format
        c_1, off = fs.deserialize_field(F, transcript.g[0], 0)
        poly, end = fs.deserialize_poly(F, transcript.g[0], off)
        assert end == len(transcript.g[0]) and fs.serialize_field(F, c_1) + fs.serialize_poly(poly) == transcript.g[0]
        # tampering with any message is caught (claim mismatch, failed final check, or codec error)
        for j in (0, n - 1):
            bad = [bytes(x) for x in transcript.g]
            k = len(bad[j]) - 1
            bad[j] = bad[j][:k] + bytes([(bad[j][k] + 1) % (p if p < 256 else 256)])
            try:
                ok = fs.verify_transcript(fs.FiatShamirTranscript(bad), scp.Verifier.new(n, g), hasher)
            except (scp.Error, AssertionError):
                ok = False
            assert not ok


def test_triangle_and_gkr_polynomials_through_fiat_shamir():
    """the transform is generic over SumCheckPolynomial: run it over triangle_counting::G and GKR's W"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(389))
    F = ctx.field
    scp, fs = pkg.sum_check_protocol, pkg.fiat_shamir
    adj = [[False, True, True, False], [True, False, True, False], [True, True, False, False], [False] * 4]
    g = pkg.triangle_counting.G.new_adj_matrix(ctx, 4, sum(adj, []))
    hasher = fs.Sha256FieldHasher(F, dst=b"thaler-study")
    tr = fs.generate_transcript(scp.Prover.new(g.clone()), hasher)
    assert fs.verify_transcript(tr, scp.Verifier.new(g.num_vars(), g), hasher)
    # a verifier deriving its challenges with a different domain-separation tag disagrees with the prover
    try:
        ok = fs.verify_transcript(tr, scp.Verifier.new(g.num_vars(), g), fs.Sha256FieldHasher(F, dst=b"other"))
    except (scp.Error, AssertionError):
        ok = False
    assert not ok


# ---- the bytes, against the oracle's ---------------------------------------------------
# tests/golden/fs_transcripts.json is written by oracle/gen_golden.py from oracle/fs_ref.py - an independent restatement of
# fiat-shamir/src/lib.rs:44-98 and of the arkworks wire format on canonical integers (tests/test_oracle_fs.py pins it on the
# CPU).  Here every message of a non-interactive proof produced over the GPU provers must equal the fixture's, byte for byte:
# c_1, the canonical term list of every round polynomial, and - through the hash chain - every challenge.  Byte identity
# with arkworks itself stays unpinned (no Rust in this image; the reference asserts accept / reject only, :231-234).
from util import load_golden  # noqa: E402

FS_FIX = load_golden("fs_transcripts.json")


def _fs_id(c):
    return "%s-%s-%s" % (c["kind"], pid(c["p"]), c.get("n", c.get("var_len", c.get("layer"))))


def _gpu_polynomial(pkg, ctx, c):
    F = ctx.field
    if c["kind"] == "matmul":
        a = pkg.DenseMultilinearExtension.generate(ctx, c["seed_a"], c["n"])
        b = pkg.DenseMultilinearExtension.generate(ctx, c["seed_b"], c["n"])
        return pkg.matrix_multiplication.G(a, b)
    if c["kind"] == "triangle":
        return pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * c["var_len"], [bool(x) for x in c["adjacency"]])
    gp = pkg.gkr_protocol
    circuit = gp.Circuit([gp.CircuitLayer([gp.Gate(t, [a, b]) for (t, a, b) in layer]) for layer in c["layers"]], len(c["inputs"]))
    evaluation = circuit.evaluate(F, F.from_ints(c["inputs"]).tolist())
    return gp.start_round_w(ctx, circuit, evaluation, c["layer"], [F.from_int(x) for x in c["r_i"]])


@pytest.mark.parametrize("native", [True, False], ids=["engine", "trait"])
@pytest.mark.parametrize("c", FS_FIX["cases"], ids=_fs_id)
def test_transcript_bytes_equal_the_oracles(c, native):
    """`engine`: Prover::round on the device-side prover (sc_prover / sc_tri_prover / sc_gkr_prover);
    `trait`: the generic path, fix_variables + to_univariate per round (sum-check-protocol/src/lib.rs:105-112)"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(c["p"]))
    F = ctx.field
    scp, fs = pkg.sum_check_protocol, pkg.fiat_shamir
    g = _gpu_polynomial(pkg, ctx, c)
    prover = scp.Prover.new(g.clone())
    if not native:
        if prover._engine is None:
            pytest.skip("no device-side engine for this polynomial: the trait path is the other case")
        prover._engine = None
        prover.c_1_value = g.hypercube_sum(F)
    else:
        assert prover._engine is not None
    hasher = fs.Sha256FieldHasher(F, dst=c["dst"].encode())
    tr = fs.generate_transcript(prover, hasher)
    assert [m.hex() for m in tr.g] == c["messages"]
    assert [F.to_int(r) for r in prover.r] == c["challenges"]
    assert fs.verify_transcript(tr, scp.Verifier.new(g.num_vars(), g, strict=False), hasher)


def test_explicit_zero_term_reaches_the_wire():
    """F_5, H(0) = 0 and exactly one of H(1), H(2) non-zero: the reference's three-term sum keeps a (0, 0) term
    (tests/test_oracle_fs.py); the GPU-backed G must put the same three terms on the wire, not the two non-zero ones"""
    import fs_ref as FS
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(5))
    F = ctx.field
    found = 0
    for a0, a1, b0, b1 in [(0, 1, 1, 0), (1, 3, 0, 1), (4, 2, 0, 1), (1, 2, 3, 4)]:      # H = (0,0,3), (0,3,0), (0,2,0), and one with H(0) != 0
        a, b = [a0, a1], [b0, b1]
        e = pyref.g_round_evals(a, b, 5)
        terms = FS.lagrange_quadratic([(0, e[0]), (1, e[1]), (2, e[2])], 5)
        g = pkg.matrix_multiplication.G(pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 1, F.from_ints(a)),
                                        pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 1, F.from_ints(b)))
        poly = pkg.sum_check_protocol.Prover.new(g.clone()).round(F.one, 0)
        assert [(d, F.to_int(x)) for d, x in poly.coeffs] == terms
        assert pkg.fiat_shamir.serialize_poly(poly) == FS.ser_sparse(terms, 5)
        found += any(x == 0 for _, x in terms)
    assert found == 3
