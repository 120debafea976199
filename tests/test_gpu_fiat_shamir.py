"""fiat-shamir/src/lib.rs:216-236 (`it_works`) with the GPU-backed polynomials: a transcript produced
by generate_transcript verifies; a tampered one does not."""
import random

import pytest

from conftest import load_package
from util import GOLD, pid

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
