"""BASELINE configurations at their stated shapes on one GPU (VERDICT r01 'configs untested'):
  * config 4: n = 28 sharded over 8 ranks - 8 virtual ranks (threads, host transport) with 2^25-entry
    shards, equal to the 1-rank transcript bit for bit, verifier identities, sharded evaluate;
  * config 5: G::new (matrix-multiplication/src/lib.rs:77-92) at n = 8, 10, 12 vs the oracle and at
    n = 14 (2^28-entry matrices) through size-independent properties;
  * the largest instances the 288 GB hold: n = 32 and 33 (entry indices beyond 2^32)."""
import numpy as np
import pytest

from conftest import load_package
from test_gpu_sharded import run_virtual_ranks
from util import GOLD, challenges, oracle, pyref, verifier_identities

pytestmark = pytest.mark.gpu


def test_eight_virtual_ranks_n28():
    pkg = load_package()
    F = pkg.Field(GOLD)
    n, world = 28, 8
    # one rank: the reference transcript for this instance (bit-exact vs the oracle at n <= 24 elsewhere
    # and at n = 28 inside bench.py)
    ctx = pkg.Context(F)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    final = g.evaluate([int(x) for x in ch])
    assert verifier_identities(F, c1, evals, ch, final) is None
    del a, b, g
    ctx.close()
    results, lb = run_virtual_ranks(pkg, GOLD, n, world, tail_log=16, vpp=3)
    for rank, (c1r, evr, chr_, finr, e0, s0) in enumerate(results):
        assert c1r == c1 and np.array_equal(evr, evals) and np.array_equal(chr_, ch), rank
        assert finr == final and s0 == c1, rank
        assert e0 == [int(x) for x in evals[0]], rank
    # 2^25-entry shards: first pass (3 rounds) + folding passes down to the 2^16 gather threshold
    assert lb.n_allreduce >= 4 and lb.n_allgather >= 2


@pytest.mark.parametrize("n", [8, 10, 12])
def test_g_new_vs_oracle(n):
    """n >= 8 takes the one-pass fix_low branch for f_B and the chunked column-dot for f_A"""
    pkg = load_package()
    o = oracle(GOLD)
    ctx = pkg.Context(pkg.Field(GOLD))
    A = pkg.DenseMultilinearExtension.generate(ctx, 11, 2 * n)
    B = pkg.DenseMultilinearExtension.generate(ctx, 12, 2 * n)
    pt = np.array([o.challenge(pyref.SEED_PT, j) for j in range(2 * n)], dtype=np.uint64)
    g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, [int(x) for x in pt])
    fa, fb = o.g_new(n, o.generate(11, 2 * n), o.generate(12, 2 * n), pt)
    assert np.array_equal(g.f_a.to_evaluations(), fa)
    assert np.array_equal(g.f_b.to_evaluations(), fb)
    ch = challenges(o, n)
    ref = o.prove(fa, fb, ch)
    c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])


def test_g_new_config5_shape():
    """n = 14: A, B are 2^28-entry matrices (2 GiB each).  Properties the reference's randomized_test
    asserts (matrix-multiplication/src/lib.rs:316-352): at a boolean point (i, j) f_a is row i of A,
    f_b column j of B and c_1 = (A B)[i][j]; at a random point every f_a[z] / f_b[z] is the matrix MLE
    evaluated at (z, r1) / (r2, z), and the proof on (f_a, f_b) passes the verifier."""
    pkg = load_package()
    F = pkg.Field(GOLD)
    o = oracle(GOLD)
    ctx = pkg.Context(F)
    n = 14
    side = 1 << n
    A = pkg.DenseMultilinearExtension.generate(ctx, 11, 2 * n)
    B = pkg.DenseMultilinearExtension.generate(ctx, 12, 2 * n)
    i, j = 0x2A5B & (side - 1), 0x1C37 & (side - 1)
    bits = lambda v: [F.one if (v >> t) & 1 else F.zero for t in range(n)]   # noqa: E731
    g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, bits(i) + bits(j))
    row_i = o.generate_range(11, i * side, side)                           # A[i][.]: row-major, column = low bits
    col_j = np.array([int(o.generate_range(12, k * side + j, 1)[0]) for k in range(side)], dtype=np.uint64)
    assert np.array_equal(g.f_a.to_evaluations(), row_i)
    assert np.array_equal(g.f_b.to_evaluations(), col_j)
    dot = 0
    for x, y in zip(row_i.tolist(), col_j.tolist()):
        dot = F.add(dot, F.mul(x, y))
    assert g.hypercube_sum() == dot                                        # c_1 == (A B)[i][j]  (:340)
    # random point
    pt = [int(o.challenge(pyref.SEED_PT, t)) for t in range(2 * n)]
    g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, pt)
    fa, fb = g.f_a.to_evaluations(), g.f_b.to_evaluations()
    for z in (0, 1, 2, 4097, side - 1, 0x1234):
        assert int(fa[z]) == A.evaluate(bits(z) + pt[:n])                  # A~(r1, z): row variables are the high bits
        assert int(fb[z]) == B.evaluate(pt[n:] + bits(z))                  # B~(z, r2): column variables are the low bits
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert verifier_identities(F, c1, evals, ch, g.evaluate([int(x) for x in ch])) is None
    ofa, ofb = np.ascontiguousarray(fa), np.ascontiguousarray(fb)
    ref = o.prove(ofa, ofb, ch)
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])


def test_mle_config2_properties_n28():
    """BASELINE configs[1] at the large shape (one 2^28-entry table): size-independent properties of evaluate and
    fix_variables - the BE evaluate (vsbw_/cti_multilinear_from_evaluations) is the LE evaluate at the reversed point;
    fixing k variables and evaluating the rest is the full evaluate, for LE prefixes and BE prefixes; a boolean
    point reads the entry back; relabel(0, 14, 14) is the matrix transpose"""
    pkg = load_package()
    F = pkg.Field(GOLD)
    o = oracle(GOLD)
    ctx = pkg.Context(F)
    n = 28
    t = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    pt = [int(o.challenge(pyref.SEED_PT, j)) for j in range(n)]
    v_le = t.evaluate(pt)
    assert t.evaluate(pt[::-1], pkg.ORDER_BE) == v_le
    for k in (1, 2, 3, 7, 8, 14, 17, 20, 27):
        assert t.fix_variables(pt[:k]).evaluate(pt[k:]) == v_le, k
    rev = pt[::-1]
    for k in (1, 2, 5, 14):
        assert t.fix_variables(rev[:k], pkg.ORDER_BE).evaluate(rev[k:], pkg.ORDER_BE) == v_le, ("BE", k)
    for idx in (0, 1, 12345678, (1 << n) - 1):
        bits = [F.one if (idx >> i) & 1 else F.zero for i in range(n)]
        assert t.evaluate(bits) == int(o.generate_range(pyref.SEED_A, idx, 1)[0])
    tr = t.relabel(0, 14, 14)
    swapped = pt[14:] + pt[:14]
    assert tr.evaluate(swapped) == v_le


@pytest.mark.parametrize("n", [32, 33])
def test_largest_instances(n):
    """the maximum sizes one MI355X holds (2^33-entry tables are 2 x 64 GiB of the 288 GB; entry indices pass 2^32):
    a full proof under the verifier's identities, BE = LE at the reversed point, fix-then-evaluate, and single entries
    read back through boolean points on both sides of the 2^32 boundary"""
    pkg = load_package()
    F = pkg.Field(GOLD)
    o = oracle(GOLD)
    ctx = pkg.Context(F)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    final = g.evaluate([int(x) for x in ch])
    assert verifier_identities(F, c1, evals, ch, final) is None
    pt = [int(o.challenge(pyref.SEED_PT, j)) for j in range(n)]
    v = a.evaluate(pt)
    assert a.evaluate(pt[::-1], pkg.ORDER_BE) == v
    for k in (1, 3, n - 1):
        assert a.fix_variables(pt[:k]).evaluate(pt[k:]) == v, k
    hi = (1 << n) - 1
    for idx in (0, hi, (1 << 32) - 1, (1 << 32) + 5 if n > 32 else hi - 7, (1 << (n - 1)) + 3):
        bits = [F.one if (idx >> d) & 1 else F.zero for d in range(n)]
        assert a.evaluate(bits) == int(o.generate_range(pyref.SEED_A, idx, 1)[0]), idx
    del a, b, g
    ctx.close()
