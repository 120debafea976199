"""The largest tables one MI355X holds: n = 31, 32, 33 (2 x 64 GiB at n = 33, index arithmetic past 2^32 entries).
Size-independent properties only (the oracle would need hours): the round identities of sum-check-protocol/src/lib.rs:286-328
over the whole transcript, g_n(r_n) == g(r) through sc_prod2_evaluate, LE/BE evaluate against fix_variables + evaluate of the
folded table, evaluate_many against single evaluations.   usage: probe_max_size.py [n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from util import oracle, pyref   # (the oracle only checks: rows and columns of generated matrices)

pkg = load_package()
GOLD = pkg.GOLDILOCKS


def check(n, modulus=GOLD):
    ctx = pkg.Context(pkg.Field(modulus))
    F = ctx.field
    t0 = time.perf_counter()
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    ctx.synchronize()
    t_gen = time.perf_counter() - t0
    # the same tables again from the pool's blocks: generation without the allocation
    del a, b
    t0 = time.perf_counter()
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    ctx.synchronize()
    t_gen2 = time.perf_counter() - t0
    g = pkg.matrix_multiplication.G(a, b)
    t0 = time.perf_counter()
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    t_prove = time.perf_counter() - t0
    t0 = time.perf_counter()
    final = g.evaluate([int(x) for x in ch])
    t_eval = time.perf_counter() - t0
    inv2 = F.inv(F.two)
    claim = c1
    for j in range(n):
        e = [int(x) for x in evals[j]]
        assert F.add(e[0], e[1]) == claim, "round %d" % j
        r = int(ch[j])
        l0 = F.mul(F.mul(F.sub(r, F.one), F.sub(r, F.two)), inv2)
        l1 = F.neg(F.mul(r, F.sub(r, F.two)))
        l2 = F.mul(F.mul(r, F.sub(r, F.one)), inv2)
        claim = F.add(F.add(F.mul(l0, e[0]), F.mul(l1, e[1])), F.mul(l2, e[2]))
    assert claim == final, "g_n(r_n) != g(r)"
    # a~(point) in both orders against a fold of some variables followed by an evaluate of the rest
    pt = [int(x) for x in ch]
    k = 7
    lo = a.fix_variables(pt[:k])
    assert lo.evaluate(pt[k:]) == a.evaluate(pt)
    del lo
    hi = a.fix_variables(pt[:k], order=pkg.ORDER_BE)
    assert hi.evaluate(pt[k:], order=pkg.ORDER_BE) == a.evaluate(pt, order=pkg.ORDER_BE)
    del hi
    pts = [pt, pt[::-1], [F.one] * n, [0] * n, pt[3:] + pt[:3]]
    many = a.evaluate_many(pts)
    assert [int(x) for x in many] == [a.evaluate(p) for p in pts]
    print("n=%d p=%d: identities ok; generate %.1f ms (again, from the pool: %.1f ms), proof %.2f ms (%.3g mul-adds/s), g(r) %.2f ms" % (
        n, modulus, t_gen * 1e3, t_gen2 * 1e3, t_prove * 1e3, (5 * 2**n - 7) / t_prove, t_eval * 1e3), flush=True)
    del a, b, g


def check_g_new(n):
    """G::new (matrix-multiplication/src/lib.rs:77-92) on 2^(2n)-entry matrices: at a boolean point (i, j) f_a is row i of A,
    f_b column j of B and c_1 = (A B)[i][j] (:340); at a random point entries of f_a / f_b are the matrices' MLEs"""
    ctx = pkg.Context(pkg.Field(GOLD))
    F = ctx.field
    o = oracle(GOLD)
    side = 1 << n
    A = pkg.DenseMultilinearExtension.generate(ctx, 11, 2 * n)
    B = pkg.DenseMultilinearExtension.generate(ctx, 12, 2 * n)
    i, j = 0xA5B7 & (side - 1), 0x9C31 & (side - 1)
    bits = lambda v: [F.one if (v >> t) & 1 else F.zero for t in range(n)]   # noqa: E731
    row_i = o.generate_range(11, i * side, side)
    col_j = np.array([int(o.generate_range(12, k * side + j, 1)[0]) for k in range(side)], dtype=np.uint64)
    dot = 0
    for x, y in zip(row_i.tolist(), col_j.tolist()):
        dot = F.add(dot, F.mul(x, y))
    t0 = time.perf_counter()
    g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, bits(i) + bits(j))
    ctx.synchronize()
    t_new = time.perf_counter() - t0
    assert np.array_equal(g.f_a.to_evaluations(), row_i) and np.array_equal(g.f_b.to_evaluations(), col_j)
    assert g.hypercube_sum() == dot
    pt = [int(o.challenge(pyref.SEED_PT, t)) for t in range(2 * n)]
    g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, pt)
    fa, fb = g.f_a.to_evaluations(), g.f_b.to_evaluations()
    for z in (0, 1, side - 1, 0x1234 & (side - 1), side // 2 + 77):
        assert int(fa[z]) == A.evaluate(bits(z) + pt[:n]) and int(fb[z]) == B.evaluate(pt[n:] + bits(z)), z
    print("G::new on 2^%d-entry matrices: row / column / (A B)[i][j] / MLE entries ok; %.2f ms" % (2 * n, t_new * 1e3), flush=True)


for n in [int(x) for x in sys.argv[1:]] or [31, 32, 33]:
    check(n)
check(31, 18446744073709551557)
check_g_new(15)
check_g_new(16)
