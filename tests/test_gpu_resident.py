"""The resident prover kernel (option "resident"): every pass after the first of a proof served by ONE launch
whose phases the host steers through a pinned command line.  Same transcripts as per-pass launches and as the
oracle; the kernel parks itself when the host is slow, is retired when other work needs the stream, and a
prover dropped mid-proof leaves the context usable."""
import time

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, challenges, oracle, pid, pyref

pytestmark = pytest.mark.gpu


def make(pkg, ctx, n, seed_a=pyref.SEED_A, seed_b=pyref.SEED_B):
    a = pkg.DenseMultilinearExtension.generate(ctx, seed_a, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, seed_b, n)
    return pkg.matrix_multiplication.G(a, b)


@pytest.mark.parametrize("p", [GOLD, 389, 2**64 - 59], ids=pid)
def test_resident_matches_oracle(p):
    pkg = load_package()
    o = oracle(p)
    ctx = pkg.Context(pkg.Field(p))
    ctx.set_option("resident", 1)
    for resident_log, first, tail in [(19, 0, 3), (25, 0, 3), (25, 2, 2), (25, 3, 2)]:
        ctx.set_option("resident_log", resident_log)
        ctx.set_option("first_pass_vars", first)
        ctx.set_option("tail_pass_vars", tail)
        for n in list(range(1, 16)) + [17, 20, 22]:
            g = make(pkg, ctx, n)
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
            kinds = [r["kind"] for r in ctx.launch_log(reset=True)]
            ctx.set_option("time_kernels", 0)
            ref = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), ch)
            assert ref["status"] == 0
            assert c1 == ref["c_1"], (resident_log, first, tail, n)
            assert np.array_equal(evals, ref["evals"]), (resident_log, first, tail, n)
            assert g.evaluate([int(x) for x in ch]) == ref["final_eval"]
            if n >= 8:
                assert "tail_resident" in kinds and len(kinds) <= 4, (n, kinds)
            del g


def test_resident_round_by_round_park_and_retire():
    """sc_prover_round driven by a slow caller: the kernel parks (park_ms) and the proof continues with ordinary
    launches; other calls on the context between rounds retire the kernel; a prover destroyed mid-proof and an
    interleaved second prover leave everything consistent"""
    pkg = load_package()
    p = GOLD
    o = oracle(p)
    F = pkg.Field(p)
    ctx = pkg.Context(F)
    ctx.set_option("resident", 1)
    ctx.set_option("resident_log", 25)
    ctx.set_option("park_ms", 2)
    n = 18
    g = make(pkg, ctx, n)
    ch = challenges(o, n)
    ref = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), ch)
    point = [int(x) for x in ch]

    def run(disturb):
        eng = g.native_prover()
        assert eng.c1() == ref["c_1"]
        got = []
        for j in range(n):
            got.append(eng.round_evals(int(ch[j - 1]) if j else F.one, j))
            disturb(j)
        assert np.array_equal(np.array(got, dtype=np.uint64), ref["evals"])

    run(lambda j: None)                                             # fast caller: all phases resident
    run(lambda j: time.sleep(0.01) if j in (6, 9) else None)        # slow caller: the kernel parks (2 ms) twice
    run(lambda j: g.evaluate(point) if j in (4, 7, 12) else None)   # other work on the stream: retired
    run(lambda j: g.hypercube_sum() if j % 3 == 1 else None)
    # a prover dropped mid-proof, while its kernel waits for a command
    eng = g.native_prover()
    for j in range(7):
        eng.round_evals(int(ch[j - 1]) if j else F.one, j)
    del eng
    run(lambda j: None)
    # two provers interleaved on one context: each call retires the other's kernel
    e1, e2 = g.native_prover(), g.native_prover()
    for j in range(n):
        r = int(ch[j - 1]) if j else F.one
        assert e1.round_evals(r, j) == [int(x) for x in ref["evals"][j]]
        assert e2.round_evals(r, j) == [int(x) for x in ref["evals"][j]]
    # and the context still serves everything else
    assert g.evaluate(point) == ref["final_eval"]
    c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])


def test_resident_under_the_triangle_prover():
    """triangle_counting's engine runs three product sumchecks back to back and folds the sub-prover's tables
    between them (prover_finish): with the resident kernel on"""
    import random
    pkg = load_package()
    p = 1572869
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    o = oracle(p)
    for resident in (0, 1):
        ctx.set_option("resident", resident)
        ctx.set_option("resident_log", 25)
        rng = random.Random(3)
        for k in (2, 4, 6):
            nv = 1 << k
            adj = [[False] * nv for _ in range(nv)]
            for i in range(nv):
                for j2 in range(i + 1, nv):
                    adj[i][j2] = adj[j2][i] = rng.random() < 0.4
            flat = sum(adj, [])
            g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * k, flat)
            ch = [F.from_int(rng.randrange(p)) for _ in range(3 * k)]
            ref = o.tri_prove(np.array(F.from_ints([1 if x else 0 for x in flat]), dtype=np.uint64), k, ch)
            eng = g.native_prover()
            assert eng.c1() == ref["c_1"]
            for j in range(3 * k):
                assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (resident, k, j)
