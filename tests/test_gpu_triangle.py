"""triangle_counting::G on the GPU: the reference's own tests replayed
(triangle-counting/src/lib.rs:232-317) and parity against the oracle restatement."""
import random

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, oracle, pid

pytestmark = pytest.mark.gpu


def random_adj(rng, n):
    """AdjMatrix::new, triangle-counting/src/lib.rs:187-205"""
    m = [[False] * n for _ in range(n)]
    for i in range(n):
        for j in range(i + 1, n):
            m[i][j] = m[j][i] = rng.random() < 0.5
    return m


def triangle_count(m):
    n = len(m)
    return sum(1 for x in range(n) for y in range(n) for z in range(n) if m[x][y] and m[y][z] and m[x][z]) // 6


def run_protocol(pkg, g, rng):
    scp = pkg.sum_check_protocol
    F = g.field
    prover = scp.Prover.new(g.clone())
    c_1 = prover.c_1()
    num_vars = g.num_vars()
    verifier = scp.Verifier.new(num_vars, g)
    verifier.set_c_1(c_1)
    r_j = F.one
    for j in range(num_vars):
        res = verifier.round(prover.round(r_j, j), rng)
        if res.is_final():
            assert res.value
            return c_1
        r_j = res.value
    raise AssertionError("should have returned on FinalRound from verifier")


def test_simple_matrix():
    """triangle-counting/src/lib.rs:232-266"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(389))
    adj = [[False, True, True, False], [True, False, True, False], [True, True, False, False], [False] * 4]
    g = pkg.triangle_counting.G.new_adj_matrix(ctx, len(adj), sum(adj, []))
    rng = pkg.sum_check_protocol.FieldRng(ctx.field, random.Random(1))
    c_1 = run_protocol(pkg, g, rng)
    assert ctx.field.to_int(c_1) == 6


def test_randomized():
    """triangle-counting/src/lib.rs:268-317: c_1 == 6 * triangles for n = 2..128, verifier accepts"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(1572869))
    rng = pkg.sum_check_protocol.FieldRng(ctx.field, random.Random(2))
    gen = random.Random(3)
    for i in range(1, 8):
        n = 1 << i
        m = random_adj(gen, n)
        g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * i, sum(m, []))
        prover = pkg.sum_check_protocol.Prover.new(g.clone())
        assert ctx.field.to_int(prover.c_1()) == 6 * triangle_count(m), "mismatch for size %d" % n
        run_protocol(pkg, g, rng)


@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_vs_oracle(p):
    """engine, generic trait path and every trait method against the oracle, bit for bit"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    o = oracle(p)
    gen = random.Random(7)
    for k in (1, 2, 3, 4):
        n = 1 << k
        m = random_adj(gen, n)
        flat = sum(m, [])
        g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * k, flat)
        oadj = o.to_mont([1 if b else 0 for b in flat])
        ch = [F.from_int(gen.randrange(p)) for _ in range(3 * k)]
        ref = o.tri_prove(oadj, k, ch)
        assert ref["status"] == 0
        assert g.num_vars() == 3 * k
        eng = g.native_prover()
        assert eng is not None and eng.c1() == ref["c_1"]
        for j in range(3 * k):
            assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (k, j)
        assert g.evaluate(ch) == ref["final_eval"] and g.evaluate(ch[:-1]) is None
        # generic path: fix_variables -> to_univariate, one variable at a time
        cur = g
        assert cur.hypercube_sum(F) == ref["c_1"]
        if k <= 3:
            assert np.array_equal(cur.to_evaluations(), o.tri_to_evaluations(oadj, oadj, oadj, k))
        for j in range(3 * k):
            if j:
                cur = cur.fix_variables([ch[j - 1]])
                assert cur.native_prover() is None
            assert cur.num_vars() == 3 * k - j
            assert cur.round_evals() == [int(x) for x in ref["evals"][j]], (k, j)
        # multi-variable fix across the x/y/z boundaries
        for kk in (1, k, k + 1, 2 * k, 2 * k + 1, 3 * k):
            g2 = g.fix_variables(ch[:kk])
            assert g2.num_vars() == 3 * k - kk
            if kk < 3 * k:
                assert g2.evaluate(ch[kk:]) == ref["final_eval"]


def test_larger_graph_identities():
    """k = 8 (256 vertices): engine output against the verifier identities and the triangle count"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(GOLD))
    F = ctx.field
    gen = random.Random(11)
    k = 8
    n = 1 << k
    m = np.zeros((n, n), dtype=bool)
    iu = np.triu_indices(n, 1)
    bits = np.array([gen.random() < 0.3 for _ in range(len(iu[0]))])
    m[iu] = bits
    m = m | m.T
    a = m.astype(np.int64)
    tri = int(np.trace(a @ a @ a)) // 6
    g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * k, m.flatten().tolist())
    rng = pkg.sum_check_protocol.FieldRng(F, random.Random(5))
    c_1 = run_protocol(pkg, g, rng)
    assert F.to_int(c_1) == 6 * tri


@pytest.mark.parametrize("world,transport", [(2, "host"), (4, "host"), (2, "peer")])
def test_triangle_prover_sharded(world, transport):
    """the triangle engine on a sharded context: every rank holds its rows of the adjacency table, computes its
    rows of the matrix square (the n^3 work, split across ranks), the square is gathered and the product sumchecks
    run replicated; c_1 and every round polynomial equal the oracle's on every rank"""
    import threading
    from test_gpu_sharded import Loopback
    pkg = load_package()
    p = 1572869
    o = oracle(p)
    F0 = pkg.Field(p)
    rng = random.Random(9)
    for k in (2, 3, 6, 7):
        nv = 1 << k
        m = random_adj(rng, nv)
        flat = np.array([F0.one if x else F0.zero for row in m for x in row], dtype=np.uint64)
        ch = [F0.from_int(rng.randrange(p)) for _ in range(3 * k)]
        ref = o.tri_prove(flat, k, ch)
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
                n_loc = flat.size // world
                shard = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * k - (world.bit_length() - 1), flat[rank * n_loc:(rank + 1) * n_loc])
                g = pkg.triangle_counting.G(shard, shard, shard, k)
                eng = pkg.triangle_counting._NativeTriProver(g)
                got = [eng.c1()]
                for j in range(3 * k):
                    got.append(eng.round_evals(ch[j - 1] if j else F0.one, j))
                results[rank] = got
                if transport == "peer":
                    lb.barrier.wait()
                del eng, g, shard
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
        assert F0.to_int(ref["c_1"]) == 6 * triangle_count(m)
        for got in results:
            assert got[0] == ref["c_1"], k
            for j in range(3 * k):
                assert got[1 + j] == [int(x) for x in ref["evals"][j]], (k, j)


@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_matrix_square_paths_vs_oracle(p):
    """n >= 64: the square of the adjacency MLE's matrix goes to the int8 matrix cores when the table is 0/1 (what
    G::new_adj_matrix builds) and to the generic field kernel otherwise; both against the oracle's reference-shaped
    prover.  DIRECTED graphs (asymmetric 0/1 matrices) pin the operand orientation of the MFMA tiles - a symmetric
    matrix would pass with rows and columns swapped - and a table with field-valued entries pins the fallback."""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    o = oracle(p)
    gen = random.Random(2027)
    for k, kind in [(6, "directed"), (7, "directed"), (6, "field"), (6, "almost01")]:
        n = 1 << k
        if kind == "directed":
            words = [F.one if gen.random() < 0.4 else 0 for _ in range(n * n)]
        elif kind == "field":
            words = [F.from_int(gen.randrange(p)) for _ in range(n * n)]
        else:   # 0/1 everywhere but one entry: the flag must send the whole square to the generic kernel
            words = [F.one if gen.random() < 0.4 else 0 for _ in range(n * n)]
            words[gen.randrange(n * n)] = F.from_int(2)
        ev = np.array(words, dtype=np.uint64)
        t = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * k, ev)
        g = pkg.triangle_counting.G(t, t, t, k)
        ch = [F.from_int(gen.randrange(p)) for _ in range(3 * k)]
        ref = o.tri_prove(ev, k, ch)
        assert ref["status"] == 0
        eng = g.native_prover()
        assert eng.c1() == ref["c_1"], (k, kind)
        for j in range(3 * k):
            assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (k, kind, j)
        assert g.evaluate(ch) == ref["final_eval"]
        it = iter(ch)
        c1, evals, _ = pkg.triangle_counting.prove(ctx, g, 0, draw=lambda _u, _j, _e: int(next(it)))     # sc_tri_prove: the same proof in one call
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])


@pytest.mark.parametrize("opts", [{}, {"grid_max_vars": 1}, {"vars_per_pass": 1}, {"grid_max_vars": 2, "host_tail_log": 10}, {"host_tail_log": 0},
                                  {"grid_max_vars": 1, "host_tail_log": 6}], ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()) or "default")
def test_engine_over_host_finished_phases(opts):
    """the triangle engine runs three product proofs back to back and takes the partially folded tables of one phase into the
    next (prover_finish).  With the host finishing small proofs (option host_tail_log, round 5) a phase's sub-prover may have
    handed its tables to the host - or, with one-round passes, be in host mode - when its phase ends: every round of every
    phase against the oracle, for schedules that end each phase on the device, in the tail slot and on the host"""
    pkg = load_package()
    p = GOLD
    o = oracle(p)
    ctx = pkg.Context(pkg.Field(p))
    for k_, v in opts.items():
        ctx.set_option(k_, v)
    F = ctx.field
    gen = random.Random(23)
    for k in (2, 3, 5, 6, 7):
        n = 1 << k
        m = random_adj(gen, n)
        flat = sum(m, [])
        g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * k, flat)
        oadj = o.to_mont([1 if b else 0 for b in flat])
        ch = [F.from_int(gen.randrange(p)) for _ in range(3 * k)]
        ref = o.tri_prove(oadj, k, ch)
        assert ref["status"] == 0
        eng = g.native_prover()
        assert eng.c1() == ref["c_1"], (k, opts)
        for j in range(3 * k):
            assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (k, j, opts)
        del eng, g
    ctx.close()
