"""BASELINE config 5 (gkr-protocol / matrix-multiplication prover on 8 GPUs) at its sharded shape, and the widened
rows at the sizes the profiles are taken at - on ONE GPU: 8 virtual ranks = 8 threads, each with its own context and
shard, joined by in-process host collectives (the control flow of every rank is the 8-GPU run's).

  * sharded G::new (matrix-multiplication/src/lib.rs:77-92) at n = 12 against the oracle and at n = 14 (2^28-entry
    matrices in eight row blocks) through the properties the reference's randomized_test asserts (:316-352);
  * the GKR W prover at world = 8, sharded sc_gkr_wiring and the sharded generic W methods against the oracle;
  * a GKR layer at k = 10 and k = 13 (2^26-entry add/mul tables: gkr_phase1_kernel's streaming variant, the chunked
    grid, fix_low on 2^26 entries) - dense prover == sparse prover every round, c_1 == W_i~(r_i), the verifier's round
    identities and the final evaluate (gkr-protocol/src/round_polynomial.rs:47-119, gkr-protocol/src/lib.rs:373-456);
  * triangle counting at 1 024 vertices: c_1 == 6 * triangles (numpy), round identities, final evaluate
    (triangle-counting/src/lib.rs:120-165, :296-300)."""
import random
import threading

import numpy as np
import pytest

from conftest import load_package
from test_gpu_gkr import make_circuit, random_circuit
from test_gpu_sharded import Loopback
from util import GOLD, challenges, oracle, pyref, verifier_identities

pytestmark = pytest.mark.gpu


def run_ranks(world, body):
    """body(rank, lb) on `world` threads; returns the list of results"""
    lb = Loopback(world)
    results, errors = [None] * world, []

    def wrap(rank):
        try:
            results[rank] = body(rank, lb)
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            errors.append(e)
            lb.barrier.abort()

    threads = [threading.Thread(target=wrap, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    return results


def sharded_ctx(pkg, p, rank, world, lb):
    ctx = pkg.Context(pkg.Field(p))
    ar, ag = lb.collectives(rank)
    ctx.comm_init_host(rank, world, ar, ag)
    return ctx


def test_sharded_g_new_world8_n12_vs_oracle():
    pkg = load_package()
    p, n, world = GOLD, 12, 8
    o = oracle(p)
    pt = np.array([o.challenge(pyref.SEED_PT, j) for j in range(2 * n)], dtype=np.uint64)
    fa, fb = o.g_new(n, o.generate(11, 2 * n), o.generate(12, 2 * n), pt)
    ch = challenges(o, n)
    ref = o.prove(fa, fb, ch)

    def body(rank, lb):
        ctx = sharded_ctx(pkg, p, rank, world, lb)
        start, length = pkg.distributed.shard_range(2 * n, rank, world)
        nl = length.bit_length() - 1
        At = pkg.DenseMultilinearExtension.generate(ctx, 11, nl, start=start)
        Bt = pkg.DenseMultilinearExtension.generate(ctx, 12, nl, start=start)
        g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, At, Bt, [int(x) for x in pt])
        assert g.num_vars() == n
        s0, l0 = pkg.distributed.shard_range(n, rank, world)
        assert np.array_equal(g.f_a.to_evaluations(), fa[s0:s0 + l0])
        assert np.array_equal(g.f_b.to_evaluations(), fb[s0:s0 + l0])
        c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        final = g.evaluate([int(x) for x in ch])
        del g, At, Bt
        ctx.close()
        return c1, evals, final

    for c1, evals, final in run_ranks(world, body):
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]) and final == ref["final_eval"]


def test_sharded_g_new_world8_config5_shape():
    """n = 14: A and B are 2^28-entry matrices, eight row blocks of 2^25 entries each."""
    pkg = load_package()
    F = pkg.Field(GOLD)
    o = oracle(GOLD)
    n, world = 14, 8
    side = 1 << n
    i, j = 0x2A5B & (side - 1), 0x1C37 & (side - 1)
    bits = lambda v: [F.one if (v >> t) & 1 else F.zero for t in range(n)]   # noqa: E731
    row_i = o.generate_range(11, i * side, side)                           # A[i][.]: row-major, column = low bits
    col_j = np.array([int(o.generate_range(12, k * side + j, 1)[0]) for k in range(side)], dtype=np.uint64)
    dot = 0
    for x, y in zip(row_i.tolist(), col_j.tolist()):
        dot = F.add(dot, F.mul(x, y))
    pt = [int(o.challenge(pyref.SEED_PT, t)) for t in range(2 * n)]
    zs = (0, 1, 4097, side - 1, 0x1234, 0x2ABC)

    def body(rank, lb):
        ctx = sharded_ctx(pkg, GOLD, rank, world, lb)
        start, length = pkg.distributed.shard_range(2 * n, rank, world)
        nl = length.bit_length() - 1
        A = pkg.DenseMultilinearExtension.generate(ctx, 11, nl, start=start)
        B = pkg.DenseMultilinearExtension.generate(ctx, 12, nl, start=start)
        s0, l0 = pkg.distributed.shard_range(n, rank, world)
        # boolean point (i, j): f_a is row i of A, f_b column j of B, c_1 = (A B)[i][j]   (:340)
        g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, bits(i) + bits(j))
        assert np.array_equal(g.f_a.to_evaluations(), row_i[s0:s0 + l0])
        assert np.array_equal(g.f_b.to_evaluations(), col_j[s0:s0 + l0])
        assert g.hypercube_sum() == dot
        # random point: every f_a[z] / f_b[z] is the matrix MLE at (z, r1) / (r2, z); the proof passes the verifier
        g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, A, B, pt)
        fa, fb = g.f_a.to_evaluations(), g.f_b.to_evaluations()
        for z in zs:                                                        # sharded evaluates: every rank takes part
            va = A.evaluate(bits(z) + pt[:n])
            vb = B.evaluate(pt[n:] + bits(z))
            if s0 <= z < s0 + l0:
                assert int(fa[z - s0]) == va and int(fb[z - s0]) == vb, z
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        final = g.evaluate([int(x) for x in ch])
        del g, A, B
        ctx.close()
        return fa, fb, c1, evals, ch, final

    res = run_ranks(world, body)
    fa = np.concatenate([r[0] for r in res])
    fb = np.concatenate([r[1] for r in res])
    c1, evals, ch, final = res[0][2:]
    assert verifier_identities(F, c1, evals, ch, final) is None
    ref = o.prove(np.ascontiguousarray(fa), np.ascontiguousarray(fb), ch)     # the oracle's prover on the gathered (f_a, f_b)
    assert ref["status"] == 0 and ref["final_eval"] == final
    for r in res:
        assert r[2] == ref["c_1"] and np.array_equal(r[3], ref["evals"]) and r[5] == final


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_wiring_and_generic_w_methods(world):
    """sc_gkr_wiring and sc_gkr_w_round_sums / _fix_variables / _evaluate on sharded contexts (every rank its rows of c)
    against the oracle, and the dense prover on those shards; sc_gkr_w_to_evaluations returns this rank's shard of the b-major
    result (round 4: add / mul are gathered for the call)"""
    pkg = load_package()
    p = GOLD
    o = oracle(p)
    F0 = pkg.Field(p)
    gp = pkg.gkr_protocol
    rng = random.Random(911)
    g = world.bit_length() - 1
    for ks in ([3, 3], [5, 4], [4, 6], [7, 7]):
        layers = random_circuit(rng, ks)
        circuit = make_circuit(pkg, layers, 1 << ks[-1])
        inputs = [F0.from_int(rng.randrange(p)) for _ in range(1 << ks[-1])]
        evaluation = circuit.evaluate(F0, inputs)
        k_i, k_next = ks[0], ks[1]
        r_i = [F0.from_int(rng.randrange(p)) for _ in range(k_i)]
        oadd, omul = o.wiring_fixed(layers[0], k_next, r_i)
        ow = np.array(evaluation[1], dtype=np.uint64)
        ch = [F0.from_int(rng.randrange(p)) for _ in range(2 * k_next)]
        ref = o.w_prove(oadd, omul, ow, ow, ch)
        assert ref["status"] == 0
        n_loc = oadd.size // world

        def body(rank, lb):
            ctx = sharded_ctx(pkg, p, rank, world, lb)
            w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i)           # sharded wiring + whole W tables
            assert np.array_equal(w.add_i.to_evaluations(), oadd[rank * n_loc:(rank + 1) * n_loc])
            assert np.array_equal(w.mul_i.to_evaluations(), omul[rank * n_loc:(rank + 1) * n_loc])
            got = [w.round_evals()]
            # fix the b variables one at a time, then the shard-local c variables, summing each round generically
            cur = w
            local_rounds = 2 * k_next - g
            for j in range(1, local_rounds):
                cur = cur.fix_variables([ch[j - 1]])
                got.append(cur.round_evals())
            final = w.evaluate(ch)
            if k_next >= g:
                full_ev = o.w_to_evaluations(oadd, omul, ow, ow)
                cnt = full_ev.size // world
                assert np.array_equal(w.to_evaluations(), full_ev[rank * cnt:(rank + 1) * cnt]), (ks, rank)
            eng = w.native_prover()
            eng_got = [eng.c1()] + [eng.round_evals(ch[j - 1] if j else F0.one, j) for j in range(2 * k_next)]
            del eng, cur, w
            ctx.close()
            return got, final, eng_got

        for got, final, eng_got in run_ranks(world, body):
            for j, e in enumerate(got):
                assert e == [int(x) for x in ref["evals"][j]], (ks, j)
            assert final == ref["final_eval"], ks
            assert eng_got[0] == ref["c_1"]
            for j in range(2 * k_next):
                assert eng_got[1 + j] == [int(x) for x in ref["evals"][j]], (ks, j)


def quadratic_at(F, e, r):
    inv2 = F.inv(F.two)
    l0 = F.mul(F.mul(F.sub(r, F.one), F.sub(r, F.two)), inv2)
    l1 = F.neg(F.mul(r, F.sub(r, F.two)))
    l2 = F.mul(F.mul(r, F.sub(r, F.one)), inv2)
    return F.add(F.add(F.mul(l0, e[0]), F.mul(l1, e[1])), F.mul(l2, e[2]))


def fast_random_layer(rng, k_i, k_next):
    kinds = ("add", "mul")
    n_next = 1 << k_next
    return [(kinds[rng.getrandbits(1)], rng.randrange(n_next), rng.randrange(n_next)) for _ in range(1 << k_i)]


@pytest.mark.parametrize("k", [10, 13])
def test_gkr_layer_at_profiled_sizes(k):
    """one GKR layer with 2^k gates over 2^k values: add_i / mul_i are 4^k-entry tables (k = 13: 2 x 512 MiB, the size
    profiles/ holds).  No oracle at this size (it walks 4^k entries per round): the dense engine, the sparse engine
    and the verifier's identities pin each other, c_1 is pinned by W_i~(r_i) and the last claim by W::evaluate."""
    pkg = load_package()
    p = GOLD
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    gp = pkg.gkr_protocol
    rng = random.Random(1000 + k)
    layer = fast_random_layer(rng, k, k)
    circuit = make_circuit(pkg, [layer], 1 << k)
    inputs = [F.from_int(rng.randrange(p)) for _ in range(1 << k)]
    evaluation = circuit.evaluate(F, inputs)
    r_i = [F.from_int(rng.randrange(p)) for _ in range(k)]
    w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i)
    assert w.num_vars() == 2 * k and w.add_i.num_vars() == 2 * k
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    dense = w.native_prover()
    sparse = gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i)
    wi = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, k, np.array(evaluation[0], dtype=np.uint64))
    c1 = dense.c1()
    assert c1 == sparse.c1() == wi.evaluate(r_i)                               # the layer's claim (Thaler, GKR)
    ch = [F.from_int(rng.randrange(p)) for _ in range(2 * k)]
    claim = c1
    for j in range(2 * k):
        rp = ch[j - 1] if j else F.one
        e = dense.round_evals(rp, j)
        assert e == sparse.round_evals(rp, j), j                              # dense == sparse, every round
        assert F.add(e[0], e[1]) == claim, j                                  # g_j(0) + g_j(1) == g_{j-1}(r_{j-1})
        claim = quadratic_at(F, e, ch[j])
    assert claim == w.evaluate(ch)                                            # g_n(r_n) == W(r): sum-check-protocol/src/lib.rs:303
    kinds = [r["kind"] for r in ctx.launch_log(reset=True)]
    ctx.set_option("time_kernels", 0)
    assert "gkr" in kinds and "fix_low" in kinds                              # the streaming phase-1 pass and the one-pass b fix ran
    # the generic path's first round on the same tables agrees with the engine
    assert w.round_evals() == dense_first(pkg, w, F)


def dense_first(pkg, w, F):
    eng = w.native_prover()
    return eng.round_evals(F.one, 0)


def test_triangle_1024_vertices():
    """k = 10: the size profiles/ holds for matsq_tiled_kernel (1.07e9 multiply-adds) and the five-round passes on
    2^20-entry tables"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(GOLD))
    F = ctx.field
    gen = np.random.default_rng(77)
    k = 10
    n = 1 << k
    upper = np.triu(gen.random((n, n)) < 0.25, 1)
    m = upper | upper.T
    a = m.astype(np.int64)
    tri = int(np.trace(a @ a @ a)) // 6
    ev = np.where(m.flatten(), np.uint64(F.one), np.uint64(0)).astype(np.uint64)
    t = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * k, ev)
    g = pkg.triangle_counting.G(t, t, t, k)
    eng = g.native_prover()
    assert eng is not None
    c1 = eng.c1()
    assert F.to_int(c1) == 6 * tri                                            # triangle-counting/src/lib.rs:296-300
    rnd = random.Random(4)
    ch = [F.from_int(rnd.randrange(GOLD)) for _ in range(3 * k)]
    claim = c1
    for j in range(3 * k):
        e = eng.round_evals(ch[j - 1] if j else F.one, j)
        assert F.add(e[0], e[1]) == claim, j
        claim = quadratic_at(F, e, ch[j])
    assert claim == g.evaluate(ch)
    # the reference-shaped generic walk agrees on the first round (2^30 evaluations: one launch)
    eng2 = g.native_prover()
    assert g.round_evals() == eng2.round_evals(F.one, 0)
