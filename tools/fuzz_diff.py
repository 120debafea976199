"""Randomized differential run of the library against the oracle (checker only): random prime moduli from 3 to just below
2^64, random sizes, random schedules (every planner option), one device or a multi-device handle, random tables (uniform,
all p-1, all zero, single spike) and challenges (uniform and the corner values 0, 1, p-1).  Per iteration: sc_prove through
a draw callback, the round-by-round prover, c_1, g(r), evaluate LE / BE, fix_variables LE / BE, evaluate_many,
restrict_to_line; one GKR layer (wiring, dense and per-gate W prover, W::evaluate); one triangle-counting proof -
all bit for bit.   usage: fuzz_diff.py [seconds = 120] [seed = 1] [max_n = 15]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from util import oracle, pyref   # the checker

pkg = load_package()
SPECIAL = [3, 5, 7, 389, 1572869, 2**31 - 1, 2**32 + 15, 2**61 - 1, 2**63 + 29, 2**64 - 59, 2**64 - 2**32 + 1, 2**64 - 2**32 + 1]


def is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d, s = d // 2, s + 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def rand_prime(rng):
    if rng.random() < 0.5:
        return rng.choice(SPECIAL)
    bits = rng.randint(2, 64)
    while True:
        c = rng.getrandbits(bits) | 1 | (1 << (bits - 1))
        if c >= 3 and is_prime(c):
            return c


def rand_table(rng, nprng, p, size):
    kind = rng.random()
    if kind < 0.70:
        t = nprng.integers(0, p, size=size, dtype=np.uint64) if p < 2**63 else (
            nprng.integers(0, 2**64, size=size, dtype=np.uint64, endpoint=False) % np.uint64(p))
    elif kind < 0.80:
        t = np.full(size, p - 1, dtype=np.uint64)
    elif kind < 0.88:
        t = np.zeros(size, dtype=np.uint64)
    else:
        t = np.zeros(size, dtype=np.uint64)
        t[rng.randrange(size)] = rng.randrange(p)
    return np.ascontiguousarray(t)


def rand_elem(rng, F):
    k = rng.random()
    if k < 0.08:
        return 0
    if k < 0.16:
        return F.one
    if k < 0.24:
        return F.p - 1
    return rng.randrange(F.p)


def one_iteration(rng, nprng, max_n, stats):
    p = rand_prime(rng)
    n = rng.randint(1, max_n) if rng.random() < 0.7 else rng.randint(min(14, max_n), max_n)   # (the matrix-core first pass: from 2^14)
    F = pkg.Field(p)
    o = oracle(p)
    n_dev = rng.choice([0, 0, 0, 1, 2, 4, 8])
    while n_dev > (1 << n):
        n_dev //= 2
    ctx = pkg.Context(F, devices=[0] * n_dev) if n_dev else pkg.Context(F)
    opts = {}
    if rng.random() < 0.8:
        opts = {"vars_per_pass": rng.choice([1, 2]), "first_pass_vars": rng.choice([0, 1, 2, 3, 4, 4]), "grid_pass": rng.randint(0, 1),
                "grid_log": rng.choice([0, 3, 6, 9, 12, 16, 20]), "grid_max_vars": rng.randint(1, 5), "tail_log": rng.choice([0, 2, 5, 9, 14]),
                "max_blocks": rng.choice([1, 2, 3, 7, 64, 256, 1024]), "grid_blocks": rng.choice([0, 0, 1, 2, 5, 64]),
                "gram_log": rng.choice([0, 14, 15, 17, 28]), "host_tail_log": rng.choice([0, 0, 2, 5, 8, 10, 11, 12, 12]),
                # (round 5: the five-round fold - (4, 5) and its (5, ks) form - at sizes the fuzz reaches)
                "wfold_log": rng.choice([0, 16, 40, 40]), "wfold_min_log": rng.choice([12, 12, 14, 21]), "wfold_always": rng.randint(0, 1),
                "wfold5_min_log": rng.choice([12, 12, 15, 24])}
        if n >= 14 and rng.random() < 0.3:      # a set the five-round fold can run under (it needs the default two-round schedule around it)
            opts = {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": rng.randint(0, 1), "wfold5_min_log": rng.choice([12, 12, 24]),
                    "host_tail_log": rng.choice([0, 5, 9, 11, 12]), "grid_log": rng.choice([12, 16, 20]), "max_blocks": rng.choice([1, 3, 7, 64, 1024]),
                    "grid_blocks": rng.choice([0, 0, 2, 64]), "nt_load_log": rng.choice([12, 22])}
        if not n_dev and rng.random() < 0.2:
            opts["use_mailbox"] = 0
        for k, v in opts.items():
            ctx.set_option(k, v)
    desc = "p=%d n=%d devices=%d opts=%s" % (p, n, n_dev, opts)
    stats["last"] = desc
    ha, hb = rand_table(rng, nprng, p, 1 << n), rand_table(rng, nprng, p, 1 << n)
    a = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, ha)
    b = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, n, hb)
    assert np.array_equal(a.to_evaluations(), ha), desc
    g = pkg.matrix_multiplication.G(a, b)
    ch = np.array([rand_elem(rng, F) for _ in range(n)], dtype=np.uint64)
    ref = o.prove(ha, hb, ch)
    assert ref["status"] == 0
    # the whole loop in the library, the verifier's draws through the callback
    draws = iter(int(x) for x in ch)
    c1, evals, got_ch = pkg.matrix_multiplication.prove(ctx, g, 0, draw=lambda user, j, e: next(draws))
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]) and np.array_equal(got_ch, ch), desc
    # round by round (Prover::new, c_1, round(r_prev, j)), two provers interleaved now and then
    pr = g.native_prover()
    pr2 = g.native_prover() if rng.random() < 0.3 else None
    assert pr.c1() == ref["c_1"], desc
    r_prev = F.one
    for j in range(n):
        e = pr.round_evals(r_prev, j)
        assert e == [int(x) for x in ref["evals"][j]], (desc, j)
        if pr2 is not None:
            assert pr2.round_evals(r_prev, j) == e, (desc, j, "second prover")
        r_prev = int(ch[j])
    del pr, pr2
    assert g.hypercube_sum() == ref["c_1"], desc
    pt = [int(x) for x in ch]
    assert g.evaluate(pt) == ref["final_eval"] == o.g_evaluate(ha, hb, ch), desc
    # the table calls
    assert a.evaluate(pt) == o.evaluate(ha, ch), desc
    assert a.evaluate(pt, pkg.ORDER_BE) == o.vsbw(ha, ch), desc
    local = n - max(n_dev, 1).bit_length() + 1
    k = rng.randint(0, local)
    assert np.array_equal(a.fix_variables(pt[:k]).to_evaluations(), o.fix_variables(ha, ch[:k])), (desc, k)
    if n_dev <= 1:
        k = rng.randint(0, n)
        assert np.array_equal(a.fix_variables(pt[:k], pkg.ORDER_BE).to_evaluations(), o.fix_variables(ha, ch[:k], 1)), (desc, "BE", k)
    m = rng.randint(1, 20)
    pts = [[rand_elem(rng, F) for _ in range(n)] for _ in range(m)]
    order = rng.choice([pkg.ORDER_LE, pkg.ORDER_BE])
    want = [o.evaluate(ha, np.array(q, dtype=np.uint64)) if order == pkg.ORDER_LE else o.vsbw(ha, np.array(q, dtype=np.uint64)) for q in pts]
    assert a.evaluate_many(pts, order) == want, (desc, m, order)
    if p > n and n <= 10:
        bb, cc = [rand_elem(rng, F) for _ in range(n)], [rand_elem(rng, F) for _ in range(n)]
        poly = pkg.gkr_protocol.restrict_poly(bb, cc, a)
        want = pyref.restrict_poly(F.to_ints(bb), F.to_ints(cc), F.to_ints(ha.tolist()), p)
        got = [0] * (n + 1)
        for d, v in poly.coeffs:
            got[d] = F.to_int(v)
        while len(got) > 1 and got[-1] == 0:
            got.pop()
        assert got == want, (desc, "restrict_poly")
    stats["n_by_dev"][n_dev] = stats["n_by_dev"].get(n_dev, 0) + 1
    popts = {k: v for k, v in opts.items() if k not in ("max_blocks", "grid_blocks", "nt_load_log")}
    plan = pkg.schedule.plan_proof(n, n_dev, "local", **popts) if n_dev > 1 else pkg.schedule.plan_proof(n, **popts)
    if plan[0]["action"] == "gram_pass":
        stats["gram"] = stats.get("gram", 0) + 1
    for st in plan:
        if st["action"] == "wfold_pass":
            key = "wfold%d" % st["kf"]
            stats[key] = stats.get(key, 0) + 1
    del a, b, g
    ctx.close()


def make_ctx(rng, F, size_log):
    n_dev = rng.choice([0, 0, 1, 2, 4, 8])
    while n_dev > (1 << size_log):
        n_dev //= 2
    ctx = pkg.Context(F, devices=[0] * n_dev) if n_dev else pkg.Context(F)
    opts = {}
    if rng.random() < 0.7:
        opts = {"grid_pass": rng.randint(0, 1), "grid_log": rng.choice([0, 3, 6, 9, 12, 16, 20]), "grid_max_vars": rng.randint(1, 5),
                "max_blocks": rng.choice([1, 2, 3, 7, 64, 256, 1024]), "grid_blocks": rng.choice([0, 0, 1, 2, 5, 64])}
        for k, v in opts.items():
            ctx.set_option(k, v)
    return ctx, n_dev, opts


def gkr_iteration(rng, nprng, stats):
    """one GKR layer: wiring tables, the dense W prover and the per-gate prover against the oracle's W (round_polynomial.rs:47-119)"""
    gp = pkg.gkr_protocol
    p = rand_prime(rng)
    F = pkg.Field(p)
    o = oracle(p)
    k_i, k_next = rng.randint(1, 6), rng.randint(1, 6)
    ctx, n_dev, opts = make_ctx(rng, F, k_next)
    desc = "gkr p=%d k_i=%d k_next=%d devices=%d opts=%s" % (p, k_i, k_next, n_dev, opts)
    stats["last"] = desc
    layer = [(rng.choice(["add", "mul"]), rng.randrange(1 << k_next), rng.randrange(1 << k_next)) for _ in range(1 << k_i)]
    circuit = gp.Circuit([gp.CircuitLayer([gp.Gate(t, [x, y]) for (t, x, y) in layer])], 1 << k_next)
    inputs = [rand_elem(rng, F) for _ in range(1 << k_next)]
    evaluation = circuit.evaluate(F, inputs)
    r_i = [rand_elem(rng, F) for _ in range(k_i)]
    ch = [rand_elem(rng, F) for _ in range(2 * k_next)]
    oadd, omul = o.wiring_fixed(layer, k_next, r_i)
    ow = np.array(evaluation[1], dtype=np.uint64)
    ref = o.w_prove(oadd, omul, ow, ow, ch) if 2 * k_next <= 12 else None
    try:
        w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i)
    except pkg.SumcheckHipError as e:
        if e.code == 6:   # SC_ERR_UNSUPPORTED: a documented refusal (shapes too small to split over the handle's devices)
            stats["unsupported"] = stats.get("unsupported", 0) + 1
            ctx.close()
            return
        raise
    engines = [w.native_prover()]
    if n_dev <= 1:
        engines.append(gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i))
    for eng in engines:
        assert eng.c1() == ref["c_1"], desc
        for j in range(2 * k_next):
            assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (desc, j)
    assert w.evaluate(ch) == ref["final_eval"], desc
    stats["gkr"] = stats.get("gkr", 0) + 1
    del engines, w
    ctx.close()


def tri_iteration(rng, nprng, stats):
    """triangle_counting::G (triangle-counting/src/lib.rs:70-166) against the oracle"""
    p = rand_prime(rng)
    F = pkg.Field(p)
    o = oracle(p)
    k = rng.randint(1, 4)
    ctx, n_dev, opts = make_ctx(rng, F, k)
    desc = "triangle p=%d k=%d devices=%d opts=%s" % (p, k, n_dev, opts)
    stats["last"] = desc
    n = 1 << k
    m = [[False] * n for _ in range(n)]
    dens = rng.random()
    for i in range(n):
        for j in range(i + 1, n):
            m[i][j] = m[j][i] = rng.random() < dens
    flat = sum(m, [])
    oadj = o.to_mont([1 if b else 0 for b in flat])
    ch = [rand_elem(rng, F) for _ in range(3 * k)]
    ref = o.tri_prove(oadj, k, ch)
    try:
        g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * k, flat)
        eng = g.native_prover()
        assert eng is not None and eng.c1() == ref["c_1"], desc
    except pkg.SumcheckHipError as e:
        if e.code == 6:   # SC_ERR_UNSUPPORTED
            stats["unsupported"] = stats.get("unsupported", 0) + 1
            ctx.close()
            return
        raise
    for j in range(3 * k):
        assert eng.round_evals(ch[j - 1] if j else F.one, j) == [int(x) for x in ref["evals"][j]], (desc, j)
    assert g.evaluate(ch) == ref["final_eval"], desc
    stats["triangle"] = stats.get("triangle", 0) + 1
    del eng, g
    ctx.close()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    stats = {"n_by_dev": {}, "last": ""}
    t0 = time.time()
    it = 0
    try:
        while time.time() - t0 < seconds:
            which = rng.random()
            if which < 0.6:
                one_iteration(rng, nprng, max_n, stats)
            elif which < 0.8:
                gkr_iteration(rng, nprng, stats)
            else:
                tri_iteration(rng, nprng, stats)
            it += 1
    except BaseException:
        print("FAILED at iteration %d (seed %d): %s" % (it, seed, stats["last"]), flush=True)
        raise
    print("fuzz_diff: %d iterations in %.0f s, seed %d, max_n %d, 0 mismatches; product-prover iterations by handle size (0 = plain context): %s; GKR layers %d, triangle graphs %d, refused as unsupported %d; proofs that opened with the matrix-core pass (handles included): %d; wfold launches planned: (4,5) %d, (5,ks) %d" % (
        it, time.time() - t0, seed, max_n, dict(sorted(stats["n_by_dev"].items())), stats.get("gkr", 0), stats.get("triangle", 0),
        stats.get("unsupported", 0), stats.get("gram", 0), stats.get("wfold4", 0), stats.get("wfold5", 0)), flush=True)


main()
