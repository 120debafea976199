"""timing probe for the GKR W layer sumcheck and the triangle prover"""
import sys, time, os, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
import numpy as np
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
gp = pkg.gkr_protocol
rng = random.Random(1)
for k in (8, 10, 12, 13):
    n_gates = 1 << k
    layer = [gp.Gate(rng.choice(["add", "mul"]), [rng.randrange(n_gates), rng.randrange(n_gates)]) for _ in range(n_gates)]
    circuit = gp.Circuit([gp.CircuitLayer(layer)], n_gates)
    inputs = [F.from_int(rng.randrange(F.p)) for _ in range(n_gates)]
    evaluation = [None, inputs]
    r_i = [F.from_int(rng.randrange(F.p)) for _ in range(k)]
    t0 = time.perf_counter(); w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i); ctx.synchronize(); t_build = time.perf_counter() - t0
    ch = [F.from_int(rng.randrange(F.p)) for _ in range(2 * k)]
    def prove():
        eng = w.native_prover()
        for j in range(2 * k):
            eng.round_evals(ch[j - 1] if j else F.one, j)
    prove()
    ts = []
    for _ in range(3):
        ctx.synchronize(); t0 = time.perf_counter(); prove(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    nbytes = 2 * 8 * 4**k
    def prove_sparse():
        eng = gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i)   # includes Python-side gate marshalling
        ctx.synchronize(); t1 = time.perf_counter()
        for j in range(2 * k):
            eng.round_evals(ch[j - 1] if j else F.one, j)
        return time.perf_counter() - t1
    prove_sparse()
    ts_sparse = prove_sparse()
    print("   sparse prover, %d rounds: %.3f ms (%.1f us/round)" % (2 * k, ts_sparse * 1e3, ts_sparse * 1e6 / (2 * k)))
    print("GKR W k=%d (add/mul 2^%d entries each): wiring %.2f ms, layer sumcheck %d rounds %.3f ms (%.1f us/round; one read of add+mul = %.1f MB)" % (k, 2*k, t_build*1e3, 2*k, t*1e3, t*1e6/(2*k), nbytes/1e6), flush=True)
for k in (6, 8, 10):
    n = 1 << k
    m = np.zeros((n, n), dtype=bool)
    iu = np.triu_indices(n, 1)
    m[iu] = np.random.RandomState(k).rand(len(iu[0])) < 0.3
    m = m | m.T
    g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * k, m.flatten().tolist())
    ch = [F.from_int(rng.randrange(F.p)) for _ in range(3 * k)]
    def prove():
        eng = g.native_prover()
        for j in range(3 * k):
            eng.round_evals(ch[j - 1] if j else F.one, j)
    prove()
    ctx.synchronize(); t0 = time.perf_counter(); prove(); t = time.perf_counter() - t0
    print("triangle k=%d (%d vertices): %d rounds %.3f ms (n^3 = %.2e field mul-adds in the matrix square)" % (k, n, 3*k, t*1e3, float(n)**3), flush=True)
