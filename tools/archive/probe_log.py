"""per-launch device times (sc_ctx_launch_log) of one GKR W layer proof (k) and one triangle proof (k)"""
import sys, os, random, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_package()
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
gp = pkg.gkr_protocol
rng = random.Random(1)
kg = int(sys.argv[1]) if len(sys.argv) > 1 else 13
kt = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def show(log):
    tot = 0.0
    for r in log:
        nb = r["bytes_read"] + r["bytes_written"]
        print("   %-14s kf=%2d ks=%2d log_in=%2d  %9.1f us  %8.1f GB/s" % (r["kind"], r["kf"], r["ks"], r["log_in"], r["ms"] * 1e3, nb / max(r["ms"], 1e-9) / 1e6))
        tot += r["ms"]
    print("   total device time %.1f us in %d launches" % (tot * 1e3, len(log)))


n_gates = 1 << kg
layer = [gp.Gate(rng.choice(["add", "mul"]), [rng.randrange(n_gates), rng.randrange(n_gates)]) for _ in range(n_gates)]
circuit = gp.Circuit([gp.CircuitLayer(layer)], n_gates)
inputs = [F.from_int(rng.randrange(F.p)) for _ in range(n_gates)]
evaluation = [None, inputs]
r_i = [F.from_int(rng.randrange(F.p)) for _ in range(kg)]
w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i)
ch = [F.from_int(rng.randrange(F.p)) for _ in range(2 * kg)]
for which in ("dense", "sparse"):
    for rep in range(2):
        ctx.set_option("time_kernels", rep)
        ctx.launch_log(reset=True)
        t0 = time.perf_counter()
        eng = w.native_prover() if which == "dense" else gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i)
        for j in range(2 * kg):
            eng.round_evals(ch[j - 1] if j else F.one, j)
        dt = time.perf_counter() - t0
    print("GKR W k=%d %s: wall %.3f ms (with event records)" % (kg, which, dt * 1e3))
    show(ctx.launch_log(reset=True))
    ctx.set_option("time_kernels", 0)
    del eng
n = 1 << kt
m = np.zeros((n, n), dtype=bool)
iu = np.triu_indices(n, 1)
m[iu] = np.random.RandomState(kt).rand(len(iu[0])) < 0.3
m = m | m.T
g = pkg.triangle_counting.G.new_adj_matrix(ctx, 2 * kt, m.flatten().tolist())
ch = [F.from_int(rng.randrange(F.p)) for _ in range(3 * kt)]
for rep in range(2):
    ctx.set_option("time_kernels", rep)
    ctx.launch_log(reset=True)
    t0 = time.perf_counter()
    eng = g.native_prover()
    for j in range(3 * kt):
        eng.round_evals(ch[j - 1] if j else F.one, j)
    dt = time.perf_counter() - t0
print("triangle k=%d: wall %.3f ms" % (kt, dt * 1e3))
show(ctx.launch_log(reset=True))
