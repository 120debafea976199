"""placement selection on/off on one box: K contexts of each kind (so K different placements of the inputs), the device
time of the first folding pass after warm-up, and the proof time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
n, K = int(sys.argv[1]) if len(sys.argv) > 1 else 28, int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
for i in range(K):
    for cands in (1, 4):
        c = pkg.Context(F)
        c.set_option("placement_candidates", cands)
        a, b = syn.tables(c, n)
        g = mm.G(a, b)
        for _ in range(6):
            mm.prove(c, g, syn.SEED_R)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); mm.prove(c, g, syn.SEED_R); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        c.set_option("time_kernels", 1); c.launch_log(reset=True)
        for _ in range(4):
            mm.prove(c, g, syn.SEED_R)
        log = c.launch_log(reset=True); c.set_option("time_kernels", 0)
        per = len(log) // 4
        fold = sum(log[1 + q * per]["ms"] for q in range(4)) / 4 * 1e3
        first = sum(log[q * per]["ms"] for q in range(4)) / 4 * 1e3
        print("inputs %d candidates %d: proof %.4f ms, first pass %.1f us, folding pass %.1f us" % (i, cands, ts[len(ts) // 2], first, fold), flush=True)
        del g, a, b
        c.close()
