"""n=28 (or argv) proof time for several resident_log settings"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
a, b = syn.tables(ctx, n)
g = mm.G(a, b)
ref = None
for rep in range(2):
    for res, rl in [(0, 0), (1, 19), (1, 21), (1, 23), (1, 25)]:
        ctx.set_option("resident", res)
        if res:
            ctx.set_option("resident_log", rl)
        r = mm.prove(ctx, g, syn.SEED_R)
        if ref is None:
            ref = r
        same = r[0] == ref[0] and np.array_equal(r[1], ref[1])
        for _ in range(30):
            mm.prove(ctx, g, syn.SEED_R)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter()
            mm.prove(ctx, g, syn.SEED_R)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        print("resident=%d log=%2d same=%s median %.4f ms  min %.4f" % (res, rl, same, ts[len(ts) // 2], ts[0]), flush=True)
