"""A/B of the LDS-DMA ring first pass (option first_ring, with ring_log lowered) at several n: transcript equality and time"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
ns = [int(x) for x in sys.argv[1:]] or [12, 13, 18, 19, 22, 26, 28]
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
ctx.set_option("ring_log", 9)
for n in ns:
    a, b = syn.tables(ctx, n)
    g = mm.G(a, b)
    ctx.set_option("first_pass_vars", 3)
    out = {}
    for rep in range(2):
        for opt in (0, 1):
            ctx.set_option("first_ring", opt)
            r = mm.prove(ctx, g, syn.SEED_R)
            for _ in range(10):
                mm.prove(ctx, g, syn.SEED_R)
            ts = []
            for _ in range(40):
                t0 = time.perf_counter()
                mm.prove(ctx, g, syn.SEED_R)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            mm.prove(ctx, g, syn.SEED_R)
            log = ctx.launch_log(reset=True)
            ctx.set_option("time_kernels", 0)
            out[opt] = (r, ts[len(ts) // 2], log[0]["ms"] * 1e3, " ".join("%.1f" % (x["ms"] * 1e3) for x in log))
        same = out[0][0][0] == out[1][0][0] and np.array_equal(out[0][0][1], out[1][0][1])
        print("n=%2d same=%s  old: proof %.4f ms first pass %.1f us   ring: proof %.4f ms first pass %.1f us" % (
            n, same, out[0][1], out[0][2], out[1][1], out[1][2]), flush=True)
        print("      old  passes us:", out[0][3])
        print("      ring passes us:", out[1][3], flush=True)
    del a, b, g
