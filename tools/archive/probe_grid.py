"""A/B of one 0/1 library option (SC_PROBE_AB, default grid_pass) at several n: transcript equality, proof time and
the launches of a proof; SC_PROBE_OPTIONS="k=v,..." sets other options first"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
ns = [int(x) for x in sys.argv[1:]] or [4, 8, 12, 13, 16, 20, 22, 24, 25, 26, 28]
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
for k, v in [kv.split("=") for kv in os.environ.get("SC_PROBE_OPTIONS", "").split(",") if kv]:
    ctx.set_option(k, int(v))
for n in ns:
    a, b = syn.tables(ctx, n)
    g = mm.G(a, b)
    out = {}
    for rep in range(2):
        for opt in (0, 1):
            ctx.set_option(os.environ.get("SC_PROBE_AB", "grid_pass"), opt)
            r = mm.prove(ctx, g, syn.SEED_R)
            for _ in range(10):
                mm.prove(ctx, g, syn.SEED_R)
            ts = []
            for _ in range(60):
                t0 = time.perf_counter()
                mm.prove(ctx, g, syn.SEED_R)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            mm.prove(ctx, g, syn.SEED_R)
            log = ctx.launch_log(reset=True)
            ctx.set_option("time_kernels", 0)
            out[opt] = (r, ts[len(ts) // 2], " ".join("%s%d,%d:%.1f" % ("g" if x["kind"] == "grid_pass" else "", x["kf"], x["ks"], x["ms"] * 1e3) for x in log))
        same = out[0][0][0] == out[1][0][0] and np.array_equal(out[0][0][1], out[1][0][1])
        print("n=%2d same=%s  off: proof %.4f ms   on: proof %.4f ms" % (n, same, out[0][1], out[1][1]), flush=True)
        if rep == 1:
            print("      off :", out[0][2])
            print("      on  :", out[1][2], flush=True)
    del a, b, g
