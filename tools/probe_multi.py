"""one-device context vs a multi-device handle (all entries device 0 on this pool): median ms per proof"""
import statistics
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package

pkg = load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)


def run(ctx, n, reps):
    a, b = syn.tables(ctx, n)
    g = mm.G(a, b)
    for _ in range(3):
        mm.prove(ctx, g, syn.SEED_R)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = mm.prove(ctx, g, syn.SEED_R)
        ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts), min(ts), out


for n in [int(x) for x in (sys.argv[1:] or ["28", "25", "20", "12"])]:
    ctx = pkg.Context(F)
    med, best, ref = run(ctx, n, 30)
    print("n=%d one-device context          median %.4f ms  min %.4f" % (n, med, best), flush=True)
    ctx.close()
    for nd in (1, 2, 4, 8):
        if n < nd.bit_length() - 1:
            continue
        ctx = pkg.Context(F, devices=[0] * nd)
        med, best, out = run(ctx, n, 30)
        ok = out[0] == ref[0] and (out[1] == ref[1]).all()
        print("n=%d multi handle, %d x device 0  median %.4f ms  min %.4f  transcript %s" % (n, nd, med, best, "same" if ok else "DIFFERENT"), flush=True)
        ctx.close()
