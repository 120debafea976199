import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
from util import pyref
pkg = load_package()
GOLD = pkg.GOLDILOCKS
for n in [int(x) for x in sys.argv[1:]] or (28, 27, 26, 25):
    for gram_log in (14, 0):
        ctx = pkg.Context(pkg.Field(GOLD))
        ctx.set_option("gram_log", gram_log)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        for _ in range(30):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R); ts.append(time.perf_counter() - t0)
        ctx.set_option("time_kernels", 1)
        ctx.launch_log()
        for _ in range(10):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        log = ctx.launch_log()
        per = len(log) // 10
        last = log[-per:]
        # mean per position
        means = [np.mean([log[i * per + k]["ms"] for i in range(10)]) * 1e3 for k in range(per)]
        print("n=%d gram_log=%d: proof median %.4f ms, mean %.4f ms; launches: %s" % (n, gram_log, np.median(ts) * 1e3, np.mean(ts) * 1e3,
              " ".join("%s(%d,%d)@%d:%.1f" % (r["kind"], r["kf"], r["ks"], r["log_in"], m) for r, m in zip(last, means))), flush=True)
        del a, b, g
        ctx.close()
