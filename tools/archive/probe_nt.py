import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
from util import pyref
pkg = load_package()
n = 28
for rep in range(2):
    for opts in ({}, {"nt_store_log": 24}, {"nt_store_log": 30}, {"nt_load_log": 40}):
        ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
        for k, v in opts.items():
            ctx.set_option(k, v)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        for _ in range(30):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R); ts.append(time.perf_counter() - t0)
        ctx.set_option("time_kernels", 1); ctx.launch_log()
        for _ in range(10):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        log = ctx.launch_log(); per = len(log) // 10
        means = [np.mean([log[i * per + k]["ms"] for i in range(10)]) * 1e3 for k in range(per)]
        print("%-22s proof median %.4f ms; %s" % (opts, np.median(ts) * 1e3, " ".join("%s(%d,%d)@%d:%.1f" % (r["kind"][:5], r["kf"], r["ks"], r["log_in"], m) for r, m in list(zip(log[-per:], means))[:5])), flush=True)
        del a, b, g
        ctx.close()
