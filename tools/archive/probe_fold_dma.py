"""pass_kernel<4,2>: the LDS-DMA form (option fold_dma = 1) against the register-staged pipelined form, same tables, same process:
transcripts compared at n = 16, 20 (first_pass_vars = 4, grid_log = 8: the fold runs on 2^16 / 2^20-entry tables), proof medians and
per-launch durations at n = 28."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from __graft_entry__ import load_package
from util import pyref
pkg = load_package()
for n in (16, 20, 23):
    outs = []
    for dma in (0, 1):
        ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
        for k, v in (("first_pass_vars", 4), ("grid_log", 8), ("fold_dma", dma)):
            ctx.set_option(k, v)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        outs.append(pkg.matrix_multiplication.prove(ctx, pkg.matrix_multiplication.G(a, b), pyref.SEED_R))
        ctx.close()
    ok = outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    print("n=%d: transcripts %s" % (n, "equal" if ok else "DIFFER"), flush=True)
n = 28
for rep in range(2):
    for dma in (0, 1):
        ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
        ctx.set_option("fold_dma", dma)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        for _ in range(30):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R); ts.append(time.perf_counter() - t0)
        ctx.set_option("time_kernels", 1); ctx.launch_log()
        for _ in range(10):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        log = ctx.launch_log(); per = len(log) // 10
        means = [np.mean([log[i * per + k]["ms"] for i in range(10)]) * 1e3 for k in range(per)]
        print("n=28 fold_dma=%d: proof median %.4f ms; %s" % (dma, np.median(ts) * 1e3, " ".join("%s(%d,%d)@%d:%.1f" % (r["kind"][:5], r["kf"], r["ks"], r["log_in"], m) for r, m in list(zip(log[-per:], means))[:4])), flush=True)
        del a, b, g
        ctx.close()
