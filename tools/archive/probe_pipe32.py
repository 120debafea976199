"""pass_kernel<3,2>: the pipelined whole-tile form (option pipe32 = 1) against the staged one, same tables, same process:
transcripts compared, proof medians, per-launch durations.   usage: probe_pipe32.py [n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from __graft_entry__ import load_package
from util import pyref
pkg = load_package()
for n in [int(x) for x in sys.argv[1:]] or [25, 26, 27, 24, 22]:
    ref = None
    for pipe in (0, 1, 0, 1):
        ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
        ctx.set_option("pipe32", pipe)
        ctx.set_option("gram_log", 0)
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        for _ in range(30):
            out = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        if ref is None:
            ref = out
        assert out[0] == ref[0] and np.array_equal(out[1], ref[1]), "transcripts differ"
        ts = []
        for _ in range(40):
            t0 = time.perf_counter(); pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R); ts.append(time.perf_counter() - t0)
        ctx.set_option("time_kernels", 1); ctx.launch_log()
        for _ in range(10):
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        log = ctx.launch_log(); per = len(log) // 10
        means = [np.mean([log[i * per + k]["ms"] for i in range(10)]) * 1e3 for k in range(per)]
        print("n=%d pipe32=%d: proof median %.4f ms; %s" % (n, pipe, np.median(ts) * 1e3,
              " ".join("%s(%d,%d)@%d:%.1f" % (r["kind"], r["kf"], r["ks"], r["log_in"], m) for r, m in zip(log[-per:], means))), flush=True)
        del a, b, g
        ctx.close()
