import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
from util import pyref
pkg = load_package()
n = 24
for mb in (256, 64, 16, 4):
    ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
    ctx.set_option("gram_log", 14)
    ctx.set_option("max_blocks", mb)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    for _ in range(20):
        pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    ctx.set_option("time_kernels", 1); ctx.launch_log()
    for _ in range(10):
        pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    log = ctx.launch_log(); per = len(log) // 10
    means = [np.mean([log[i * per + k]["ms"] for i in range(10)]) * 1e3 for k in range(per)]
    print("max_blocks=%d: %s" % (mb, " ".join("%s:%.1f" % (r["kind"], m) for r, m in zip(log[-per:], means))), flush=True)
    del a, b, g
    ctx.close()
