"""Back-to-back proofs through the matrix-core first pass (gram_pass_kernel -> gram_finish_kernel -> pass_kernel<4,2>) of several
sizes in random order, in ONE context (totals and ticket of the finish kernel cleared by the next pass kernel, partial buffers from
the pool, block caps changed on the way), interleaved with proofs on the 27-cell schedule: every transcript must equal the first
one of its size.   usage: stress_gram.py [iterations = 4000]"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
ctx.set_option("first_pass_vars", 4)
ctx.set_option("grid_log", 10)
old = pkg.Context(pkg.Field(pkg.GOLDILOCKS))      # the 27-cell schedule on the same device, its own stream
old.set_option("gram_log", 0)
gs, go = {}, {}
for n in (14, 15, 17, 20, 22, 24):
    a, b = syn.tables(ctx, n)
    gs[n] = (mm.G(a, b), a, b)
    a2, b2 = syn.tables(old, n)
    go[n] = (mm.G(a2, b2), a2, b2)
ref = {n: mm.prove(old, g[0], syn.SEED_R) for n, g in go.items()}
rng = random.Random(5)
bad = 0
for it in range(iters):
    n = rng.choice(list(gs))
    if it % 400 == 0:
        ctx.set_option("max_blocks", rng.choice([256, 64, 7, 1, 256]))
    which = rng.random()
    r = mm.prove(ctx, gs[n][0], syn.SEED_R) if which < 0.8 else mm.prove(old, go[n][0], syn.SEED_R)
    if r[0] != ref[n][0] or not np.array_equal(r[1], ref[n][1]):
        bad += 1
        print("MISMATCH at iteration", it, "n", n, flush=True)
print("stress_gram: %d proofs (80 %% through the matrix-core pass), mismatches: %d" % (iters, bad))
