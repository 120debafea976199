"""Back-to-back proofs of five sizes in random order (ticket counters that rest at zero, both ticket levels, block caps
changed on the way): every transcript must equal the first one of its size."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
gs = {}
for n in (10, 16, 20, 21, 23):
    a, b = syn.tables(ctx, n)
    gs[n] = (mm.G(a, b), a, b)
ref = {n: mm.prove(ctx, g[0], syn.SEED_R) for n, g in gs.items()}
import random
rng = random.Random(3)
bad = 0
for it in range(3000):
    n = rng.choice(list(gs))
    if it % 500 == 0:
        ctx.set_option("grid_blocks", rng.choice([0, 1, 5, 33, 64]))
    r = mm.prove(ctx, gs[n][0], syn.SEED_R)
    if r[0] != ref[n][0] or not np.array_equal(r[1], ref[n][1]):
        bad += 1
        print("MISMATCH at iteration", it, "n", n, flush=True)
print("stress done, mismatches:", bad)
