"""(run by hand: python tests/soak_handoff.py [proofs]) soak test of the in-kernel hand-off (finish_pass): many back-to-back proofs over instances whose
partial sums differ, every transcript compared with a precomputed oracle transcript"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
sys.path.insert(0, os.path.join(ge.ROOT, "oracle"))
import numpy as np, pyref
from oracle import Oracle
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # first_pass_vars (3: the 27-cell hand-off at every size)
o = Oracle(pkg.GOLDILOCKS)
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
ctx.set_option("first_pass_vars", first)
mm = pkg.matrix_multiplication
inst = []
for n in (14, 17, 20, 22):
    for k in range(3):
        a = pkg.DenseMultilinearExtension.generate(ctx, 1000 + 7 * n + 2 * k, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, 1001 + 7 * n + 2 * k, n)
        ch = np.array([o.challenge(pyref.SEED_R, j + 1) for j in range(n)], dtype=np.uint64)
        c1, ev = o.prover_run(o.generate(1000 + 7 * n + 2 * k, n), o.generate(1001 + 7 * n + 2 * k, n), ch)
        inst.append((mm.G(a, b), c1, ev))
t0 = time.time(); bad = 0
for it in range(iters):
    g, c1, ev = inst[(it * 7) % len(inst)]
    c, e, _ = mm.prove(ctx, g, pyref.SEED_R)
    if c != c1 or not np.array_equal(e, ev):
        bad += 1
        print("MISMATCH at iteration", it, flush=True)
print("soak: %d proofs, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
