"""Looped hand-over stress: thousands of back-to-back proofs of alternating instances on ONE context, all reusing
the prover's one pinned tail slot, with the five-round passes capped at block counts above the XCD count (9, 12, 17, 40, 257 ...)
so that the blocks that store the handed-over tables sit on several XCDs - a store that is still in some XCD's L2 when the host
reads the slot (the race WgOut::host_out describes) shows as a transcript of the PREVIOUS instance.  Every transcript is compared
with the oracle's.  python tools/stress_handover.py [proofs]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
sys.path.insert(0, os.path.join(ge.ROOT, "oracle"))
import numpy as np  # noqa: E402
import pyref  # noqa: E402
from oracle import Oracle  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
o = Oracle(pkg.GOLDILOCKS)
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
mm = pkg.matrix_multiplication
inst = []
for n in (12, 13, 16, 17, 19, 21):        # hand-overs of 2^7 .. 2^12-entry tables, from passes of 1 .. 512 blocks
    for k in range(3):
        sa, sb = 3000 + 11 * n + 2 * k, 3001 + 11 * n + 2 * k
        a = pkg.DenseMultilinearExtension.generate(ctx, sa, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, sb, n)
        ch = np.array([o.challenge(pyref.SEED_R, j + 1) for j in range(n)], dtype=np.uint64)
        c1, ev = o.prover_run_mt(o.generate(sa, n), o.generate(sb, n), ch)
        inst.append((n, mm.G(a, b), c1, ev))
caps = [0, 9, 12, 17, 40, 257, 1024]
t0, bad = time.time(), 0
for it in range(iters):
    if it % 50 == 0:
        ctx.set_option("grid_blocks", caps[(it // 50) % len(caps)])
    n, g, c1, ev = inst[(it * 7) % len(inst)]
    c, e, _ = mm.prove(ctx, g, pyref.SEED_R)
    if c != c1 or not np.array_equal(e, ev):
        bad += 1
        print("MISMATCH at proof %d (n = %d, grid_blocks = %d)" % (it, n, ctx.get_option("grid_blocks")), flush=True)
print("stress_handover: %d proofs, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
sys.exit(1 if bad else 0)
