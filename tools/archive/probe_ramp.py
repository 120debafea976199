"""does the device reach its steady clocks inside a short bench run?  n = 28 proofs back to back from a cold start: wall time per
proof and the HIP-event durations of the two large passes, in groups of 10 proofs; then after idle gaps of 0.05 / 0.5 / 3 s."""
import statistics
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from __graft_entry__ import load_package

pkg = load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
a, b = syn.tables(ctx, n)
g = mm.G(a, b)
ctx.set_option("time_kernels", 1)


def burst(count, label):
    ctx.launch_log(reset=True)
    ts = []
    for _ in range(count):
        t0 = time.perf_counter()
        mm.prove(ctx, g, syn.SEED_R)
        ts.append((time.perf_counter() - t0) * 1e3)
    log = ctx.launch_log(reset=True)
    first = [r["ms"] * 1e3 for r in log if r["kind"] == "pass" and r["kf"] == 0]
    fold = [r["ms"] * 1e3 for r in log if r["kind"] == "pass" and r["kf"] == 3]
    for i in range(0, count, 10):
        print("%-18s proofs %3d-%3d: %.4f ms/proof  first pass %6.1f us  fold pass %6.1f us" % (
            label, i, i + 9, statistics.median(ts[i:i + 10]), statistics.median(first[i:i + 10]), statistics.median(fold[i:i + 10])), flush=True)


burst(int(sys.argv[2]) if len(sys.argv) > 2 else 200, "cold start")
for gap in (0.05, 0.5, 3.0):
    time.sleep(gap)
    burst(40, "after %.2f s idle" % gap)
