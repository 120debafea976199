"""fix_variables of <= 3 variables on tables below the block-per-CU threshold - four-wave blocks drawing runs
of 4 tiles (round 3: one wave per block works) against one tile per draw (now).  HIP-event time per launch."""
import statistics
import sys

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from __graft_entry__ import load_package

pkg = load_package()
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
ctx.set_option("time_kernels", 1)
print("%-4s %-3s %12s %12s" % ("n", "k", "grab=4 (r03)", "grab=1 (now)"))
for n in (16, 18, 19, 20, 21, 22):
    t = pkg.DenseMultilinearExtension.generate(ctx, 1, n)
    for k in (1, 2, 3):
        row = []
        for grab in (4, 1):
            ctx.set_option("dbg_fold_grab", grab)
            for _ in range(5):
                t.fix_variables([3] * k)
            ctx.synchronize()
            ctx.launch_log(reset=True)
            for _ in range(40):
                t.fix_variables([3] * k)
            ctx.synchronize()
            log = ctx.launch_log(reset=True)
            row.append(statistics.median(r["ms"] for r in log) * 1e3)
        print("%-4d %-3d %10.2f us %10.2f us" % (n, k, row[0], row[1]), flush=True)
