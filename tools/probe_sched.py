"""Two schedules on the SAME context and tables, alternating: per-launch HIP-event durations of one proof each way (medians of 8).
usage: probe_sched.py <option> <v1> <v2> [n ...]      e.g. probe_sched.py fold_rounds 3 2 25 28"""
import sys, os, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
SEED_A, SEED_B, SEED_R = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003
mm = pkg.matrix_multiplication
opt, v1, v2 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ns = [int(x) for x in sys.argv[4:]] or [25, 28]
for n in ns:
    ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
    a = pkg.DenseMultilinearExtension.generate(ctx, SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, SEED_B, n)
    g = mm.G(a, b)
    for _ in range(30):
        mm.prove(ctx, g, SEED_R)
    rows = {v1: [], v2: []}
    wall = {v1: [], v2: []}
    import time
    for rep in range(8):
        for v in (v1, v2):
            ctx.set_option(opt, v)
            mm.prove(ctx, g, SEED_R)
            for _ in range(5):
                ctx.synchronize()
                t0 = time.perf_counter()
                mm.prove(ctx, g, SEED_R)
                wall[v].append((time.perf_counter() - t0) * 1e6)
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            mm.prove(ctx, g, SEED_R)
            log = ctx.launch_log(reset=True)
            ctx.set_option("time_kernels", 0)
            rows[v].append([(r["kind"], r["kf"], r["ks"], r["log_in"], r["ms"] * 1e3) for r in log])
    for v in (v1, v2):
        k = len(rows[v][0])
        med = [statistics.median(rep[i][4] for rep in rows[v]) for i in range(k)]
        print("n=%d %s=%d: wall %.1f us | kernels %.1f us: " % (n, opt, v, statistics.median(wall[v]), sum(med)) +
              "  ".join("%s(%d,%d)@%d %.1f" % (rows[v][0][i][:4] + (med[i],)) for i in range(k)), flush=True)
    del g, a, b
    ctx.close()
