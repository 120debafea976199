"""Whole-proof wall time (median of 60) at the shard sizes, default schedule against variants, one process, alternating
contexts (so that a box's mood hits both alike).  usage: probe_tail.py [n ...]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
SEED_A, SEED_B, SEED_R = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003
mm = pkg.matrix_multiplication
ns = [int(x) for x in sys.argv[1:]] or [25, 28]
VARIANTS = [("default", {}), ("host_tail_log=0", {"host_tail_log": 0}), ("fold_dma=0", {"fold_dma": 0}), ("default (again)", {})]
PLAN_KEYS = ("host_tail_log", "grid_max_vars", "grid_log")
for n in ns:
    ctxs = []
    for name, o in VARIANTS:
        ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
        for k, v in o.items():
            ctx.set_option(k, v)
        a = pkg.DenseMultilinearExtension.generate(ctx, SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, SEED_B, n)
        ctxs.append((name, ctx, mm.G(a, b), []))
    ref = None
    stats = {}
    for rep in range(4):
        for name, ctx, g, ts in ctxs:
            for _ in range(3):
                out = mm.prove(ctx, g, SEED_R)
            if ref is None:
                ref = out
            assert out[0] == ref[0] and (out[1] == ref[1]).all(), name
            ctx.set_option("stat_reset", 1)
            for _ in range(15):
                ctx.synchronize()
                t0 = time.perf_counter()
                mm.prove(ctx, g, SEED_R)
                ts.append(time.perf_counter() - t0)
            stats[name] = (ctx.get_option("stat_wait_ns") / 15e3, ctx.get_option("stat_launch_ns") / 15e3)
    for name, ctx, g, ts in ctxs:
        ts.sort()
        popts = {k: v for k, v in dict(VARIANTS)[name].items() if k in PLAN_KEYS}
        print("n=%d %-18s median %.1f us  p10 %.1f  min %.1f | per proof: %.1f us waiting for kernels, %.1f us in launches | %s" % (
            n, name, ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6, ts[0] * 1e6, stats[name][0], stats[name][1],
            " ".join("%s(%d,%d)@%d" % (s["action"], s["kf"], s["ks"], s["log_in"]) for s in pkg.schedule.plan_proof(n, **popts))), flush=True)
    for name, ctx, g, ts in ctxs:
        del g
        ctx.close()
