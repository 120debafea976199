"""resident kernel vs per-pass launches: same transcript, timing.  usage: probe_resident.py [n ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
ns = [int(x) for x in sys.argv[1:]] or [6, 9, 12, 14, 17, 20, 22, 24, 26, 28]
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
for n in ns:
    a, b = syn.tables(ctx, n)
    g = mm.G(a, b)
    out = {}
    for res in (0, 1):
        ctx.set_option("resident", res)
        r = mm.prove(ctx, g, syn.SEED_R)
        for _ in range(20):
            mm.prove(ctx, g, syn.SEED_R)
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            mm.prove(ctx, g, syn.SEED_R)
        dt = (time.perf_counter() - t0) / reps * 1e3
        out[res] = (r, dt)
    same = out[0][0][0] == out[1][0][0] and np.array_equal(out[0][0][1], out[1][0][1])
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    mm.prove(ctx, g, syn.SEED_R)
    log = ctx.launch_log(reset=True)
    ctx.set_option("time_kernels", 0)
    print("n=%2d same=%s  launches %.4f ms   resident %.4f ms   (%d launches: %s)" % (
        n, same, out[0][1], out[1][1], len(log), " ".join("%s%d,%d@%d:%.0fus" % (r["kind"][:1], r["kf"], r["ks"], r["log_in"], r["ms"] * 1e3) for r in log)), flush=True)
    del a, b, g
# per-phase stamps of block 0 (10 ns ticks): wait-enter, cmd seen, body start, body end, drained, published
for n in ns[-3:]:
    a, b = syn.tables(ctx, n)
    g = mm.G(a, b)
    ctx.set_option("resident", 1)
    ctx.set_option("resident_stamps", 1)
    for _ in range(3):
        mm.prove(ctx, g, syn.SEED_R)
    ctx.synchronize()
    st = [ctx.get_option("resident_stamp_%d" % i) for i in range(128)]
    t0 = st[0]
    print("n=%d stamps (us since phase 0 entry): [enter, cmd, body, body_end, drained, published]" % n)
    for p in range(16):
        row = st[p * 8:p * 8 + 6]
        if row[0] == 0:
            break
        print("  phase %2d: %s" % (p, " ".join("%8.2f" % ((x - t0) / 100.0) for x in row)))
    ctx.set_option("resident_stamps", 0)
    del a, b, g
print("host think time per command: %.2f us" % (ctx.get_option("resident_host_ns") / 1e3 / max(1, 3 * sum(1 for _ in range(1)))))
