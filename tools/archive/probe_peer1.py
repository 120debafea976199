"""The sharded code path on ONE rank (peer transport attached with world = 1: every sharded pass exchanges with itself):
n = 25 proofs with and without a transport, with five-round sharded passes (grid_sharded) and without - what the
exchange and the sharded schedule cost per pass, minus the fabric."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn, D = pkg.matrix_multiplication, pkg.synthetic, pkg.distributed
F = pkg.Field(pkg.GOLDILOCKS)
for mode in ("none", "peer"):
    ctx = pkg.Context(F)
    if mode == "peer":
        D.attach_peer(ctx, 0, 1)
    n = 25
    a, b = syn.tables(ctx, n)
    g = mm.G(a, b)
    for gs in (1, 0):
        ctx.set_option("grid_sharded", gs)
        for _ in range(10): mm.prove(ctx, g, syn.SEED_R)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter(); mm.prove(ctx, g, syn.SEED_R); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        ctx.set_option("time_kernels", 1); ctx.launch_log(reset=True)
        mm.prove(ctx, g, syn.SEED_R)
        log = ctx.launch_log(reset=True); ctx.set_option("time_kernels", 0)
        print(mode, "grid_sharded", gs, "proof %.4f ms" % ts[len(ts)//2], " ".join("%s%d,%d:%.1f" % ("g" if r["kind"] == "grid_pass" else "", r["kf"], r["ks"], r["ms"] * 1e3) for r in log), flush=True)
    del a, b, g
    ctx.close()
