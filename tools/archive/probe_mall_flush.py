"""Does the speed of the first folding pass depend on what the Infinity Cache holds when it starts?  K contexts (for both
of its modes, experiments/r03_fold_pass_two_modes.md); per context the pass is timed (a) right behind the first pass, as in
a proof, and (b) with a 2 GiB streaming read of an unrelated table in between (round-by-round API)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn, scp = pkg.matrix_multiplication, pkg.synthetic, pkg.sum_check_protocol
F = pkg.Field(pkg.GOLDILOCKS)
n, K = 28, int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = pkg.load()
import ctypes


def fold_time(ctx, g, other, pt, flush):
    ts = []
    for rep in range(4):
        ctx.set_option("time_kernels", 1)
        ctx.launch_log(reset=True)
        h = ctypes.c_void_p()
        ctx.check(lib.sc_prover_create(ctx.h, g.f_a.h, g.f_b.h, ctypes.byref(h)))
        e = (ctypes.c_uint64 * 3)()
        r = F.one
        for j in range(4):                     # rounds 0..2 come from the first pass; round 3 launches the folding pass
            if j == 3 and flush:
                other.evaluate(pt)
            ctx.check(lib.sc_prover_round(h, r, j, e))
            r = F.from_int(12345 + j)
        log = ctx.launch_log(reset=True)
        ctx.set_option("time_kernels", 0)
        lib.sc_prover_destroy(h)
        t = [x["ms"] * 1e3 for x in log if x["kind"] == "pass" and x["kf"] == 3]
        if rep:
            ts.append(t[0])
    return sum(ts) / len(ts)


for i in range(K):
    c = pkg.Context(F)
    a, b = syn.tables(c, n)
    g = mm.G(a, b)
    other = pkg.DenseMultilinearExtension.generate(c, 77, n)
    pt = [F.from_int(1000 + j) for j in range(n)]
    for _ in range(3):
        mm.prove(c, g, syn.SEED_R)
    print("context %d: folding pass behind the first pass %.1f us; with a 2 GiB read in between %.1f us" % (
        i, fold_time(c, g, other, pt, False), fold_time(c, g, other, pt, True)), flush=True)
