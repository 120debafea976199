"""timing probe for the single-table paths (BASELINE config 2 and G::new)"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
class pyref:  # seeds of the synthetic instance; tools never load oracle/
    SEED_A, SEED_B, SEED_R, SEED_PT = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003, 0xD8D8000000000004
F = pkg.Field(pkg.GOLDILOCKS)
ctx = pkg.Context(F)
for kv in sys.argv[2:]:
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
def med(fn, reps=9):
    fn(); fn()
    ts = []
    for _ in range(reps):
        ctx.synchronize(); t0 = time.perf_counter(); fn(); ctx.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts)//2]
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "24,28").split(",")]:
    t = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    import random as _r
    pt = [F.from_int(_r.Random(j).randrange(F.p)) for j in range(n)]
    for name, fn, nbytes in [
        ("evaluate LE", lambda: t.evaluate(pt), 8 * 2**n),
        ("evaluate BE (vsbw/cti)", lambda: t.evaluate(pt, order=pkg.ORDER_BE), 8 * 2**n),
        ("fix_variables k=1 LE", lambda: t.fix_variables(pt[:1]), 8 * 2**n + 4 * 2**n),
        ("fix_variables k=3 LE", lambda: t.fix_variables(pt[:3]), 8 * 2**n + 2**n),
        ("fix_variables k=%d LE" % (n // 2), lambda: t.fix_variables(pt[:n // 2]), 8 * 2**n + 8 * 2**(n - n // 2)),
        ("fix_variables k=1 BE", lambda: t.fix_variables(pt[:1], order=pkg.ORDER_BE), 8 * 2**n + 4 * 2**n),
        ("fix_variables k=%d BE" % (n // 2), lambda: t.fix_variables(pt[:n // 2], order=pkg.ORDER_BE), 8 * 2**n + 8 * 2**(n - n // 2)),
    ]:
        s = med(fn)
        print("n=%d %-28s %.1f us  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (n, name, s * 1e6, nbytes / s / 1e9, nbytes / s / 8e10), flush=True)
    del t
for p_ in (12, 13, 14):
    A = pkg.DenseMultilinearExtension.generate(ctx, 11, 2 * p_)
    B = pkg.DenseMultilinearExtension.generate(ctx, 12, 2 * p_)
    pt = [F.from_int(_r.Random(100 + j).randrange(F.p)) for j in range(2 * p_)]
    s = med(lambda: pkg.matrix_multiplication.G.new_from_tables(ctx, p_, A, B, pt), reps=5)
    nbytes = 16 * 2**(2 * p_)
    print("G::new n=%d (2 x 2^%d entries) %.1f us  %.0f GB/s algorithmic" % (p_, 2 * p_, s * 1e6, nbytes / s / 1e9), flush=True)
    del A, B
