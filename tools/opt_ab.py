"""same-process A/B of a context option on whole proofs: opt_ab.py <option> <value_a> <value_b> [n ...]
(alternating repeats, medians; transcripts of both settings must agree)"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_package()
mm = pkg.matrix_multiplication
SEED_A, SEED_B, SEED_R = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003
opt, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ns = [int(x) for x in sys.argv[4:]] or [24, 28]
ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
for kv in os.environ.get("SC_AB_FIXED", "").split(","):   # options held fixed during the comparison
    if kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for n in ns:
    a = pkg.DenseMultilinearExtension.generate(ctx, SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, SEED_B, n)
    g = mm.G(a, b)
    res, outs = {va: [], vb: []}, {}
    for rep in range(4):
        for v in (va, vb):
            ctx.set_option(opt, v)
            for _ in range(2):
                outs[v] = mm.prove(ctx, g, SEED_R)
            ts = []
            for _ in range(9):
                ctx.synchronize(); t0 = time.perf_counter(); mm.prove(ctx, g, SEED_R); ts.append(time.perf_counter() - t0)
            res[v].append(sorted(ts)[4] * 1e3)
    same = outs[va][0] == outs[vb][0] and np.array_equal(outs[va][1], outs[vb][1])
    print("n=%d %s=%d: %s ms | %s=%d: %s ms | transcripts %s" % (
        n, opt, va, " ".join("%.3f" % x for x in res[va]), opt, vb, " ".join("%.3f" % x for x in res[vb]),
        "identical" if same else "DIFFER"), flush=True)
    del a, b, g
