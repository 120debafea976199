"""placement probe 3: K contexts, each with its own input tables (at its own addresses) and - after its first proof - its
own output buffers.  Prove every (inputs of context i) x (outputs of context j) combination and print the device time of
the first folding pass: do the rows (inputs) or the columns (outputs) decide its fast / slow mode?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
n, K = 28, int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctxs = [pkg.Context(F) for _ in range(K)]
tabs = [syn.tables(c, n) for c in ctxs]
gs = [mm.G(a, b) for a, b in tabs]
for c, g in zip(ctxs, gs):          # every context allocates its outputs with its own inputs
    mm.prove(c, g, syn.SEED_R)


def fold_pass(ctx, g):
    mm.prove(ctx, g, syn.SEED_R)
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    for _ in range(3):
        mm.prove(ctx, g, syn.SEED_R)
    log = ctx.launch_log(reset=True)
    ctx.set_option("time_kernels", 0)
    per = len(log) // 3
    return sum(log[1 + q * per]["ms"] for q in range(3)) / 3 * 1e3, sum(log[q * per]["ms"] for q in range(3)) / 3 * 1e3


print("rows: inputs of context i; columns: outputs (pool) of context j; first folding pass, us   [first pass, us]")
for i in range(K):
    row, firsts = [], []
    for j in range(K):
        t2, t1 = fold_pass(ctxs[j], gs[i])
        row.append(t2)
        firsts.append(t1)
    print("inputs %d: " % i + " ".join("%6.1f" % x for x in row) + "   [" + " ".join("%5.0f" % x for x in firsts) + "]", flush=True)
