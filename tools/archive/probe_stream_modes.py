"""Does the speed of the first folding pass follow the context's STREAM (hardware queue) rather than its buffers?
One context, one pair of tables, the same pool buffers throughout; the pass is timed, then the context moves to a new HIP
stream (the old ones are kept alive so that every new stream is another queue) and it is timed again.  Needs the
dbg_renew_stream option of the experiment build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
n = 28
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def passes(c, g):
    for _ in range(2):
        mm.prove(c, g, syn.SEED_R)
    c.set_option("time_kernels", 1)
    c.launch_log(reset=True)
    for _ in range(3):
        mm.prove(c, g, syn.SEED_R)
    log = c.launch_log(reset=True)
    c.set_option("time_kernels", 0)
    per = len(log) // 3
    return sum(log[q * per]["ms"] for q in range(3)) / 3 * 1e3, sum(log[1 + q * per]["ms"] for q in range(3)) / 3 * 1e3


for ci in range(2):
    c = pkg.Context(F)
    a, b = syn.tables(c, n)
    g = mm.G(a, b)
    row = []
    for s in range(K):
        t1, t2 = passes(c, g)
        row.append("%.0f/%.0f" % (t1, t2))
        c.set_option("dbg_renew_stream", 1)
    print("context %d, same tables and pool, stream 0..%d: first/fold pass us: %s" % (ci, K - 1, " ".join(row)), flush=True)
