"""Does the speed of the first folding pass follow the OUTPUT buffers?  One context, one pair of tables, one stream; after
every measurement the pool's cached output blocks are taken away (two 2^25-entry tables are created and kept: they get
exactly those blocks) together with a spacer of 2^(26 + t mod 3) entries, so the next proof folds into freshly allocated
buffers somewhere else."""
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


for ci in range(3):
    c = pkg.Context(F)
    a, b = syn.tables(c, n)
    g = mm.G(a, b)
    held, row = [], []
    for t in range(K):
        t1, t2 = passes(c, g)
        row.append("%.0f" % t2)
        held.append(pkg.DenseMultilinearExtension.generate(c, 100 + t, 25))
        held.append(pkg.DenseMultilinearExtension.generate(c, 200 + t, 25))
        held.append(pkg.DenseMultilinearExtension.generate(c, 300 + t, 26 + t % 3))
    print("context %d, same tables and stream, output placement 0..%d: fold pass us: %s" % (ci, K - 1, " ".join(row)), flush=True)
    del held, g, a, b
    c.close()
