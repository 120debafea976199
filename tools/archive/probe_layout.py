"""One context, one proof size (n = 28), ONE process per call: does the allocation history decide the mode of the first
folding pass?  variant: plain | spacer4 | spacer16 | outputs_first | small_first | tables_last
   plain          tables, then the proofs (bench.py's order)
   spacer4 / 16   a 4 / 16 GiB table allocated (and kept) between the tables and the first proof
   outputs_first  a proof over other tables first (its output buffers go back to the pool), then these tables
   small_first    proofs at n = 20, 24, 26 first (their buffers stay cached in the pool)
   tables_last    a 16 GiB table allocated first and kept, then as plain"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
variant = sys.argv[1] if len(sys.argv) > 1 else "plain"
n = 28
c = pkg.Context(F)
keep = []
if variant == "tables_last":
    keep.append(pkg.DenseMultilinearExtension.generate(c, 9, 31))
if variant == "outputs_first":
    a0, b0 = syn.tables(c, n)
    mm.prove(c, mm.G(a0, b0), syn.SEED_R)
    del a0, b0
if variant == "small_first":
    for m in (20, 24, 26):
        am, bm = syn.tables(c, m)
        mm.prove(c, mm.G(am, bm), syn.SEED_R)
        keep += [am, bm]
a, b = syn.tables(c, n)
if variant == "spacer4":
    keep.append(pkg.DenseMultilinearExtension.generate(c, 9, 29))
if variant == "spacer16":
    keep.append(pkg.DenseMultilinearExtension.generate(c, 9, 31))
g = mm.G(a, b)
for _ in range(5):
    mm.prove(c, g, syn.SEED_R)
c.set_option("time_kernels", 1)
c.launch_log(reset=True)
for _ in range(5):
    mm.prove(c, g, syn.SEED_R)
log = c.launch_log(reset=True)
per = len(log) // 5
t1 = sum(log[q * per]["ms"] for q in range(5)) / 5 * 1e3
t2 = sum(log[1 + q * per]["ms"] for q in range(5)) / 5 * 1e3
print("%-14s first pass %.0f us, fold pass %.0f us" % (variant, t1, t2), flush=True)
