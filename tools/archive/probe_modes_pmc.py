"""K identical contexts in one process, two proofs each at n = 28 (after warm-up): run under
   rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/probe_modes_pmc.py K
and compare the counters of the fold pass's launches between the contexts that run it at ~800 us and those at ~880 us
(experiments/r03_fold_pass_two_modes.md).  Prints the event-timed duration of every context's fold pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = 28
for i in range(K):
    c = pkg.Context(F)
    a, b = syn.tables(c, n)
    g = mm.G(a, b)
    for _ in range(2):
        mm.prove(c, g, syn.SEED_R)
    c.set_option("time_kernels", 1)
    c.launch_log(reset=True)
    for _ in range(2):
        mm.prove(c, g, syn.SEED_R)
    log = c.launch_log(reset=True)
    t = [x["ms"] * 1e3 for x in log if x["kind"] == "pass" and x["kf"] == 3]
    t0 = [x["ms"] * 1e3 for x in log if x["kind"] == "pass" and x["kf"] == 0]
    print("context %d: first pass %s us, fold pass %s us" % (i, " ".join("%.0f" % x for x in t0), " ".join("%.0f" % x for x in t)), flush=True)
