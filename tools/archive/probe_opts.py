"""alternate option sets on one box: tools/probe_opts.py 24,25 "reverse_log=0" "reverse_log=26" ...
prints, per n and option set, the median proof time of 3 x 60 proofs (sets interleaved) and the launches of one proof"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
ns = [int(x) for x in sys.argv[1].split(",")]
sets = sys.argv[2:] or [""]
F = pkg.Field(pkg.GOLDILOCKS)
ctxs = []
for s in sets:
    c = pkg.Context(F)
    for kv in [x for x in s.split(",") if x]:
        k, v = kv.split("=")
        c.set_option(k, int(v))
    ctxs.append(c)
for n in ns:
    tabs = [syn.tables(c, n) for c in ctxs]
    gs = [mm.G(a, b) for a, b in tabs]
    ref = None
    meds = [[] for _ in sets]
    for rep in range(3):
        for i, c in enumerate(ctxs):
            r = mm.prove(c, gs[i], syn.SEED_R)
            if ref is None:
                ref = r
            assert r[0] == ref[0] and np.array_equal(r[1], ref[1]), "transcripts differ"
            for _ in range(10):
                mm.prove(c, gs[i], syn.SEED_R)
            ts = []
            for _ in range(60):
                t0 = time.perf_counter()
                mm.prove(c, gs[i], syn.SEED_R)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            meds[i].append(ts[len(ts) // 2])
    for i, c in enumerate(ctxs):
        c.set_option("time_kernels", 1)
        c.launch_log(reset=True)
        for _ in range(4):
            mm.prove(c, gs[i], syn.SEED_R)
        log = c.launch_log(reset=True)
        c.set_option("time_kernels", 0)
        per = len(log) // 4
        avg = [sum(log[j + q * per]["ms"] for q in range(4)) / 4 for j in range(per)]
        desc = " ".join("%s%d,%d:%.1f" % ("g" if log[j]["kind"] == "grid_pass" else "", log[j]["kf"], log[j]["ks"], avg[j] * 1e3) for j in range(per))
        lib = pkg.load()
        where = "a@%#x b@%#x" % (int(lib.sc_table_device_ptr(tabs[i][0].h) or 0), int(lib.sc_table_device_ptr(tabs[i][1].h) or 0))
        print("n=%2d %-40s proof %s ms | %s | %s" % (n, sets[i] or "(default)", " ".join("%.4f" % m for m in meds[i]), desc, where), flush=True)
    del gs, tabs
