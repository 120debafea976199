"""Needs the instrumented build (experiments/r03_pass_stamps.patch).  Every pass of one proof, round by round: per-block
start / loop done / block reduced stamps and the publish stamp against the launch's duration."""
import os, sys, ctypes
os.environ["SC_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
lib = pkg.load()
lib.sc_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]


def report(tag, dur_us):
    buf = np.zeros(4 * 4096 + 16, dtype=np.uint64)
    assert lib.sc_dbg_stamps(buf.ctypes.data, buf.size) == 0
    lib.sc_dbg_stamps_clear()
    blocks = buf[:4 * 4096].reshape(4096, 4)
    grid = int(np.count_nonzero(blocks[:, 0]))
    if grid == 0:
        print("%s: no stamps" % tag)
        return
    s = blocks[:grid].astype(np.int64)
    t0, t1, t2 = s[:, 0], s[:, 1], s[:, 2]
    fin = int(buf[4 * 4096])
    base = t0.min()
    us = lambda x: (x - base) / 100.0
    q = lambda v: "min %5.1f p10 %5.1f med %5.1f p90 %5.1f max %5.1f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
    hw = blocks[:grid, 3]
    h = hw.astype(np.int64) & 0xFFFFFFFF
    slot = h & 0xF
    print("%s grid %d kernel %.1f us (event); published %.1f" % (tag, grid, dur_us, us(fin)))
    print("   block start : %s" % q(us(t0)))
    print("   loop done   : %s" % q(us(t1)))
    print("   reduced     : %s" % q(us(t2)))
    print("   loop done by wave slot: " + "  ".join("s%d n=%d med %.1f" % (x, int((slot == x).sum()), np.median(us(t1)[slot == x])) for x in sorted(set(slot.tolist()))))
    sys.stdout.flush()


for n in [int(x) for x in sys.argv[1:]] or [25]:
    c = pkg.Context(F)
    a, b = syn.tables(c, n)
    g = mm.G(a, b)
    for _ in range(3):
        mm.prove(c, g, syn.SEED_R)
    for rep in range(2):
        c.set_option("time_kernels", 1)
        c.launch_log(reset=True)
        lib.sc_dbg_stamps_clear()
        h = ctypes.c_void_p()
        c.check(lib.sc_prover_create(c.h, g.f_a.h, g.f_b.h, ctypes.byref(h)))
        log = c.launch_log(reset=True)
        report("n=%d %s(%d,%d)@%d" % (n, log[0]["kind"], log[0]["kf"], log[0]["ks"], log[0]["log_in"]), log[0]["ms"] * 1e3)
        e = (ctypes.c_uint64 * 3)()
        r = F.one
        for j in range(n):
            c.check(lib.sc_prover_round(h, r, j, e))
            r = F.from_int(12345 + j)
            log = c.launch_log(reset=True)
            for p in log:
                report("n=%d %s(%d,%d)@%d" % (n, p["kind"], p["kf"], p["ks"], p["log_in"]), p["ms"] * 1e3)
        c.set_option("time_kernels", 0)
        lib.sc_prover_destroy(h)
