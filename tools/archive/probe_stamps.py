"""Needs the instrumented build: `git apply experiments/r03_pass_stamps.patch && make -C thaler-study_amd/csrc` (the product
library carries no stamps; results of round 3: profiles/r03_pass_block_stamps.txt).  Where a streaming pass's time goes: per-block start / end-of-stream / end-of-block-reduce
stamps (100 MHz wall clock), the block's XCC and CU, and the last block's publish stamp, against the launch's own duration."""
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
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)


def report(tag, dur_us, save=None):
    buf = np.zeros(4 * 4096 + 16, dtype=np.uint64)
    assert lib.sc_dbg_stamps(buf.ctypes.data, buf.size) == 0
    lib.sc_dbg_stamps_clear()
    blocks = buf[:4 * 4096].reshape(4096, 4)
    grid = int(np.count_nonzero(blocks[:, 0]))
    s = blocks[:grid].astype(np.int64)
    t0, t1, t2, hw = s[:, 0], s[:, 1], s[:, 2], blocks[:grid, 3]
    fin = int(buf[4 * 4096])
    base = t0.min()
    us = lambda x: (x - base) / 100.0
    q = lambda v: "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
    print("%s grid %d kernel %.1f us (event)" % (tag, grid, dur_us))
    print("   block start      : %s" % q(us(t0)))
    print("   stream loop done : %s" % q(us(t1)))
    print("   block reduced    : %s" % q(us(t2)))
    print("   published        : %.1f   (last block reduced -> published %.1f us)" % (us(fin), us(fin) - us(t2).max()))
    x = buf[4 * 4096 + 1:4 * 4096 + 5].astype(np.int64)
    if x.all():
        print("   last block       : ticket %.1f  fence done %.1f  partials summed %.1f  values stored %.1f  seq stored %.1f" % (us(x[0]), us(x[1]), us(x[2]), us(x[3]), us(fin)))
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
    hwid = hw.astype(np.int64) & 0xFFFFFFFF
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
    print("   loop done by XCC : " + "  ".join("x%d n=%d med %.0f" % (x, int((xcc == x).sum()), np.median(us(t1)[xcc == x])) for x in sorted(set(xcc.tolist()))))
    print("   blockIdx%%8 == XCC for %d of %d blocks" % (int((np.arange(grid) % 8 == xcc).sum()), grid))
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    per_cu = {}
    for k, v in zip(key.tolist(), us(t1).tolist()):
        per_cu.setdefault(k, []).append(v)
    cnt = np.bincount([len(v) for v in per_cu.values()])
    print("   CUs used %d; blocks per CU histogram %s" % (len(per_cu), cnt.tolist()))
    for nb in sorted(set(len(v) for v in per_cu.values())):
        vals = [x for v in per_cu.values() if len(v) == nb for x in v]
        print("      CUs holding %d block(s): loop done med %.0f (min %.0f max %.0f)" % (nb, np.median(vals), min(vals), max(vals)))
    sys.stdout.flush()
    if save:
        np.save(os.path.join(ROOT, "gpurun_out", save), buf)


for n in [int(x) for x in sys.argv[1:]] or [25, 28]:
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
        report("n=%d first pass (0,3)" % n, log[0]["ms"] * 1e3, "stamps_n%d_first_%d.npy" % (n, rep))
        e = (ctypes.c_uint64 * 3)()
        r = F.one
        for j in range(4):
            c.check(lib.sc_prover_round(h, r, j, e))
            r = F.from_int(12345 + j)
        log = c.launch_log(reset=True)
        p = [x for x in log if x["kind"] == "pass"]
        if p:
            report("n=%d fold pass (%d,%d)" % (n, p[0]["kf"], p[0]["ks"]), p[0]["ms"] * 1e3, "stamps_n%d_fold_%d.npy" % (n, rep))
        c.set_option("time_kernels", 0)
        lib.sc_prover_destroy(h)
