"""Needs the instrumented build (experiments/r03_pass_stamps.patch + stamps in evaluate_kernel).  evaluate at n: per block
start -> eq table built -> stream done; last block's publish."""
import os, sys, ctypes
os.environ["SC_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import numpy as np
pkg = ge.load_package()
F = pkg.Field(pkg.GOLDILOCKS)
lib = pkg.load()
lib.sc_dbg_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for n in [int(x) for x in sys.argv[1:]] or [24]:
    c = pkg.Context(F)
    t = pkg.DenseMultilinearExtension.generate(c, 5, n)
    pt = [F.from_int(1000 + j) for j in range(n)]
    for _ in range(5):
        t.evaluate(pt)
    for rep in range(3):
        c.set_option("time_kernels", 1)
        c.launch_log(reset=True)
        lib.sc_dbg_stamps_clear()
        t.evaluate(pt)
        log = c.launch_log(reset=True)
        c.set_option("time_kernels", 0)
        buf = np.zeros(4 * 4096 + 16, dtype=np.uint64)
        assert lib.sc_dbg_stamps(buf.ctypes.data, buf.size) == 0
        b = buf[:4 * 4096].reshape(4096, 4)
        g = int(np.count_nonzero(b[:, 0]))
        s = b[:g].astype(np.int64)
        base = s[:, 0].min()
        us = lambda x: (x - base) / 100.0
        q = lambda v: "min %5.1f med %5.1f max %5.1f" % (v.min(), np.median(v), v.max())
        x = buf[4 * 4096:4 * 4096 + 5].astype(np.int64)
        print("n=%d evaluate grid %d kernel %s us" % (n, g, ["%.1f" % (l["ms"] * 1e3) for l in log]))
        print("   start %s | eq built %s | stream done %s" % (q(us(s[:, 0])), q(us(s[:, 1])), q(us(s[:, 2]))))
        print("   last block: ticket %.1f fence %.1f published %.1f" % (us(x[1]), us(x[2]), us(x[0])), flush=True)
