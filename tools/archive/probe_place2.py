"""placement probe 2: inputs fixed; every new context first allocates (and keeps) a pad of `pad_log` entries so that its
OUTPUT buffers land further and further from the inputs; prints the two large passes' device times and the addresses"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
n = 28
pad_log = int(sys.argv[1]) if len(sys.argv) > 1 else 29      # 4 GiB pads
lib = pkg.load()


def passes(ctx, g):
    for _ in range(3):
        mm.prove(ctx, g, syn.SEED_R)
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    for _ in range(4):
        mm.prove(ctx, g, syn.SEED_R)
    log = ctx.launch_log(reset=True)
    ctx.set_option("time_kernels", 0)
    per = len(log) // 4
    return [sum(log[j + q * per]["ms"] for q in range(4)) / 4 * 1e3 for j in range(3)]


def addr(t):
    return int(lib.sc_table_device_ptr(t.h) or 0)


c0 = pkg.Context(F)
a0, b0 = syn.tables(c0, n)
g0 = mm.G(a0, b0)
print("inputs at %#x / %#x" % (addr(a0), addr(b0)))
keep = []
for i in range(8):
    c = pkg.Context(F)
    pad = pkg.DenseMultilinearExtension.generate(c, 1, pad_log) if i else None
    t = passes(c, g0)
    probe = pkg.DenseMultilinearExtension.generate(c, 3, 10)     # where does this context allocate now?
    print("   context %d (pad at %s, small block at %#x): first %.1f us, folding %.1f us, third %.1f us" % (
        i, ("%#x" % addr(pad)) if pad else "-", addr(probe), t[0], t[1], t[2]), flush=True)
    keep += [c, pad, probe]
