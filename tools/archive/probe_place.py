"""Does the PLACEMENT of the buffers decide whether the first folding pass of an n = 28 proof runs in its fast (~790 us) or
slow (~880 us) mode?  Same process, same kernels:
  phase 1: one pair of input tables, several contexts (= several placements of the OUTPUT buffers)
  phase 2: one context (outputs fixed after its first proof), several pairs of input tables at different addresses
prints the device time of the two large passes for every combination"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 28


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
    return [sum(log[j + q * per]["ms"] for q in range(4)) / 4 * 1e3 for j in range(2)]


def addr(t):
    return int(pkg.load().sc_table_device_ptr(t.h) or 0)


c0 = pkg.Context(F)
a0, b0 = syn.tables(c0, n)
g0 = mm.G(a0, b0)
print("inputs at %#x / %#x (distance %d MiB)" % (addr(a0), addr(b0), (addr(b0) - addr(a0)) >> 20))
print("phase 1: same inputs, outputs of different contexts")
keep = []
for i in range(6):
    c = pkg.Context(F)
    t = passes(c, g0)
    print("   context %d: first pass %.1f us, folding pass %.1f us" % (i, t[0], t[1]), flush=True)
    keep.append(c)
    keep.append(pkg.DenseMultilinearExtension.generate(c, 1, 20 + i))      # shifts what the next context gets
print("phase 2: context 0's outputs, inputs at different addresses")
t = passes(c0, g0)
print("   inputs 0 (%#x): first pass %.1f us, folding pass %.1f us" % (addr(a0), t[0], t[1]), flush=True)
for i in range(5):
    pad = pkg.DenseMultilinearExtension.generate(c0, 2, 18 + 2 * i)
    a, b = syn.tables(c0, n)
    t = passes(c0, mm.G(a, b))
    print("   inputs %d (%#x / %#x, distance %d MiB): first pass %.1f us, folding pass %.1f us" % (i + 1, addr(a), addr(b), (addr(b) - addr(a)) >> 20, t[0], t[1]), flush=True)
    keep += [pad, a, b]
