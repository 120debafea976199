"""pass_kernel<4,2> (the fold behind the matrix-core first pass) in its three forms - LDS-DMA (default, Goldilocks), register
staging with two sets (fold_dma=0), ONE staging set at three waves per SIMD (fold_oneset=1) - on the SAME context and tables,
alternating (the pass has two speeds per context: experiments/r03_fold_pass_two_modes.md; only a same-context comparison isolates
the kernel).  HIP-event durations from the launch log.  usage: probe_fold42.py [n ...]"""
import sys, os, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
SEED_A, SEED_B, SEED_R = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003
mm = pkg.matrix_multiplication
ns = [int(x) for x in sys.argv[1:]] or [28, 25]
# (the third form, fold_oneset=1, is experiments/r05_fold_oneset.patch: apply it and add ("one set x3 waves", {"fold_dma": 0, "fold_oneset": 1}))
FORMS = [("dma", {"fold_dma": 1}), ("two sets", {"fold_dma": 0})]
field = os.environ.get("SC_PROBE_FIELD", "gold")
p = pkg.GOLDILOCKS if field == "gold" else 2**64 - 59
for n in ns:
    for c in range(3):          # three contexts: the pass's two placement modes show up between contexts
        ctx = pkg.Context(pkg.Field(p))
        a = pkg.DenseMultilinearExtension.generate(ctx, SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, SEED_B, n)
        g = mm.G(a, b)
        ref = None
        for _ in range(30):
            mm.prove(ctx, g, SEED_R)      # clocks
        us = {name: [] for name, _ in FORMS}
        tot = {name: [] for name, _ in FORMS}
        for rep in range(8):
            for name, o in FORMS:
                for k, v in o.items():
                    ctx.set_option(k, v)
                mm.prove(ctx, g, SEED_R)
                ctx.set_option("time_kernels", 1)
                ctx.launch_log(reset=True)
                out = mm.prove(ctx, g, SEED_R)
                log = ctx.launch_log(reset=True)
                ctx.set_option("time_kernels", 0)
                if ref is None:
                    ref = out
                assert out[0] == ref[0] and (out[1] == ref[1]).all(), name
                us[name] += [r["ms"] * 1e3 for r in log if r["kind"] == "pass" and r["kf"] == 4]
                tot[name].append(sum(r["ms"] for r in log) * 1e3)
        print("n=%d %s context %d: " % (n, field, c) + " | ".join("%s %.1f us (min %.1f; proof kernels %.1f)" % (
            name, statistics.median(us[name]), min(us[name]), statistics.median(tot[name])) for name, _ in FORMS), flush=True)
        del g, a, b
        ctx.close()
