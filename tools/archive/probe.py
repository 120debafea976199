"""quick timing probe (not the bench): whole-prover wall time and pass-kernel device time
usage: probe.py n1,n2 [vpp list] [opt=val ...]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import __graft_entry__ as ge
pkg = ge.load_package()
class pyref:  # seeds of the synthetic instance; tools never load oracle/
    SEED_A, SEED_B, SEED_R, SEED_PT = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003, 0xD8D8000000000004
mm = pkg.matrix_multiplication
ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [20, 24, 26, 28]
vpps = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2]
opts = dict(kv.split("=") for kv in sys.argv[3:])
for vpp in vpps:
    ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS))
    ctx.set_option("vars_per_pass", vpp)
    for k, v in opts.items():
        ctx.set_option(k, int(v))
    for n in ns:
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = mm.G(a, b)
        for _ in range(2):
            mm.prove(ctx, g, pyref.SEED_R)
        ts = []
        for _ in range(7):
            ctx.synchronize()
            t0 = time.perf_counter()
            mm.prove(ctx, g, pyref.SEED_R)
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[len(ts) // 2]
        ctx.set_option("time_kernels", 1)
        ctx.kernel_time(reset=True)
        mm.prove(ctx, g, pyref.SEED_R)
        nk, kms = ctx.kernel_time(reset=True)
        ctx.set_option("time_kernels", 0)
        alg = 64 * 2**n - 96
        print("%s vpp=%d n=%d wall=%.3f ms  muladds/s=%.3e  alg GB/s=%.0f  | pass kernels: %d launches %.3f ms -> alg GB/s=%.0f"
              % (opts, vpp, n, t * 1e3, (5 * 2**n - 7) / t, alg / t / 1e9, nk, kms, alg / (max(kms, 1e-9) * 1e-3) / 1e9), flush=True)
        del a, b, g
    ctx.close()
