"""A handle's host tail, serial against on the launcher threads (option host_par_min 0 / 256): whole proofs at sizes where almost
nothing but the host tail is left (n = 12 .. 16 over 8 / 4 / 2 entries of device 0), same handle and tables, alternating.
usage: probe_host_tail.py [n ...]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
SEED_A, SEED_B, SEED_R = 0xA5A5000000000001, 0xB6B6000000000002, 0xC7C7000000000003
mm = pkg.matrix_multiplication
ns = [int(x) for x in sys.argv[1:]] or [12, 14, 16, 20, 25]
for n_dev in (8, 4, 2):
    ctx = pkg.Context(pkg.Field(pkg.GOLDILOCKS), devices=[0] * n_dev)
    for n in ns:
        a = pkg.DenseMultilinearExtension.generate(ctx, SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, SEED_B, n)
        g = mm.G(a, b)
        ref = None
        t = {0: [], 256: []}
        for _ in range(20):
            mm.prove(ctx, g, SEED_R)
        for rep in range(12):
            for v in (0, 256):
                ctx.set_option("host_par_min", v)
                mm.prove(ctx, g, SEED_R)
                for _ in range(20):
                    t0 = time.perf_counter()
                    out = mm.prove(ctx, g, SEED_R)
                    t[v].append((time.perf_counter() - t0) * 1e6)
                sig = (out[0], out[1].tobytes())
                ref = ref or sig
                assert sig == ref, "transcripts differ"
        plan = pkg.schedule.plan_proof(n, n_dev, "local")
        print("devices %d n=%d  serial %.1f us  threads %.1f us   (plan ends %s)" % (
            n_dev, n, statistics.median(t[0]), statistics.median(t[256]), [(s["action"], s["kf"], s["ks"], s["log_in"]) for s in plan][-1]), flush=True)
        del g, a, b
    ctx.close()
