"""Largest instances one MI355X holds: a full proof + the verifier's identities + evaluate/fix_variables properties
at n = 31, 32, 33 (2^33-entry tables: 2 x 64 GiB).  Usage: python tools/probe_big.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from conftest import load_package  # noqa: E402
from util import GOLD, oracle, pyref, verifier_identities  # noqa: E402


def main():
    ns = [int(x) for x in sys.argv[1:]] or [31, 32, 33]
    pkg = load_package()
    F = pkg.Field(GOLD)
    o = oracle(GOLD)
    for n in ns:
        ctx = pkg.Context(F)
        t0 = time.time()
        a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
        b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
        g = pkg.matrix_multiplication.G(a, b)
        ctx.synchronize() if hasattr(ctx, "synchronize") else None
        t1 = time.time()
        c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        t2 = time.time()
        final = g.evaluate([int(x) for x in ch])
        err = verifier_identities(F, c1, evals, ch, final)
        # the same proof through round 4's schedule (pass_kernel<4,2> and two-round passes: no wfold pass): bit for bit the same
        ctx.set_option("wfold_log", 0)
        t3 = time.time()
        c1b, evals_b, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        t4 = time.time()
        ctx.set_option("wfold_log", 40)
        same = c1b == c1 and (evals_b == evals).all()
        plan = " ".join("%s(%d,%d)@%d" % (st["action"], st["kf"], st["ks"], st["log_in"]) for st in pkg.schedule.plan_proof(n))
        pt = [int(o.challenge(pyref.SEED_PT, j)) for j in range(n)]
        v = a.evaluate(pt)
        ok_be = a.evaluate(pt[::-1], pkg.ORDER_BE) == v
        ok_fix = all(a.fix_variables(pt[:k]).evaluate(pt[k:]) == v for k in (1, 3, n - 1))
        hi = (1 << n) - 1
        ok_idx = all(a.evaluate([F.one if (i >> d) & 1 else F.zero for d in range(n)]) == int(o.generate_range(pyref.SEED_A, i, 1)[0])
                     for i in (0, hi, (1 << 32) + 5 if n > 32 else hi - 7, (1 << (n - 1)) + 3))
        print("n=%d generate %.2fs prove %.1f ms (first proofs of the context; wfold_log=0: %.1f ms, transcript %s) identities %s BE %s fix %s index %s | %s" %
              (n, t1 - t0, (t2 - t1) * 1e3, (t4 - t3) * 1e3, "identical" if same else "DIFFERS", "ok" if err is None else err, ok_be, ok_fix, ok_idx, plan), flush=True)
        assert same and err is None and ok_be and ok_fix and ok_idx
        del a, b, g
        ctx.close()


main()
