"""soak of the multi-device handle's runtime (launcher threads, mailboxes, tail slots, parking): thousands of proofs of random
sizes on handles of 2 / 4 / 8 entries, interleaved with idle gaps around the threads' parking threshold, table calls and prover
churn; every transcript compared with the one-device transcript of the same instance"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from __graft_entry__ import load_package

pkg = load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = random.Random(12)
one = pkg.Context(F)
ref = {}
for n in range(3, 21):
    a, b = syn.tables(one, n)
    g = mm.G(a, b)
    c1, ev, ch = mm.prove(one, g, syn.SEED_R)
    ref[n] = (c1, ev.tobytes(), g.evaluate([int(x) for x in ch]), [int(x) for x in ch])
handles = {nd: pkg.Context(F, devices=[0] * nd) for nd in (2, 4, 8)}
tabs = {}
t0 = time.time()
bad = 0
for it in range(rounds):
    nd = rng.choice((2, 4, 8))
    n = rng.randrange(max(3, nd.bit_length() - 1), 21 if it % 50 else 19)
    ctx = handles[nd]
    key = (nd, n)
    if key not in tabs or rng.random() < 0.05:
        tabs[key] = syn.tables(ctx, n)
    a, b = tabs[key]
    g = mm.G(a, b)
    mode = rng.random()
    if mode < 0.7:
        c1, ev, ch = mm.prove(ctx, g, syn.SEED_R)
        ok = (c1, ev.tobytes()) == ref[n][:2]
    elif mode < 0.85:
        pr = g.native_prover()
        ok = pr.c1() == ref[n][0]
        evs = []
        for j in range(n):
            evs.append(pr.round_evals(ref[n][3][j - 1] if j else 1, j))
            if rng.random() < 0.1:
                time.sleep(rng.choice((0.0001, 0.0004, 0.002)))      # around the launcher threads' parking threshold
        ok = ok and np.array(evs, dtype=np.uint64).tobytes() == ref[n][1]
    else:
        ok = g.evaluate(ref[n][3]) == ref[n][2]
    if not ok:
        bad += 1
        print("MISMATCH at iteration %d: %d devices, n = %d, mode %.2f" % (it, nd, n, mode), flush=True)
    if rng.random() < 0.02:
        time.sleep(rng.choice((0.0002, 0.001, 0.01)))
print("%d iterations, %d mismatches, %.1f s" % (rounds, bad, time.time() - t0))
sys.exit(1 if bad else 0)
