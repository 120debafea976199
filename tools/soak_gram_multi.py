"""handles of 2 / 4 / 8 entries with the matrix-core first pass on their shards (first_pass_vars = 4), sizes 16..22 in random order:
every transcript against the one-device transcript of the 27-cell schedule"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package
pkg = load_package()
mm, syn = pkg.matrix_multiplication, pkg.synthetic
F = pkg.Field(pkg.GOLDILOCKS)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = random.Random(21)
one = pkg.Context(F)
one.set_option("gram_log", 0)
ref = {}
for n in range(17, 24):
    a, b = syn.tables(one, n)
    g = mm.G(a, b)
    c1, ev, ch = mm.prove(one, g, syn.SEED_R)
    ref[n] = (c1, ev.tobytes())
    del a, b, g
handles = {}
for nd in (2, 4, 8):
    h = pkg.Context(F, devices=[0] * nd)
    h.set_option("first_pass_vars", 4)
    handles[nd] = h
tabs = {}
bad = 0
for it in range(rounds):
    nd = rng.choice([2, 4, 8])
    n = rng.randint(17, 23)
    key = (nd, n)
    if key not in tabs:
        a, b = syn.tables(handles[nd], n)
        tabs[key] = (mm.G(a, b), a, b)
    c1, ev, _ = mm.prove(handles[nd], tabs[key][0], syn.SEED_R)
    if (c1, ev.tobytes()) != ref[n]:
        bad += 1
        print("MISMATCH it", it, key, flush=True)
    if rng.random() < 0.05:
        time.sleep(rng.choice([0.0002, 0.001, 0.01]))
print("soak_gram_multi: %d proofs on handles of 2 / 4 / 8 entries, shards of 2^14 .. 2^22 entries through the matrix-core pass, mismatches: %d" % (rounds, bad))
