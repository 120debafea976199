import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref  # noqa: E402
from oracle import Oracle  # noqa: E402

GOLD = pyref.GOLDILOCKS
TOY_MODULI = [5, 389, 1572869]


def load_golden(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def pid(p):
    return "gold" if p == GOLD else "p%d" % p


_oracles = {}


def oracle(p):
    if p not in _oracles:
        _oracles[p] = Oracle(p)
    return _oracles[p]


def challenges(o, n, seed=pyref.SEED_R):
    return np.array([o.challenge(seed, j + 1) for j in range(n)], dtype=np.uint64)


def verifier_identities(F, c1, evals, ch, final_eval):
    """the sumcheck verifier's checks on a transcript of (H(0), H(1), H(2)) triples
    (sum-check-protocol/src/lib.rs:286, :316-318, :303); returns None or the first failure"""
    inv2 = F.inv(F.two)
    claim = c1
    for j in range(len(evals)):
        e0, e1, e2 = (int(x) for x in evals[j])
        if F.add(e0, e1) != claim:
            return "round %d: g_j(0)+g_j(1) != previous claim" % j
        r = int(ch[j])
        l0 = F.mul(F.mul(F.sub(r, F.one), F.sub(r, F.two)), inv2)
        l1 = F.neg(F.mul(r, F.sub(r, F.two)))
        l2 = F.mul(F.mul(r, F.sub(r, F.one)), inv2)
        claim = F.add(F.add(F.mul(l0, e0), F.mul(l1, e1)), F.mul(l2, e2))
    if claim != final_eval:
        return "g_n(r_n) != g(r)"
    return None
