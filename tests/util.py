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
