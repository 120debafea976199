"""field.hpp (the header the HIP kernels include) compiled for the host with g++ and
checked against Python big integers.  No GPU needed."""
import ctypes
import os
import random
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = 2**64 - 2**32 + 1
R = 2**64
u64p = ctypes.POINTER(ctypes.c_uint64)


@pytest.fixture(scope="module")
def fh(tmp_path_factory):
    out = tmp_path_factory.mktemp("fh") / "libfield_host.so"
    src = os.path.join(ROOT, "tests", "cpp", "field_host_harness.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out), src])
    lib = ctypes.CDLL(str(out))
    lib.fh_gold_binop.argtypes = [ctypes.c_int, u64p, u64p, u64p, ctypes.c_size_t]
    lib.fh_gen_binop.argtypes = [ctypes.c_uint64, ctypes.c_int, u64p, u64p, u64p, ctypes.c_size_t]
    lib.fh_gold_dot.argtypes = [u64p, u64p, ctypes.c_size_t]
    lib.fh_gold_dot.restype = ctypes.c_uint64
    lib.fh_gold_dot3.argtypes = [u64p, u64p, ctypes.c_size_t]
    lib.fh_gold_dot3.restype = ctypes.c_uint64
    lib.fh_gen_dot.argtypes = [ctypes.c_uint64, u64p, u64p, ctypes.c_size_t]
    lib.fh_gen_dot.restype = ctypes.c_uint64
    for name in ("fh_gen_dot3", "fh_gen_dot_split"):
        getattr(lib, name).argtypes = [ctypes.c_uint64, u64p, u64p, ctypes.c_size_t]
        getattr(lib, name).restype = ctypes.c_uint64
    lib.fh_gold_dot_split.argtypes = [u64p, u64p, ctypes.c_size_t]
    lib.fh_gold_dot_split.restype = ctypes.c_uint64
    lib.fh_params.argtypes = [ctypes.c_uint64, u64p]
    lib.fh_splitmix64.argtypes = [ctypes.c_uint64]
    lib.fh_splitmix64.restype = ctypes.c_uint64
    return lib


def _arr(xs):
    return np.array(xs, dtype=np.uint64)


def _p(a):
    return a.ctypes.data_as(u64p)


def _edge_values(p):
    vals = {0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2}
    for s in (31, 32, 33, 63):
        for d in (-1, 0, 1):
            v = (1 << s) + d
            if 0 <= v < p:
                vals.add(v)
    vals |= {v for v in (0xFFFFFFFF, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001, 0x100000000) if v < p}
    return sorted(vals)


def _pairs(p, n_random, seed):
    rng = random.Random(seed)
    e = _edge_values(p)
    pairs = [(x, y) for x in e for y in e]
    pairs += [(rng.randrange(p), rng.randrange(p)) for _ in range(n_random)]
    return _arr([x for x, _ in pairs]), _arr([y for _, y in pairs])


def _check_ops(call, p):
    a, b = _pairs(p, 20000, p & 0xFFFF)
    out = np.empty_like(a)
    rinv = pow(R, -1, p)
    ai = [int(x) for x in a]
    bi = [int(x) for x in b]
    expect = {
        0: [(x + y) % p for x, y in zip(ai, bi)],
        1: [(x - y) % p for x, y in zip(ai, bi)],
        2: [(x * y * rinv) % p for x, y in zip(ai, bi)],
        3: [(2 * x) % p for x in ai],
        4: [(x * R) % p for x in ai],
        5: [(x * rinv) % p for x in ai],
        6: [((x << 64 | y) * rinv) % p for x, y in zip(ai, bi)],   # redc(hi=a, lo=b), hi < p
    }
    for op, exp in expect.items():
        call(op, _p(a), _p(b), _p(out), a.size)
        got = [int(x) for x in out]
        assert got == exp, "op %d mismatch for p=%d" % (op, p)


def test_goldilocks_ops(fh):
    _check_ops(lambda *args: fh.fh_gold_binop(*args), GOLD)


@pytest.mark.parametrize("p", [5, 389, 1572869, GOLD, 2**64 - 59, 2**63 + 29, 2**61 - 1, 3])
def test_generic_ops(fh, p):
    _check_ops(lambda *args: fh.fh_gen_binop(p, *args), p)


def test_field_params(fh):
    for p in (5, 389, 1572869, GOLD, 2**64 - 59):
        out = np.empty(4, dtype=np.uint64)
        fh.fh_params(p, _p(out))
        assert int(out[0]) == p
        assert (int(out[1]) * p) % R == R - 1
        assert int(out[2]) == R % p
        assert int(out[3]) == (R * R) % p
    assert (R % GOLD) == 0xFFFFFFFF and (R * R) % GOLD == 0xFFFFFFFE00000001


def test_lazy_accumulator(fh):
    rng = random.Random(7)
    rinv = pow(R, -1, GOLD)
    for n in (0, 1, 2, 3, 1000, 70000):
        a = _arr([rng.randrange(GOLD) for _ in range(n)])
        b = _arr([rng.randrange(GOLD) for _ in range(n)])
        exp = sum(int(x) * int(y) for x, y in zip(a, b)) * rinv % GOLD
        assert int(fh.fh_gold_dot(_p(a), _p(b), n)) == exp
        assert int(fh.fh_gold_dot3(_p(a), _p(b), n)) == exp
    # worst case: every product is (p-1)^2, enough terms to carry into the third word
    n = 300000
    a = np.full(n, GOLD - 1, dtype=np.uint64)
    exp = n * (GOLD - 1) ** 2 * rinv % GOLD
    assert int(fh.fh_gold_dot(_p(a), _p(a), n)) == exp
    assert int(fh.fh_gold_dot3(_p(a), _p(a), n)) == exp
    assert int(fh.fh_gold_dot_split(_p(a), _p(a), n)) == exp
    # the generic modulus (the reference's Fp64<MontBackend<T,1>> with any T): 160-bit lazy sums, reduced once
    for p in (3, 5, 389, 1572869, 2**61 - 1, 2**63 + 29, 2**64 - 59, GOLD):
        for n in (0, 1, 2, 7, 5000):
            a = _arr([rng.randrange(p) for _ in range(n)])
            b = _arr([rng.randrange(p) for _ in range(n)])
            exp = sum(int(x) * int(y) for x, y in zip(a, b)) * pow(R, -1, p) % p
            assert int(fh.fh_gen_dot(p, _p(a), _p(b), a.size)) == exp
            assert int(fh.fh_gen_dot3(p, _p(a), _p(b), a.size)) == exp
            assert int(fh.fh_gen_dot_split(p, _p(a), _p(b), a.size)) == exp
        # worst case: every product is (p-1)^2 and there are enough of them to carry well into the fifth limb / the counters
        n = 300000
        a = np.full(n, p - 1, dtype=np.uint64)
        exp = n * (p - 1) ** 2 * pow(R, -1, p) % p
        assert int(fh.fh_gen_dot(p, _p(a), _p(a), n)) == exp
        assert int(fh.fh_gen_dot3(p, _p(a), _p(a), n)) == exp
        assert int(fh.fh_gen_dot_split(p, _p(a), _p(a), n)) == exp


def test_splitmix_matches_pyref(fh):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyref
    for x in (0, 1, 0xA5A5000000000001, 2**64 - 1, 123456789):
        assert int(fh.fh_splitmix64(x)) == pyref.splitmix64(x)
