"""Fp64<MontBackend<_, 1>> as seen from Python: field elements are *Montgomery words*
(ints in [0,p) holding x*2^64 mod p), the same words the C ABI and ark-ff use.
`from_int` / `to_int` are F::from_bigint / into_bigint (sum-check-protocol/src/lib.rs:390)."""
import ctypes

from . import _lib

GOLDILOCKS = 2**64 - 2**32 + 1
_R = 2**64


class Field:
    def __init__(self, p):
        if p < 3 or p % 2 == 0:
            raise ValueError("modulus must be an odd prime < 2^64")
        self.p = p
        self._rinv = pow(_R, -1, p)
        self.one = _R % p
        self.zero = 0
        self.two = (2 * _R) % p
        self.c = _lib.ScField(p, (-pow(p, -1, _R)) % _R, _R % p, (_R * _R) % p)

    # conversions
    def from_int(self, x):
        return (int(x) * _R) % self.p

    def to_int(self, m):
        return (int(m) * self._rinv) % self.p

    def from_ints(self, xs):
        import numpy as np
        return np.array([self.from_int(x) for x in xs], dtype=np.uint64)

    def to_ints(self, ms):
        return [self.to_int(m) for m in ms]

    # arithmetic on Montgomery words
    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def neg(self, a):
        return (-a) % self.p

    def mul(self, a, b):
        return (a * b * self._rinv) % self.p

    def inv(self, a):
        if a % self.p == 0:
            raise ZeroDivisionError("inverse of zero")
        return self.from_int(pow(self.to_int(a), -1, self.p))

    def div(self, a, b):
        return self.mul(a, self.inv(b))

    def rand(self, rng):
        """F::rand (sum-check-protocol/src/lib.rs:19) with a Python `random.Random`"""
        return self.from_int(rng.randrange(self.p))

    def ref(self):
        return ctypes.byref(self.c)
