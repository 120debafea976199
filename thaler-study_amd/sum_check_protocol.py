"""Python mirror of the reference crate `sum-check-protocol` (src/lib.rs): same names,
argument meaning and error behaviour, so protocol code and tests read like the Rust.

  RngF                :13-21      BooleanHypercube   :34-70
  Prover              :73-117     SumCheckPolynomial :121-156
  Verifier            :227-331    VerifierRoundResult:246-253     Error :24-31

Field elements are Montgomery words (see field.py).  Polynomial types that are backed by
device tables (matrix_multiplication.G) override `hypercube_sum` and `native_prover` so that
`Prover.new` / `Prover.round` run as fused GPU passes; any other implementor of the trait
goes through the generic fix_variables -> to_univariate path exactly like the reference.
"""
from abc import ABC, abstractmethod


class Error(Exception):
    """sum-check-protocol/src/lib.rs:24-31"""


class ProverClaimMismatch(Error):
    def __init__(self, a, b):
        super().__init__("prover claim mismatches evaluation %s %s" % (a, b))
        self.claimed, self.evaluated = a, b


class NoPolySet(Error):
    def __init__(self):
        super().__init__("verifier has no oracle access to the polynomial")


class RngF(ABC):
    """:13-15"""

    @abstractmethod
    def draw(self):
        ...


class FieldRng(RngF):
    """the blanket `impl<F: Field, T: Rng> RngF<F> for T` (:17-21) for a random.Random"""

    def __init__(self, field, rng):
        self.field, self.rng = field, rng

    def draw(self):
        return self.field.rand(self.rng)


class BooleanHypercube:
    """:34-70 - iterates {0,1}^n as lists of field elements, index bit 0 first"""

    def __init__(self, field, n):
        self.field, self.n, self.current = field, n, 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.current == 2 ** self.n:
            raise StopIteration
        v = self.current
        self.current += 1
        return [self.field.one if (v >> i) & 1 else self.field.zero for i in range(self.n)]


class SparsePolynomial:
    """ark_poly::univariate::SparsePolynomial<F>: (degree, coeff) terms sorted by degree, in arkworks'
    canonical form - which is what `serialize_uncompressed` puts on the wire (fiat-shamir/src/lib.rs:45-61):

      from_coefficients_vec  pops the TRAILING zero terms of the vector as given, then sorts; zero terms
                             elsewhere stay
      a + b                  if a is zero: b as is; if b is zero: a as is; else a merge in which a term of
                             both is dropped when the sum is zero and a term of one is copied as is
      from_dense             (`From<DensePolynomial>`) keeps the non-zero coefficients only"""

    def __init__(self, field, coeffs):
        self.field = field
        self.coeffs = [(int(d), int(c)) for d, c in coeffs]

    @classmethod
    def from_coefficients_vec(cls, field, coeffs):
        coeffs = [(int(d), int(c)) for d, c in coeffs]
        while coeffs and coeffs[-1][1] == 0:
            coeffs.pop()
        coeffs.sort(key=lambda t: t[0])
        assert not coeffs or coeffs[-1][1] != 0, "from_coefficients_vec: the highest term is zero (arkworks panics here)"
        return cls(field, coeffs)

    @classmethod
    def from_dense(cls, field, dense):
        return cls(field, [(d, int(c)) for d, c in enumerate(dense) if int(c) != 0])

    def is_zero(self):
        return all(c == 0 for _, c in self.coeffs)

    def evaluate(self, x):
        f = self.field
        acc = 0
        for d, c in self.coeffs:
            term = c
            for _ in range(d):
                term = f.mul(term, x)
            acc = f.add(acc, term)
        return acc

    def degree(self):
        return self.coeffs[-1][0] if self.coeffs else 0

    def __add__(self, other):
        if self.is_zero():
            return SparsePolynomial(self.field, other.coeffs)
        if other.is_zero():
            return SparsePolynomial(self.field, self.coeffs)
        a, b, out, i, k = self.coeffs, other.coeffs, [], 0, 0
        while i < len(a) and k < len(b):
            if a[i][0] < b[k][0]:
                out.append(a[i])
                i += 1
            elif a[i][0] > b[k][0]:
                out.append(b[k])
                k += 1
            else:
                c = self.field.add(a[i][1], b[k][1])
                if c != 0:
                    out.append((a[i][0], c))
                i, k = i + 1, k + 1
        return SparsePolynomial(self.field, out + a[i:] + b[k:])

    def __eq__(self, other):
        return isinstance(other, SparsePolynomial) and self.coeffs == other.coeffs

    def __repr__(self):
        return "SparsePolynomial(%r)" % (self.coeffs,)


class SumCheckPolynomial(ABC):
    """:121-156"""

    @abstractmethod
    def evaluate(self, point):
        """None if the dimensionality of `point` does not match"""

    @abstractmethod
    def fix_variables(self, partial_point):
        ...

    @abstractmethod
    def to_univariate(self):
        ...

    @abstractmethod
    def num_vars(self):
        ...

    @abstractmethod
    def to_evaluations(self):
        ...

    # provided methods (additive; callers unchanged - SURVEY.md section 8b "Ownership")
    def hypercube_sum(self, field):
        """sum of to_evaluations(): what Prover::new computes at :89"""
        acc = 0
        for v in self.to_evaluations():
            acc = field.add(acc, int(v))
        return acc

    def native_prover(self):
        """a device-side prover engine, or None for the generic path"""
        return None


class Prover:
    """:73-117"""

    def __init__(self, g, field=None):
        self.g = g
        self.field = field if field is not None else g.field
        self._engine = g.native_prover()
        if self._engine is not None:
            self.c_1_value = self._engine.c1()
        else:
            self.c_1_value = g.hypercube_sum(self.field)       # :89
        self._num_vars = g.num_vars()
        self.r = []

    @classmethod
    def new(cls, g, field=None):
        return cls(g, field)

    def c_1(self):
        return self.c_1_value

    def round(self, r_prev, j):
        """:105-112 - r_prev is ignored at j == 0"""
        if self._engine is not None:
            if j != 0:
                self.r.append(r_prev)
            return self._engine.round(r_prev, j)
        if j != 0:
            self.r.append(r_prev)
            self.g = self.g.fix_variables([r_prev])
        return self.g.to_univariate()

    def num_vars(self):
        return self._num_vars


class VerifierRoundResult:
    """:246-253"""

    def __init__(self, kind, value):
        self.kind, self.value = kind, value

    @classmethod
    def JthRound(cls, r):
        return cls("JthRound", r)

    @classmethod
    def FinalRound(cls, ok):
        return cls("FinalRound", ok)

    def is_final(self):
        return self.kind == "FinalRound"

    def __repr__(self):
        return "%s(%r)" % (self.kind, self.value)


class Verifier:
    """:227-331.

    strict=True (default) adds, in the final round, the check the reference omits: g_n(0) + g_n(1)
    must equal g_{n-1}(r_{n-1}) (the reference's last branch :298-310 only tests g_n(r_n) == g(r), so
    a prover that is self-consistent on a false c_1 up to round n-1 and then sends the honest g_n is
    accepted), and reports a failed oracle check as FinalRound(False) instead of panicking.
    strict=False is the reference bit for bit: no such check, and a failed `assert_eq!` (:303) raises."""

    def __init__(self, n, g, field=None, strict=True):
        self.n = n
        self.g = g
        self.field = field if field is not None else g.field
        self.strict = strict
        self.c_1 = 0
        self.g_part = []
        self.r = []

    @classmethod
    def new(cls, n, g, field=None, strict=True):
        return cls(n, g, field, strict)

    def set_c_1(self, c_1):
        self.c_1 = c_1

    def round(self, g_j, rng):
        f = self.field
        r_j = rng.draw()                                                   # :283
        if not self.r:                                                     # :284-297
            evaluation = f.add(g_j.evaluate(f.zero), g_j.evaluate(f.one))
            if self.c_1 != evaluation:
                raise ProverClaimMismatch("start %d" % f.to_int(self.c_1), "%d" % f.to_int(evaluation))
            self.g_part.append(g_j)
            self.r.append(r_j)
            return VerifierRoundResult.JthRound(r_j)
        if len(self.r) == self.n - 1:                                      # :298-310
            if self.strict:
                prev_evaluation = self.g_part[-1].evaluate(self.r[-1])
                evaluation = f.add(g_j.evaluate(f.zero), g_j.evaluate(f.one))
                if prev_evaluation != evaluation:
                    raise ProverClaimMismatch("%d" % f.to_int(prev_evaluation), "%d" % f.to_int(evaluation))
            self.r.append(r_j)
            if self.g is None:
                raise NoPolySet()
            lhs = g_j.evaluate(r_j)
            rhs = self.g.evaluate(self.r)
            if lhs != rhs and not self.strict:                                                   # :303
                raise AssertionError("assert_eq!(g_j.evaluate(&r_j), g.evaluate(&self.r).unwrap())")
            return VerifierRoundResult.FinalRound(lhs == rhs)
        prev_evaluation = self.g_part[-1].evaluate(self.r[-1])             # :313-328
        evaluation = f.add(g_j.evaluate(f.zero), g_j.evaluate(f.one))
        if prev_evaluation != evaluation:
            raise ProverClaimMismatch("%d" % f.to_int(prev_evaluation), "%d" % f.to_int(evaluation))
        self.g_part.append(g_j)
        self.r.append(r_j)
        return VerifierRoundResult.JthRound(r_j)
