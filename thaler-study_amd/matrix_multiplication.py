"""Python mirror of the reference crate `matrix-multiplication` (src/lib.rs):
`G` = f_A(r1, z) * f_B(z, r2) as two device tables (:12-15), `G::new` (:77-92), the
`SumCheckPolynomial` impl (:95-147) and `interpolate_quadratic_poly` (:17-60)."""
import ctypes

import numpy as np

from . import _lib
from ._lib import u64, u64p, voidp
from .dense_mle import DenseMultilinearExtension, _u64p, _words
from .sum_check_protocol import SparsePolynomial, SumCheckPolynomial


def interpolate_quadratic_poly(field, points):
    """:17-60 - generic three-point Lagrange form, like the reference"""
    polys = []
    for i in range(3):
        (xi, yi), (xj, _), (xk, _) = points[i], points[(i + 1) % 3], points[(i + 2) % 3]
        den = field.mul(field.sub(xi, xj), field.sub(xi, xk))
        coeffs = [(0, field.mul(xj, xk)), (1, field.sub(field.neg(xj), xk)), (2, field.one)]
        coeffs = [(d, field.div(field.mul(c, yi), den)) for d, c in coeffs]
        polys.append(SparsePolynomial.from_coefficients_vec(field, coeffs))
    return polys[0] + polys[1] + polys[2]


def _round_poly_from_evals(ctx, e):
    """the three sums -> the round polynomial as triangle_counting::G and W hand it out: coefficients through
    sc_interpolate_quadratic, then `DensePolynomial -> SparsePolynomial` (`p.into()`,
    triangle-counting/src/lib.rs:129-131, gkr-protocol/src/round_polynomial.rs:87-89): non-zero terms only"""
    ev = (u64 * 3)(*[int(x) for x in e])
    c = (u64 * 3)()
    ctx.check(ctx.lib.sc_interpolate_quadratic(ctx.field.ref(), ev, c))
    return SparsePolynomial.from_dense(ctx.field, [int(c[d]) for d in range(3)])


def _round_poly_lagrange(ctx, e):
    """the three sums -> the round polynomial as matrix_multiplication::G hands it out (:124-130): the sum of three
    Lagrange terms, which in arkworks' canonical form can carry an explicit zero constant term (e.g. H(0) = H(2) = 0)
    - the same values as sc_interpolate_quadratic's, and on the wire the reference's bytes"""
    f = ctx.field
    poly = interpolate_quadratic_poly(f, [(f.zero, int(e[0])), (f.one, int(e[1])), (f.two, int(e[2]))])
    return poly


class _NativeProver:
    """sc_prover: the fused fold + round-sum engine behind Prover::round"""

    def __init__(self, g):
        self.ctx = g.ctx
        self._g = g  # keeps the borrowed tables alive
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_prover_create(self.ctx.h, g.f_a.h, g.f_b.h, ctypes.byref(h)))
        self.h = h
        self.last_evals = None

    def c1(self):
        out = u64()
        self.ctx.check(self.ctx.lib.sc_prover_c1(self.h, ctypes.byref(out)))
        return int(out.value)

    def round_evals(self, r_prev, j):
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_prover_round(self.h, int(r_prev), j, e))
        self.last_evals = [int(x) for x in e]
        return self.last_evals

    def round(self, r_prev, j):
        return _round_poly_lagrange(self.ctx, self.round_evals(r_prev, j))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.sc_prover_destroy(self.h)
                self.h = None
        except Exception:
            pass


class G(SumCheckPolynomial):
    """:12-15"""

    def __init__(self, f_a, f_b):
        self.f_a, self.f_b = f_a, f_b
        self.ctx = f_a.ctx
        self.field = self.ctx.field

    @classmethod
    def new(cls, ctx, n, a, b, point):
        """:77-92 - a, b: the 2^n x 2^n matrices flattened row-major (Montgomery words)"""
        A = DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * n, a)
        B = DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * n, b)
        pt = _words(point)
        if pt.size != 2 * n:
            raise ValueError("point must have 2n entries")
        ha, hb = voidp(), voidp()
        ctx.check(ctx.lib.sc_matmul_g_new(ctx.h, A.h, B.h, n, _u64p(pt), ctypes.byref(ha), ctypes.byref(hb)))
        f_a, f_b = DenseMultilinearExtension(ctx, ha), DenseMultilinearExtension(ctx, hb)
        assert f_a.num_vars() == n and f_b.num_vars() == n                 # :88-89
        return cls(f_a, f_b)

    @classmethod
    def new_from_tables(cls, ctx, n, A, B, point):
        """G::new on matrices that already live in HBM (2^(2n)-entry device tables)"""
        pt = _words(point)
        ha, hb = voidp(), voidp()
        ctx.check(ctx.lib.sc_matmul_g_new(ctx.h, A.h, B.h, n, _u64p(pt), ctypes.byref(ha), ctypes.byref(hb)))
        return cls(DenseMultilinearExtension(ctx, ha), DenseMultilinearExtension(ctx, hb))

    def clone(self):
        """#[derive(Clone)] :11 - tables are never written by the prover, so share them"""
        return G(self.f_a, self.f_b)

    # ---- SumCheckPolynomial (:95-147) ---------------------------------------------------
    def evaluate(self, point):
        pt = _words(point)
        if pt.size != self.num_vars():
            return None
        out = u64()
        self.ctx.check(self.ctx.lib.sc_prod2_evaluate(self.ctx.h, self.f_a.h, self.f_b.h, _u64p(pt), pt.size,
                                                     ctypes.byref(out)))
        return int(out.value)

    def fix_variables(self, partial_point):
        return G(self.f_a.fix_variables(partial_point), self.f_b.fix_variables(partial_point))

    def round_evals(self):
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_prod2_round_sums(self.ctx.h, self.f_a.h, self.f_b.h, e))
        return [int(x) for x in e]

    def to_univariate(self):
        return _round_poly_lagrange(self.ctx, self.round_evals())

    def fold_and_univariate(self, r):
        """fix_variables(&[r]) + to_univariate in one pass over HBM"""
        rr = (u64 * 1)(int(r))
        ha, hb = voidp(), voidp()
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_prod2_fold_and_sums(self.ctx.h, self.f_a.h, self.f_b.h, rr,
                                                          ctypes.byref(ha), ctypes.byref(hb), e))
        g = G(DenseMultilinearExtension(self.ctx, ha), DenseMultilinearExtension(self.ctx, hb))
        return g, _round_poly_lagrange(self.ctx, [int(x) for x in e])

    def num_vars(self):
        return self.f_a.num_vars() + self._log_world()

    def to_evaluations(self):
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_prod2_to_evaluations(self.ctx.h, self.f_a.h, self.f_b.h, ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h).to_evaluations()

    # ---- provided-method overrides: keep Prover::new / round on the device ---------------
    def hypercube_sum(self, field=None):
        out = u64()
        self.ctx.check(self.ctx.lib.sc_prod2_sum(self.ctx.h, self.f_a.h, self.f_b.h, ctypes.byref(out)))
        return int(out.value)

    def native_prover(self):
        return _NativeProver(self)

    def _log_world(self):
        _, world = self.ctx.rank_world()
        return world.bit_length() - 1


def prove(ctx, g, seed_r, draw=None):
    """sc_prove: the whole loop of benches/mm_benchmark.rs:88-96 in one native call.
    Returns (c_1, evals[n][3], challenges[n])."""
    buf = getattr(g, "_prove_buf", None)
    if buf is None:  # output buffers and their ctypes views are built once per G (a benchmark loop
        n = g.num_vars()  # proves the same G many times; this keeps the wrapper out of its timing)
        ev = np.zeros(3 * max(n, 1), dtype=np.uint64)
        ch = np.zeros(max(n, 1), dtype=np.uint64)
        c1 = u64()
        buf = g._prove_buf = (n, ev, ch, c1, _u64p(ev), _u64p(ch), ctypes.byref(c1), ctypes.cast(None, _lib.DRAW_FN))
    n, ev, ch, c1, p_ev, p_ch, p_c1, no_draw = buf
    cb = _lib.DRAW_FN(draw) if draw is not None else no_draw
    ctx.check(ctx.lib.sc_prove(ctx.h, g.f_a.h, g.f_b.h, cb, None, seed_r, p_c1, p_ev, p_ch))
    return int(c1.value), ev[: 3 * n].reshape(n, 3).copy(), ch[:n].copy()
