"""Python mirror of the reference crate `triangle-counting` (src/lib.rs):
g(X,Y,Z) = f(X,Y) f(Y,Z) f(X,Z) over three copies of the adjacency MLE (:22-27),
`G::new_adj_matrix` (:32-51) and the SumCheckPolynomial impl (:70-166)."""
import ctypes

import numpy as np

from ._lib import u64, voidp
from .dense_mle import DenseMultilinearExtension, _u64p, _words
from .matrix_multiplication import _round_poly_from_evals
from .sum_check_protocol import SumCheckPolynomial


class _NativeTriProver:
    """sc_tri_prover: one n^3 pass + three product-of-two-tables sumchecks"""

    def __init__(self, g):
        self.ctx, self._g = g.ctx, g
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_tri_prover_create(self.ctx.h, g.f_a_1.h, g.var_len, ctypes.byref(h)))
        self.h = h

    def c1(self):
        out = u64()
        self.ctx.check(self.ctx.lib.sc_tri_prover_c1(self.h, ctypes.byref(out)))
        return int(out.value)

    def round_evals(self, r_prev, j):
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_tri_prover_round(self.h, int(r_prev), j, e))
        return [int(x) for x in e]

    def round(self, r_prev, j):
        return _round_poly_from_evals(self.ctx, self.round_evals(r_prev, j))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.sc_tri_prover_destroy(self.h)
                self.h = None
        except Exception:
            pass


class G(SumCheckPolynomial):
    """:22-27"""

    def __init__(self, f_a_1, f_a_2, f_a_3, var_len):
        self.f_a_1, self.f_a_2, self.f_a_3, self.var_len = f_a_1, f_a_2, f_a_3, var_len
        self.ctx = f_a_1.ctx
        self.field = self.ctx.field

    @classmethod
    def new_adj_matrix(cls, ctx, num_vars, matrix):
        """:32-51 - matrix: iterable of bools, row-major"""
        F = ctx.field
        ev = np.array([F.one if b else F.zero for b in matrix], dtype=np.uint64)
        g = DenseMultilinearExtension.from_evaluations_vec(ctx, num_vars, ev)
        return cls(g, g, g, num_vars // 2)

    def clone(self):
        return G(self.f_a_1, self.f_a_2, self.f_a_3, self.var_len)

    def _h(self):
        return self.f_a_1.h, self.f_a_2.h, self.f_a_3.h

    # :53-67
    def x_vars_num(self):
        return max(self.f_a_1.num_vars() - self.var_len, 0)

    def y_vars_num(self):
        return max(self.f_a_2.num_vars() - self.var_len, 0)

    def z_vars_num(self):
        n3 = self.f_a_3.num_vars()
        return n3 if n3 < self.var_len else self.var_len

    # ---- SumCheckPolynomial (:70-166) -----------------------------------------------------
    def evaluate(self, point):
        pt = _words(point)
        if pt.size != self.num_vars():
            return None
        out = u64()
        self.ctx.check(self.ctx.lib.sc_tri_evaluate(self.ctx.h, *self._h(), self.var_len, _u64p(pt), pt.size,
                                                   ctypes.byref(out)))
        return int(out.value)

    def fix_variables(self, partial_point):
        r = _words(partial_point)
        hs = [voidp() for _ in range(3)]
        self.ctx.check(self.ctx.lib.sc_tri_fix_variables(self.ctx.h, *self._h(), self.var_len, _u64p(r), r.size,
                                                        *[ctypes.byref(h) for h in hs]))
        return G(*[DenseMultilinearExtension(self.ctx, h) for h in hs], self.var_len)

    def round_evals(self):
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_tri_round_sums(self.ctx.h, *self._h(), self.var_len, e))
        return [int(x) for x in e]

    def to_univariate(self):
        return _round_poly_from_evals(self.ctx, self.round_evals())

    def num_vars(self):
        return self.x_vars_num() + self.y_vars_num() + self.z_vars_num()

    def to_evaluations(self):
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_tri_to_evaluations(self.ctx.h, *self._h(), self.var_len, ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h).to_evaluations()

    def native_prover(self):
        """the fast engine applies to the polynomial as G::new_adj_matrix builds it"""
        if (self.f_a_1 is self.f_a_2 and self.f_a_2 is self.f_a_3 and self.var_len >= 1
                and self.f_a_1.num_vars() == 2 * self.var_len):
            return _NativeTriProver(self)
        return None


def prove(ctx, g, seed_r, draw=None):
    """sc_tri_prove: all 3 * var_len rounds of Prover<F, G> for G::new_adj_matrix in one native call.
    Returns (c_1, evals[n][3], challenges[n])."""
    from . import _lib
    n = 3 * g.var_len
    ev = np.zeros(3 * n, dtype=np.uint64)
    ch = np.zeros(n, dtype=np.uint64)
    c1 = u64()
    cb = _lib.DRAW_FN(draw) if draw is not None else ctypes.cast(None, _lib.DRAW_FN)
    ctx.check(ctx.lib.sc_tri_prove(ctx.h, g.f_a_1.h, g.var_len, cb, None, seed_r, ctypes.byref(c1), _u64p(ev), _u64p(ch)))
    return int(c1.value), ev.reshape(n, 3).copy(), ch.copy()
