"""Python mirror of the GKR pieces that sit on the sumcheck hot path:

  circuit.rs:7-213      GateType, Gate, CircuitLayer, Circuit (evaluate, num_vars_at, add_i, mul_i)
  round_polynomial.rs   W = add_i(r_i,b,c)(W(b)+W(c)) + mul_i(r_i,b,c) W(b) W(c)  (:23-119)
  lib.rs:373-436        Prover::start_round's construction of W for layer i (`start_round_w`)

The GKR message state machines (gkr-protocol/src/lib.rs Prover/Verifier) are control plane and
stay on the host side of the reference; they drive `sum_check_protocol.Prover` on a `W` exactly
as they do today."""
import ctypes

import numpy as np

from . import _lib
from ._lib import u64, voidp
from .dense_mle import DenseMultilinearExtension, _u64p, _words
from .matrix_multiplication import _round_poly_from_evals
from .sum_check_protocol import SumCheckPolynomial


class GateType:
    Add = "add"
    Mul = "mul"


class Gate:
    """circuit.rs:17-31"""

    def __init__(self, ttype, inputs):
        self.ttype, self.inputs = ttype, list(inputs)


class CircuitLayer:
    """circuit.rs:34-52"""

    def __init__(self, layer):
        self.layer = list(layer)

    def __len__(self):
        return len(self.layer)


class Circuit:
    """circuit.rs:72-213: layers[0] is the output layer"""

    def __init__(self, layers, num_inputs):
        self.layers, self.num_inputs = list(layers), num_inputs

    def num_vars_at(self, layer):
        if layer < len(self.layers):
            n = len(self.layers[layer])
        elif layer == len(self.layers):
            n = self.num_inputs
        else:
            return None
        return (n & -n).bit_length() - 1            # trailing_zeros

    def evaluate(self, field, inputs):
        """:99-124 -> per-layer values (Montgomery words), outputs first"""
        layers = [list(inputs)]
        cur = layers[0]
        for layer in reversed(self.layers):
            cur = [field.add(cur[g.inputs[0]], cur[g.inputs[1]]) if g.ttype == GateType.Add
                   else field.mul(cur[g.inputs[0]], cur[g.inputs[1]]) for g in layer.layer]
            layers.append(cur)
        layers.reverse()
        return layers

    def add_i(self, i, a, b, c):
        g = self.layers[i].layer[a]
        return g.ttype == GateType.Add and g.inputs[0] == b and g.inputs[1] == c

    def mul_i(self, i, a, b, c):
        g = self.layers[i].layer[a]
        return g.ttype == GateType.Mul and g.inputs[0] == b and g.inputs[1] == c


class _NativeWProver:
    def __init__(self, w):
        self.ctx, self._w = w.ctx, w
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_gkr_prover_create(self.ctx.h, w.add_i.h, w.mul_i.h, w.w_b.h, w.w_c.h,
                                                        ctypes.byref(h)))
        self.h = h

    def c1(self):
        out = u64()
        self.ctx.check(self.ctx.lib.sc_gkr_prover_c1(self.h, ctypes.byref(out)))
        return int(out.value)

    def round_evals(self, r_prev, j):
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_gkr_prover_round(self.h, int(r_prev), j, e))
        return [int(x) for x in e]

    def round(self, r_prev, j):
        return _round_poly_from_evals(self.ctx, self.round_evals(r_prev, j))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.sc_gkr_prover_destroy(self.h)
                self.h = None
        except Exception:
            pass


class W(SumCheckPolynomial):
    """round_polynomial.rs:23-44"""

    def __init__(self, add_i, mul_i, w_b, w_c):
        self.add_i, self.mul_i, self.w_b, self.w_c = add_i, mul_i, w_b, w_c
        self.ctx = add_i.ctx
        self.field = self.ctx.field

    @classmethod
    def new(cls, add_i, mul_i, w_b, w_c):
        return cls(add_i, mul_i, w_b, w_c)

    def clone(self):
        return W(self.add_i, self.mul_i, self.w_b, self.w_c)

    def _h(self):
        return self.add_i.h, self.mul_i.h, self.w_b.h, self.w_c.h

    # ---- SumCheckPolynomial (:47-119) ---------------------------------------------------
    def evaluate(self, point):
        pt = _words(point)
        if pt.size != self.num_vars():
            return None
        out = u64()
        self.ctx.check(self.ctx.lib.sc_gkr_w_evaluate(self.ctx.h, *self._h(), _u64p(pt), pt.size, ctypes.byref(out)))
        return int(out.value)

    def fix_variables(self, partial_point):
        r = _words(partial_point)
        hs = [voidp() for _ in range(4)]
        self.ctx.check(self.ctx.lib.sc_gkr_w_fix_variables(self.ctx.h, *self._h(), _u64p(r), r.size,
                                                          *[ctypes.byref(h) for h in hs]))
        return W(*[DenseMultilinearExtension(self.ctx, h) for h in hs])

    def round_evals(self):
        e = (u64 * 3)()
        self.ctx.check(self.ctx.lib.sc_gkr_w_round_sums(self.ctx.h, *self._h(), e))
        return [int(x) for x in e]

    def to_univariate(self):
        return _round_poly_from_evals(self.ctx, self.round_evals())

    def num_vars(self):
        return self.add_i.num_vars()

    def to_evaluations(self):
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_gkr_w_to_evaluations(self.ctx.h, *self._h(), ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h).to_evaluations()

    def native_prover(self):
        return _NativeWProver(self)


def wiring(ctx, circuit, i, r_i):
    """add_i(r_i,.,.) and mul_i(r_i,.,.) as device tables (gkr-protocol/src/lib.rs:388-416)"""
    layer = circuit.layers[i].layer
    k_i, k_next = circuit.num_vars_at(i), circuit.num_vars_at(i + 1)
    gt = (ctypes.c_int32 * len(layer))(*[0 if g.ttype == GateType.Add else 1 for g in layer])
    i0 = (ctypes.c_uint32 * len(layer))(*[g.inputs[0] for g in layer])
    i1 = (ctypes.c_uint32 * len(layer))(*[g.inputs[1] for g in layer])
    r = _words(r_i)
    assert r.size == k_i
    ha, hm = voidp(), voidp()
    ctx.check(ctx.lib.sc_gkr_wiring(ctx.h, gt, i0, i1, k_i, k_next, _u64p(r), ctypes.byref(ha), ctypes.byref(hm)))
    return DenseMultilinearExtension(ctx, ha), DenseMultilinearExtension(ctx, hm)


class SparseLayerProver(_NativeWProver):
    """SumCheckProver<F, W<F>> for layer i built straight from the gate list: every round visits the
    2^k_i gates instead of the 4^k_{i+1} entries of the dense predicate tables (sc_gkr_prover_create_sparse)"""

    def __init__(self, ctx, circuit, evaluation, i, r_i):
        self.ctx = ctx
        layer = circuit.layers[i].layer
        k_i, k_next = circuit.num_vars_at(i), circuit.num_vars_at(i + 1)
        gt = (ctypes.c_int32 * len(layer))(*[0 if g.ttype == GateType.Add else 1 for g in layer])
        i0 = (ctypes.c_uint32 * len(layer))(*[g.inputs[0] for g in layer])
        i1 = (ctypes.c_uint32 * len(layer))(*[g.inputs[1] for g in layer])
        r = _words(r_i)
        assert r.size == k_i
        self._w_next = DenseMultilinearExtension.from_evaluations_vec(ctx, k_next, np.array(evaluation[i + 1], dtype=np.uint64))
        self._num_vars = 2 * k_next
        h = voidp()
        ctx.check(ctx.lib.sc_gkr_prover_create_sparse(ctx.h, gt, i0, i1, k_i, k_next, _u64p(r), self._w_next.h,
                                                     ctypes.byref(h)))
        self.h = h

    def num_vars(self):
        return self._num_vars


def start_round_w(ctx, circuit, evaluation, i, r_i):
    """the W polynomial Prover::start_round builds for layer i (gkr-protocol/src/lib.rs:373-423)"""
    k_next = circuit.num_vars_at(i + 1)
    w_b = DenseMultilinearExtension.from_evaluations_vec(ctx, k_next, np.array(evaluation[i + 1], dtype=np.uint64))
    add_i, mul_i = wiring(ctx, circuit, i, r_i)
    assert add_i.num_vars() == 2 * w_b.num_vars()                        # :419
    return W.new(add_i, mul_i, w_b, w_b)


def line(field, b, c):
    """gkr-protocol/src/lib.rs:278-289: l_i(t) = b_i + t (c_i - b_i)"""
    from .sum_check_protocol import SparsePolynomial
    return [SparsePolynomial.from_coefficients_vec(field, [(0, bi), (1, field.sub(ci, bi))]) for bi, ci in zip(b, c)]


def restrict_poly(b, c, mle):
    """gkr-protocol/src/lib.rs:291-321: W~ restricted to the line through b and c"""
    from .sum_check_protocol import SparsePolynomial
    ctx = mle.ctx
    k = mle.num_vars()
    bb, cc = _words(b), _words(c)
    assert bb.size == k and cc.size == k
    out = np.zeros(k + 1, dtype=np.uint64)
    ctx.check(ctx.lib.sc_table_restrict_to_line(ctx.h, mle.h, _u64p(bb), _u64p(cc), k, _u64p(out)))
    return SparsePolynomial.from_coefficients_vec(ctx.field, [(d, int(v)) for d, v in enumerate(out)])
