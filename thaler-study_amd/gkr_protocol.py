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
        return self.w_b.num_vars() + self.w_c.num_vars()    # = add_i's (over all ranks on a sharded context)

    def to_evaluations(self):
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_gkr_w_to_evaluations(self.ctx.h, *self._h(), ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h).to_evaluations()

    def native_prover(self):
        return _NativeWProver(self)


def prove_w(ctx, w, seed_r, draw=None):
    """sc_gkr_prove: the whole 2k-round W sumcheck of one layer in one native call (no Python per round).
    Returns (c_1, evals[n][3], challenges[n])."""
    from . import _lib
    n = w.w_b.num_vars() + w.w_c.num_vars()
    ev = np.zeros(3 * max(n, 1), dtype=np.uint64)
    ch = np.zeros(max(n, 1), dtype=np.uint64)
    c1 = u64()
    cb = _lib.DRAW_FN(draw) if draw is not None else ctypes.cast(None, _lib.DRAW_FN)
    ctx.check(ctx.lib.sc_gkr_prove(ctx.h, w.add_i.h, w.mul_i.h, w.w_b.h, w.w_c.h, cb, None, seed_r, ctypes.byref(c1), _u64p(ev), _u64p(ch)))
    return int(c1.value), ev[: 3 * n].reshape(n, 3).copy(), ch[:n].copy()


def _gate_arrays(layer):
    """the gate list as three contiguous arrays (type 0 = add, 1 = mul; the two input labels), cached on the layer"""
    cached = getattr(layer, "_arrays", None)
    if cached is None:
        gt = np.fromiter((0 if g.ttype == GateType.Add else 1 for g in layer.layer), dtype=np.int32, count=len(layer.layer))
        i0 = np.fromiter((g.inputs[0] for g in layer.layer), dtype=np.uint32, count=len(layer.layer))
        i1 = np.fromiter((g.inputs[1] for g in layer.layer), dtype=np.uint32, count=len(layer.layer))
        cached = layer._arrays = (gt, i0, i1)
    gt, i0, i1 = cached
    return (gt.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), i0.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
            i1.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))


def wiring(ctx, circuit, i, r_i):
    """add_i(r_i,.,.) and mul_i(r_i,.,.) as device tables (gkr-protocol/src/lib.rs:388-416)"""
    k_i, k_next = circuit.num_vars_at(i), circuit.num_vars_at(i + 1)
    gt, i0, i1 = _gate_arrays(circuit.layers[i])
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
        k_i, k_next = circuit.num_vars_at(i), circuit.num_vars_at(i + 1)
        gt, i0, i1 = _gate_arrays(circuit.layers[i])
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
    world = ctx.rank_world()[1]                                          # sharded: add_i / mul_i are this rank's rows of c
    assert add_i.num_vars() + world.bit_length() - 1 == 2 * w_b.num_vars()   # :419
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
    return SparsePolynomial.from_dense(ctx.field, [int(v) for v in out])


# ---- the GKR message state machines (gkr-protocol/src/lib.rs:38-218, :324-474) -----------------------
# Host control flow, mirrored so that the reference's protocol tests (protocol_test_from_book :550-624,
# three_layer_protocol_test :626-702) replay on the GPU `W` prover unchanged: every table operation
# below (wiring predicates, W's sumcheck, restrict_poly, the MLE evaluations of the verifier) is a
# libsumcheck_hip.so call.

class WrongVerifierState(Exception):
    """Error::WrongVerifierState (:27-31)"""

    def __init__(self):
        super().__init__("Verifier is in the wrong state.")


class ProverMessage:
    """:255-289"""

    def __init__(self, kind, **fields):
        self.kind = kind
        self.__dict__.update(fields)

    @classmethod
    def Begin(cls, circuit_outputs):
        return cls("Begin", circuit_outputs=list(circuit_outputs))

    @classmethod
    def SumCheckProverMessage(cls, p):
        return cls("SumCheckProverMessage", p=p)

    @classmethod
    def FinalRoundMessage(cls, p, q):
        return cls("FinalRoundMessage", p=p, q=q)

    @classmethod
    def StartSumCheck(cls, c_1, round, num_vars):
        return cls("StartSumCheck", c_1=c_1, round=round, num_vars=num_vars)

    def __eq__(self, other):
        return isinstance(other, ProverMessage) and self.__dict__ == other.__dict__

    def __repr__(self):
        return "ProverMessage.%s(%r)" % (self.kind, {k: v for k, v in self.__dict__.items() if k != "kind"})


class VerifierMessage:
    """:231-252"""

    def __init__(self, kind, **fields):
        self.kind = kind
        self.__dict__.update(fields)

    @classmethod
    def SumCheckRoundResult(cls, res):
        return cls("SumCheckRoundResult", res=res)

    @classmethod
    def RoundStarted(cls, round):
        return cls("RoundStarted", round=round)

    @classmethod
    def R(cls, r):
        return cls("R", r=list(r))

    def __repr__(self):
        return "VerifierMessage.%s(%r)" % (self.kind, {k: v for k, v in self.__dict__.items() if k != "kind"})


class Verifier:
    """:38-218.  `rng` arguments are sum_check_protocol.RngF objects (F::rand(rng) = rng.draw())."""

    def __init__(self, ctx, circuit):
        self.ctx, self.field, self.circuit = ctx, ctx.field, circuit
        self.r, self.m = [], []
        self.state = None                      # VerifierState::Empty

    @classmethod
    def new(cls, ctx, circuit):
        return cls(ctx, circuit)

    def _start_round(self, c_1, round, num_vars):
        """:89-107 - add_i_ext / mul_i_ext (circuit.rs:156-213) straight from the gate list"""
        from .sum_check_protocol import Verifier as SumCheckVerifier
        add_i, mul_i = wiring(self.ctx, self.circuit, round, self.r[-1])
        verifier = SumCheckVerifier.new(num_vars, None, self.field)
        verifier.set_c_1(c_1)
        self.state = {"bc": [], "verifier": verifier, "add_i": add_i, "mul_i": mul_i}
        return VerifierMessage.RoundStarted(round)

    def final_random_point(self, rng):
        """:110-121"""
        from .sum_check_protocol import VerifierRoundResult
        if self.state is None:
            raise WrongVerifierState()
        final_point = rng.draw()
        self.state["bc"].append(final_point)
        return VerifierMessage.SumCheckRoundResult(VerifierRoundResult.JthRound(final_point))

    def _sum_check_step(self, message, rng):
        """:123-139"""
        if self.state is None:
            raise WrongVerifierState()
        res = self.state["verifier"].round(message, rng)        # .unwrap(): a mismatch raises
        if not res.is_final():
            self.state["bc"].append(res.value)
        return VerifierMessage.SumCheckRoundResult(res)

    def _final_round_message(self, p, q, rng):
        """:141-174"""
        if self.state is None:
            raise WrongVerifierState()
        f = self.field
        bc, add_i, mul_i = self.state["bc"], self.state["add_i"], self.state["mul_i"]
        q_0, q_1 = q.evaluate(f.zero), q.evaluate(f.one)
        ev = f.add(f.mul(add_i.evaluate(bc), f.add(q_0, q_1)), f.mul(f.mul(mul_i.evaluate(bc), q_0), q_1))
        if ev != p.evaluate(bc[-1]):
            raise AssertionError("assert_eq!(eval, p.evaluate(bc.last().unwrap()))")        # :151
        r = rng.draw()
        half = len(bc) // 2
        r_next = [l.evaluate(r) for l in line(f, bc[:half], bc[half:])]
        self.r.append(r_next)
        self.m.append(q.evaluate(r))
        return VerifierMessage.R(r_next)

    def receive_prover_msg(self, msg, rng):
        """:177-207"""
        if msg.kind == "SumCheckProverMessage":
            return self._sum_check_step(msg.p, rng)
        if msg.kind == "StartSumCheck":
            return self._start_round(msg.c_1, msg.round, msg.num_vars)
        if msg.kind == "FinalRoundMessage":
            return self._final_round_message(msg.p, msg.q, rng)
        num_output_vars = self.circuit.num_vars_at(0)
        d = DenseMultilinearExtension.from_evaluations_vec(self.ctx, num_output_vars,
                                                           np.array(msg.circuit_outputs, dtype=np.uint64))
        r_zero = [rng.draw() for _ in range(num_output_vars)]
        self.r, self.m = [r_zero], [d.evaluate(r_zero)]
        return VerifierMessage.R(r_zero)

    def check_input(self, input):
        """:210-217"""
        w = DenseMultilinearExtension.from_evaluations_vec(self.ctx, (len(input)).bit_length() - 1,
                                                           np.array(input, dtype=np.uint64))
        return w.evaluate(self.r[-1]) == self.m[-1]


class Prover:
    """:324-474.  `sparse=True` builds each layer's sumcheck straight from the gate list
    (sc_gkr_prover_create_sparse) instead of the dense add_i / mul_i tables - same messages."""

    def __init__(self, ctx, circuit, input, sparse=False):
        self.ctx, self.field, self.circuit = ctx, ctx.field, circuit
        self.evaluation = circuit.evaluate(self.field, list(input))           # :346
        self.i, self.prover, self.w, self.r, self.sparse = 0, None, None, [], sparse

    @classmethod
    def new(cls, ctx, circuit, input, sparse=False):
        return cls(ctx, circuit, input, sparse)

    def start_protocol(self):
        """:363-367"""
        return ProverMessage.Begin(self.evaluation[0])

    def start_round(self, i, r_i):
        """:373-436"""
        from .sum_check_protocol import Prover as SumCheckProver
        k_next = self.circuit.num_vars_at(i + 1)
        if self.sparse:
            eng = SparseLayerProver(self.ctx, self.circuit, self.evaluation, i, r_i)
            self.w = eng._w_next
            prover = _EngineProver(eng, self.field)
            num_vars = 2 * k_next
        else:
            w = start_round_w(self.ctx, self.circuit, self.evaluation, i, r_i)
            self.w = w.w_b
            num_vars = w.add_i.num_vars()
            prover = SumCheckProver.new(w)
        self.i, self.prover, self.r = i, prover, []
        return ProverMessage.StartSumCheck(prover.c_1(), i, num_vars)

    def round_msg(self, j):
        """:439-456"""
        if j == 2 * self.circuit.num_vars_at(self.i + 1) - 1:
            half = len(self.r) // 2
            q = restrict_poly(self.r[:half], self.r[half:], self.w)
            p = self.prover.round(self.r[j - 1] if j else self.field.one, j)
            return ProverMessage.FinalRoundMessage(p, q)
        point = self.field.one if j == 0 else self.r[j - 1]
        return ProverMessage.SumCheckProverMessage(self.prover.round(point, j))

    def receive_verifier_msg(self, verifier_msg):
        """:459-468"""
        if verifier_msg.kind == "SumCheckRoundResult":
            if verifier_msg.res.is_final():
                raise RuntimeError("panic!()")
            self.r.append(verifier_msg.res.value)

    def c_1(self):
        """:471-473"""
        return self.prover.c_1()


class _EngineProver:
    """sum_check_protocol.Prover's surface over a bare native engine (the sparse layer prover has no
    table-backed W to hand to Prover::new)"""

    def __init__(self, engine, field):
        self._engine, self.field = engine, field
        self.c_1_value = engine.c1()

    def c_1(self):
        return self.c_1_value

    def round(self, r_prev, j):
        return self._engine.round(r_prev, j)

    def num_vars(self):
        return self._engine.num_vars()
