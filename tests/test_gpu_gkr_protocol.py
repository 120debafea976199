"""End-to-end GKR on the GPU `W` prover: the reference's protocol tests replayed through the host
mirrors of gkr_protocol::{Prover, Verifier} (gkr-protocol/src/lib.rs:550-702) - outputs [36,6] / [2,2],
every layer's sumcheck verifier accepts, restrict_poly hands the claim to the next layer, check_input -
and compared message by message with the oracle's restatement of the same loop on the same randomness."""
import random

import numpy as np
import pytest

from conftest import load_package
from test_gpu_gkr import make_circuit, random_circuit
from test_host_protocols import BOOK, THREE, gkr_draw_count
from util import GOLD, pid, pyref

pytestmark = pytest.mark.gpu


class Scripted:
    """RngF fed from a list (canonical ints -> Montgomery words), in the order the reference draws"""

    def __init__(self, F, draws):
        self.F, self.draws, self.used = F, list(draws), 0

    def draw(self):
        v = self.F.from_int(self.draws[self.used])
        self.used += 1
        return v


def dense(F, poly, n):
    out = [0] * n
    for d, c in poly.coeffs:
        out[d] = F.to_int(c)
    return out


def run_protocol(pkg, ctx, layers, num_inputs, inputs, draws, sparse):
    """the body of protocol_test_from_book / three_layer_protocol_test, recording every message"""
    gp = pkg.gkr_protocol
    F = ctx.field
    circuit = make_circuit(pkg, layers, num_inputs)
    rng = Scripted(F, draws)
    win = F.from_ints(inputs).tolist()
    prover = gp.Prover.new(ctx, circuit, win, sparse=sparse)
    begin = prover.start_protocol()
    verifier = gp.Verifier.new(ctx, circuit)
    msg = verifier.receive_prover_msg(begin, rng)
    assert msg.kind == "R"
    r_i = msg.r
    rec = {"circuit_outputs": F.to_ints(begin.circuit_outputs), "r_0": F.to_ints(r_i), "m_0": F.to_int(verifier.m[0]),
           "layers": []}
    for i in range(len(circuit.layers)):
        start = prover.start_round(i, r_i)
        num_vars = 2 * circuit.num_vars_at(i + 1)
        assert start.kind == "StartSumCheck" and start.num_vars == num_vars and start.round == i
        assert verifier.receive_prover_msg(start, rng).kind == "RoundStarted"
        coeffs = []
        for j in range(num_vars - 1):
            pm = prover.round_msg(j)
            assert pm.kind == "SumCheckProverMessage"
            coeffs.append(dense(F, pm.p, 3))
            vm = verifier.receive_prover_msg(pm, rng)
            assert vm.kind == "SumCheckRoundResult" and not vm.res.is_final()
            prover.receive_verifier_msg(vm)
        prover.receive_verifier_msg(verifier.final_random_point(rng))
        pm = prover.round_msg(num_vars - 1)
        assert pm.kind == "FinalRoundMessage"
        coeffs.append(dense(F, pm.p, 3))
        vm = verifier.receive_prover_msg(pm, rng)
        assert vm.kind == "R"
        r_i = vm.r
        rec["layers"].append({"c_1": F.to_int(start.c_1), "coeffs": coeffs, "q": dense(F, pm.q, num_vars // 2 + 1),
                              "r_next": F.to_ints(r_i), "m_next": F.to_int(verifier.m[-1])})
    rec["check_input"] = verifier.check_input(win)
    assert rng.used == len(draws)
    return rec, verifier


def compare(rec, ref):
    assert rec["circuit_outputs"] == ref["circuit_outputs"]
    assert rec["r_0"] == ref["r_0"] and rec["m_0"] == ref["m_0"]
    for i, (a, b) in enumerate(zip(rec["layers"], ref["layers"])):
        assert a["c_1"] == b["c_1"], i
        want = [c + [0] * (3 - len(c)) for c in b["coeffs"]]
        assert a["coeffs"] == want, i
        assert a["q"] == b["q"] + [0] * (len(a["q"]) - len(b["q"])), i
        assert a["r_next"] == b["r_next"] and a["m_next"] == b["m_next"], i
    assert rec["check_input"] == ref["check_input"]


@pytest.mark.parametrize("sparse", [False, True], ids=["dense", "sparse"])
@pytest.mark.parametrize("layers,num_inputs,inputs,outputs", [
    (BOOK, 4, [3, 2, 3, 1], [36, 6]),              # protocol_test_from_book
    (THREE, 8, [0, 1] * 4, [2, 2]),                # three_layer_protocol_test
], ids=["book", "three_layer"])
def test_reference_protocol_tests_replayed(layers, num_inputs, inputs, outputs, sparse):
    pkg = load_package()
    p = 389
    ctx = pkg.Context(pkg.Field(p))
    for seed in range(6):
        rng = random.Random(100 + seed)
        draws = [rng.randrange(p) for _ in range(gkr_draw_count(layers, num_inputs))]
        rec, verifier = run_protocol(pkg, ctx, layers, num_inputs, inputs, draws, sparse)
        assert rec["circuit_outputs"] == outputs                     # :569-585, :649-663
        assert rec["check_input"] is True                            # :623, :701
        compare(rec, pyref.gkr_transcript(layers, num_inputs, inputs, draws, p))
        # a different input does not satisfy the final claim
        bad = ctx.field.from_ints([(inputs[0] + 1) % p] + list(inputs[1:])).tolist()
        assert verifier.check_input(bad) is False


@pytest.mark.parametrize("p", [GOLD, 389], ids=pid)
def test_random_deep_circuits(p):
    """deeper random circuits (mixed add/mul, fan-in collisions), dense and sparse layer provers, vs oracle"""
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(p))
    rng = random.Random(p % 1000)
    for ks in ([1, 2, 3, 2], [2, 3, 4, 4, 3], [3, 5, 4], [1, 1, 1, 1]):
        layers = random_circuit(rng, ks)
        num_inputs = 1 << ks[-1]
        inputs = [rng.randrange(p) for _ in range(num_inputs)]
        draws = [rng.randrange(p) for _ in range(gkr_draw_count(layers, num_inputs))]
        ref = pyref.gkr_transcript(layers, num_inputs, inputs, draws, p)
        assert ref["check_input"]
        for sparse in (False, True):
            rec, _ = run_protocol(pkg, ctx, layers, num_inputs, inputs, draws, sparse)
            compare(rec, ref)


def test_cheating_prover_is_caught():
    """a prover that lies about an output is rejected at the first layer's sumcheck claim"""
    pkg = load_package()
    gp, scp = pkg.gkr_protocol, pkg.sum_check_protocol
    p = 389
    ctx = pkg.Context(pkg.Field(p))
    F = ctx.field
    circuit = make_circuit(pkg, BOOK, 4)
    rng = Scripted(F, [random.Random(4).randrange(p) for _ in range(64)])
    prover = gp.Prover.new(ctx, circuit, F.from_ints([3, 2, 3, 1]).tolist())
    begin = prover.start_protocol()
    lie = gp.ProverMessage.Begin([F.from_int(37)] + begin.circuit_outputs[1:])
    verifier = gp.Verifier.new(ctx, circuit)
    r_0 = verifier.receive_prover_msg(lie, rng).r
    start = prover.start_round(0, r_0)
    # the honest layer sumcheck proves W_0~(r_0) of the TRUE outputs; the verifier's m_0 is of the lie
    assert start.c_1 != verifier.m[0]
    with pytest.raises(gp.WrongVerifierState):
        gp.Verifier.new(ctx, circuit).final_random_point(rng)
    # feeding the layer sumcheck a claim that does not match its first polynomial raises
    verifier.receive_prover_msg(gp.ProverMessage.StartSumCheck(verifier.m[0], 0, start.num_vars), rng)
    with pytest.raises(scp.ProverClaimMismatch):
        verifier.receive_prover_msg(prover.round_msg(0), rng)
