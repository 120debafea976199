"""Round-2 additions to the C ABI: the per-launch log, the new options, the peer-transport entry points' state
machine, and the poisoned-context / error behaviour."""
import ctypes

import numpy as np
import pytest

from conftest import load_package
from util import GOLD, pyref

pytestmark = pytest.mark.gpu


def test_launch_log_records_what_ran():
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(GOLD))
    n = 20
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    g = pkg.matrix_multiplication.G(a, b)
    assert ctx.launch_log() == []                       # nothing is recorded unless asked
    pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert ctx.launch_log() == []
    ctx.set_option("time_kernels", 1)
    pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    log = ctx.launch_log(reset=False)
    assert log == ctx.launch_log(reset=True) and ctx.launch_log() == []
    # the schedule of DESIGN.md section 4 at n = 20: three grid passes (4 + 5 + 4 rounds), the host serves the last seven; sizes
    # shrink by kf
    assert log[0]["kind"] == "grid_pass" and (log[0]["kf"], log[0]["ks"], log[0]["log_in"]) == (0, 4, 20)
    size, rounds = 20, 0
    for r in log:
        assert r["kind"] in ("pass", "tail_pass", "grid_pass") and r["log_in"] == size and r["ms"] > 0
        assert r["bytes_read"] == 16 << size and r["bytes_written"] == ((16 << (size - r["kf"])) if r["kf"] else 0)
        size -= r["kf"]
        rounds += r["ks"]
    # the smallest tables: up to five rounds per launch; the last seven rounds are the host's (option "host_tail_log": the third
    # launch left its 2^11-entry tables in pinned host memory) - no record, there is no launch
    assert rounds == 13 and size == 11 and log[-1]["kind"] == "grid_pass" and len(log) == 3
    n_launch, ms = ctx.kernel_time(reset=True)
    assert n_launch == len(log) and abs(ms - sum(r["ms"] for r in log)) < 1e-6   # the totals of the same records
    assert ctx.kernel_time(reset=True) == (0, 0.0)
    # single-table operations are logged with their own kinds and byte counts
    pt = [int(x) for x in np.arange(1, n + 1, dtype=np.uint64)]
    a.evaluate(pt)
    a.fix_variables(pt[:1])
    a.fix_variables(pt[:3])
    a.fix_variables(pt[:10])
    kinds = [(r["kind"], r["kf"], r["bytes_read"], r["bytes_written"]) for r in ctx.launch_log()]
    assert kinds == [("evaluate", n, 8 << n, 0), ("fold", 1, 8 << n, 8 << (n - 1)), ("fold", 3, 8 << n, 8 << (n - 3)),
                     ("fix_low", 10, 8 << n, 8 << (n - 10))]
    ctx.set_option("time_kernels", 0)


def test_options_round_trip_and_validation():
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(GOLD))
    for key, good, bad in [("grid_pass", 0, None), ("grid_log", 21, 99), ("grid_max_vars", 3, 0), ("arena_log", 12, 2),
                           ("peer_spin_ms", 100, 0), ("peer_connect_ms", 5000, 0), ("dbg_delay_ms", 3, -1),
                           ("tail_log", 14, -1), ("vars_per_pass", 1, 3)]:
        ctx.set_option(key, good)
        assert ctx.get_option(key) == good
        if bad is not None:
            with pytest.raises(pkg.SumcheckHipError) as ei:
                ctx.set_option(key, bad)
            assert ei.value.code == 1
            assert ctx.get_option(key) == good
    with pytest.raises(pkg.SumcheckHipError):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(pkg.SumcheckHipError):
        ctx.get_option("no_such_option")


def test_peer_transport_state_machine():
    pkg = load_package()
    lib = pkg.load()
    ctx = pkg.Context(pkg.Field(GOLD))
    buf = (ctypes.c_uint8 * 64)()
    with pytest.raises(pkg.SumcheckHipError) as ei:
        ctx.comm_peer_connect([bytes(64)])               # connect before export
    assert ei.value.code == 5
    assert lib.sc_ctx_comm_peer_export(ctx.h, 0, 16, buf) == 1     # more than 8 ranks
    assert lib.sc_ctx_comm_peer_export(ctx.h, 3, 2, buf) == 1      # rank out of range
    handle = ctx.comm_peer_export(0, 1)
    assert len(handle) == 64 and any(handle)
    with pytest.raises(pkg.SumcheckHipError) as ei:
        ctx.comm_peer_export(0, 1)                       # twice
    assert ei.value.code == 5
    with pytest.raises(pkg.SumcheckHipError) as ei:
        ctx.set_option("arena_log", 12)                  # the region is sized already
    assert ei.value.code == 5
    ctx.comm_peer_connect([handle])
    assert ctx.rank_world() == (0, 1)
    with pytest.raises(pkg.SumcheckHipError):
        ctx.comm_peer_connect([handle])                  # twice
    # one rank through the whole sharded path: exchange with itself, gather with itself
    ctx.set_option("tail_log", 6)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, 14)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, 14)
    g = pkg.matrix_multiplication.G(a, b)
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    plain = pkg.Context(pkg.Field(GOLD))
    a2 = pkg.DenseMultilinearExtension.generate(plain, pyref.SEED_A, 14)
    b2 = pkg.DenseMultilinearExtension.generate(plain, pyref.SEED_B, 14)
    c1b, evalsb, _ = pkg.matrix_multiplication.prove(plain, pkg.matrix_multiplication.G(a2, b2), pyref.SEED_R)
    assert c1 == c1b and np.array_equal(evals, evalsb)
    # a gather larger than the arena is refused with a message, not a crash
    small = pkg.Context(pkg.Field(GOLD))
    small.set_option("arena_log", 4)
    small.comm_peer_connect([small.comm_peer_export(0, 1)])
    small.set_option("tail_log", 10)                     # asks for a gather at 2^10 entries per rank: capped to the arena (2^4)
    a3 = pkg.DenseMultilinearExtension.generate(small, pyref.SEED_A, 12)
    b3 = pkg.DenseMultilinearExtension.generate(small, pyref.SEED_B, 12)
    c1c, evalsc, _ = pkg.matrix_multiplication.prove(small, pkg.matrix_multiplication.G(a3, b3), pyref.SEED_R)
    a4 = pkg.DenseMultilinearExtension.generate(plain, pyref.SEED_A, 12)
    b4 = pkg.DenseMultilinearExtension.generate(plain, pyref.SEED_B, 12)
    c1d, evalsd, _ = pkg.matrix_multiplication.prove(plain, pkg.matrix_multiplication.G(a4, b4), pyref.SEED_R)
    assert c1c == c1d and np.array_equal(evalsc, evalsd)


def test_randomized_differential_run():
    """tools/fuzz_diff.py for 20 s on a fixed seed: random primes (3 ... 2^64 - 59), sizes, schedules, handles of 1-8 entries, table and
    challenge corner values; the product prover, the table calls, a GKR layer and a triangle proof against the oracle, bit for bit"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_diff.py"), "20", "7", "14"], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "0 mismatches" in out.stdout


def test_more_provers_than_tail_slots_on_one_context():
    """a context has 32 pinned tail slots; the 33rd prover alive at the same time has none, so the plan it runs is NOT the one
    sc_plan_proof prints for the default options (no host tail: the device serves it to the last round) - same transcript, and the
    slot comes back when a holder is destroyed"""
    from util import challenges, oracle
    pkg = load_package()
    o = oracle(GOLD)
    ctx = pkg.Context(pkg.Field(GOLD))
    n = 13
    ch = challenges(o, n)
    ref = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), ch)
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, n)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, n)
    G = pkg.matrix_multiplication.G(a, b)
    provers = [G.native_prover() for _ in range(40)]
    ctx.set_option("time_kernels", 1)
    logs = {}
    for k in (0, 31, 32, 39):
        ctx.launch_log(reset=True)
        assert provers[k].c1() == ref["c_1"]
        for j in range(n):
            assert provers[k].round_evals(int(ch[j - 1]) if j else 1, j) == [int(x) for x in ref["evals"][j]], (k, j)
        logs[k] = [(r["kind"], r["kf"], r["ks"]) for r in ctx.launch_log(reset=True)]
    # provers 0 and 31 hold a slot: their second launch hands over and the host serves the rest; 32 and 39 have none: the device
    # goes on (one more launch here), which is the plan of host_tail_log = 0, not the default one
    assert logs[0] == logs[31] and logs[32] == logs[39] and len(logs[32]) > len(logs[0]), logs
    with_tail = [(s["action"], s["kf"], s["ks"]) for s in pkg.schedule.plan_proof(n) if s["action"] != "host_tail"]
    without = [(s["action"], s["kf"], s["ks"]) for s in pkg.schedule.plan_proof(n, host_tail_log=0) if s["action"] != "host_tail"]
    assert logs[0] == with_tail[1:] and logs[32] == without[1:], (logs, with_tail, without)      # ([0]: round 0's pass ran at creation)
    ctx.set_option("time_kernels", 0)
    del provers
    c1, evals, _ = pkg.matrix_multiplication.prove(ctx, G, pyref.SEED_R)      # slots are back: the default plan again
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
    ctx.close()
