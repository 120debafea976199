"""The launches of real proofs are exactly what sc_plan_proof plans (the engine calls the same planner at every pass):
unsharded over sizes and options, and sharded on virtual ranks - so the CPU-side invariants of tests/test_schedule_cpu.py
speak about the engine that runs."""
import numpy as np
import pytest

from conftest import load_package
from test_gpu_sharded import run_virtual_ranks, Loopback
from util import GOLD, pyref

pytestmark = pytest.mark.gpu


def launches(pkg, ctx, n, start=0, nl=None):
    nl = n if nl is None else nl
    a = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_A, nl, start=start)
    b = pkg.DenseMultilinearExtension.generate(ctx, pyref.SEED_B, nl, start=start)
    g = pkg.matrix_multiplication.G(a, b)
    ctx.set_option("time_kernels", 1)
    ctx.launch_log(reset=True)
    pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    log = ctx.launch_log(reset=True)
    ctx.set_option("time_kernels", 0)
    return [(r["kind"], r["kf"], r["ks"], r["log_in"]) for r in log]


@pytest.mark.parametrize("opts", [{}, {"grid_pass": 0}, {"vars_per_pass": 1}, {"first_pass_vars": 2}, {"grid_max_vars": 3}, {"grid_log": 8},
                                  {"first_pass_vars": 3, "grid_max_vars": 4}, {"first_pass_vars": 4}, {"gram_log": 0}, {"gram_log": 19, "grid_log": 12},
                                  {"host_tail_log": 0}, {"host_tail_log": 4}, {"host_tail_log": 8, "grid_max_vars": 3}], ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()) or "default")
def test_unsharded_launches_equal_the_plan(opts):
    pkg = load_package()
    ctx = pkg.Context(pkg.Field(GOLD))
    for k, v in opts.items():
        ctx.set_option(k, v)
    for n in (1, 2, 3, 5, 8, 11, 12, 16, 19, 22, 24):
        # (the host tail - always the plan's last step - is not a launch)
        plan = [(s["action"], s["kf"], s["ks"], s["log_in"]) for s in pkg.schedule.plan_proof(n, **opts) if s["action"] != "host_tail"]
        assert launches(pkg, ctx, n) == plan, (n, opts)
    ctx.close()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_launches_equal_the_plan(world):
    """virtual ranks over host callbacks: every rank's pass launches are the plan's steps (the gather is a collective, not
    a launch of the log; rank passes exist on the peer transport only)"""
    import threading
    pkg = load_package()
    g = world.bit_length() - 1
    for n, opts in [(12, {}), (16, {}), (20, {}), (16, {"grid_sharded": 0, "tail_log": 6}), (14, {"grid_pass": 0, "tail_log": 3}),
                    (20, {"first_pass_vars": 4}), (21, {"gram_log": 15, "grid_log": 10})]:
        plan = [(s["action"], s["kf"], s["ks"], s["log_in"]) for s in pkg.schedule.plan_proof(n, world, "host", **opts) if s["action"] not in ("gather", "host_tail")]
        lb = Loopback(world)
        got, errors = [None] * world, []

        def body(rank):
            try:
                ctx = pkg.Context(pkg.Field(GOLD))
                for k, v in opts.items():
                    ctx.set_option(k, v)
                ar, ag = lb.collectives(rank)
                ctx.comm_init_host(rank, world, ar, ag)
                start, length = pkg.distributed.shard_range(n, rank, world)
                got[rank] = launches(pkg, ctx, n, start=start, nl=length.bit_length() - 1)
                ctx.close()
            except Exception as e:  # pragma: no cover
                import traceback
                traceback.print_exc()
                errors.append(e)
                lb.barrier.abort()

        threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        for rank in range(world):
            assert got[rank] == plan, (n, opts, rank, got[rank], plan)
