"""World-size-2 (and 4) gloo tests on CPU for the N > 1 path: the shard plan, the 32-bit
limb all-reduce, the tail gather and the torch.distributed adapter that
`Context.comm_init_host` uses.  The GPU kernels cannot run here, so the oracle stands in for
them - but the SCHEDULE is the engine's own: sc_plan_proof (pure host code of the product library,
the planner prover_pass calls at every pass) says which launches and gathers a sharded proof
consists of, the model executes them over gloo collectives, and the result is compared against
the full-table oracle transcript (the engine itself: tests/test_gpu_sharded.py,
tests/test_gpu_schedule.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_package

sys.path.insert(0, os.path.join(ROOT, "oracle"))


HOST_TABLES = [None]


def host_round(o, shards, pending, sub, add):
    """engine/abi_prover.inc: host_tail + host_round - every shard folds the challenges it has not folded yet, the folded shards
    in device order are the whole tables, and the round is H(0), H(1), H(inf) over their pairs"""
    parts = [((o.fix_variables(a, pending), o.fix_variables(b, pending)) if pending else (a, b)) for a, b in shards]
    a = np.concatenate([x for x, _ in parts])
    b = np.concatenate([y for _, y in parts])
    HOST_TABLES[0] = (a, b)
    h = [int(x) for x in o.gridk_sums(a, b, 1)]
    t = add(h[1], h[2])
    return [h[0], h[1], sub(add(t, t), h[0])]          # H(2) = 2H(1) - H(0) + 2H(inf)


def model_sharded_prove(o, D, plan_proof, p, n, rank, world, opts, allreduce, allgather, pyref):
    """The sharded prover with the oracle in place of the kernels, driven by the ENGINE'S OWN PLANNER: the launches come
    from sc_plan_proof (the function prover_pass calls at every pass, thaler-study_amd/csrc/engine/abi_prover.inc: plan_pass),
    this model only executes them - fold the pending challenges, accumulate the 3^ks-cell grid, sum its limbs across
    the ranks, gather when the plan says so - and answers the rounds from the grid like prover_answer."""
    start, length = D.shard_range(n, rank, world)
    a = o.generate_range(pyref.SEED_A, start, length)
    b = o.generate_range(pyref.SEED_B, start, length)
    ch = [o.challenge(pyref.SEED_R, j + 1) for j in range(n)]
    steps = list(plan_proof(n, world, "host", **opts))
    sharded, pending, cache = True, [], None
    host = False          # the plan's last step, SC_PLAN_HOST_TAIL: the host serves every remaining round from the (whole) tables
    evals = []
    n_allreduce = n_gather = 0
    L = o.lib
    add, sub, mul = (lambda x, y: L.sco_add(o.fp, x, y)), (lambda x, y: L.sco_sub(o.fp, x, y)), \
        (lambda x, y: L.sco_mul(o.fp, x, y))
    for j in range(n):
        if j:
            pending.append(ch[j - 1])
        covered = cache is not None and 0 <= j - cache[1] < cache[0] and len(pending) == j - cache[1]
        if not covered and not host:
            step = steps.pop(0)
            if step["action"] == "gather":
                assert sharded and step["log_in"] == int(a.size).bit_length() - 1
                a = allgather(a)
                b = allgather(b)
                sharded = False
                n_gather += 1
                step = steps.pop(0)
            if step["action"] == "host_tail":
                # (only a whole prover hands over on a host transport: after the gather) engine/abi_prover.inc: host_tail
                assert not sharded and not steps and step["kf"] == len(pending) and step["ks"] == n - j, (step, j)
                assert step["log_in"] == int(a.size).bit_length() - 1 <= 12
                host = True
        if host:
            evals.append(host_round(o, [[a, b]], pending, sub, add))
            a, b = HOST_TABLES[0]
            pending = []
            continue
        if not covered:
            assert step["action"] in ("pass", "grid_pass", "wfold_pass", "gram_pass"), step      # (rank passes exist on the peer transport only)
            kf, ks = step["kf"], step["ks"]
            assert kf == len(pending) and step["log_in"] == int(a.size).bit_length() - 1 and step["sharded"] == sharded, (step, len(pending), a.size)
            if kf:
                a = o.fix_variables(a, pending)
                b = o.fix_variables(b, pending)
                pending = []
            S = o.gridk_sums(a, b, ks)
            if sharded:
                limbs = D.split_limbs([int(x) for x in S])
                allreduce(limbs)
                n_allreduce += 1
                S = D.recombine_limbs(limbs, p)
            cache = (ks, j, [int(x) for x in S])
        ks, j0, S = cache
        # prover_answer: collapse the leading axes at the challenges received since the pass,
        # the round's variable is the next axis, the axes after it are summed over {0,1}
        grid = list(S)
        cells = len(grid)
        for r in pending:
            r2 = mul(r, r)
            cells //= 3
            grid = [add(add(grid[c], mul(r, sub(sub(grid[cells + c], grid[c]), grid[2 * cells + c]))),
                        mul(r2, grid[2 * cells + c])) for c in range(cells)]
        rest = cells // 3
        h = []
        for x in range(3):
            t = 0
            for c in range(rest):
                digits, d = [], c
                while d:
                    digits.append(d % 3)
                    d //= 3
                if 2 not in digits:
                    t = add(t, grid[x * rest + c])
            h.append(t)
        t = add(h[1], h[2])
        evals.append([h[0], h[1], sub(add(t, t), h[0])])   # H(2) = 2H(1) - H(0) + 2H(inf)
    assert not steps, steps          # every planned launch was needed, none was missing
    return evals, ch, n_allreduce, n_gather


def _worker(rank, world, port, cases, q):
    try:
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        pkg = load_package()
        D = pkg.distributed
        import pyref
        from oracle import Oracle
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
        r, w, _ = D.init_process_group_from_env("gloo")
        assert (r, w) == (rank, world)
        allreduce, allgather = D.torch_collectives()
        # control plane: bytes broadcast (what carries the RCCL unique id)
        payload = D.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
        assert payload == bytes(range(128))
        out = []
        for (p, n, opts) in cases:
            o = Oracle(p)
            evals, ch, nar, ng = model_sharded_prove(o, D, pkg.schedule.plan_proof, p, n, rank, world, opts, allreduce, allgather, pyref)
            full = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), np.array(ch, dtype=np.uint64))
            ok = full["status"] == 0 and evals == [[int(x) for x in row] for row in full["evals"]]
            out.append((p, n, opts, ok, nar, ng))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, out))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "ERR %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_sharded_schedule(world):
    GOLD = 2**64 - 2**32 + 1
    two = {"grid_pass": 0, "first_pass_vars": 2, "tail_log": 0}      # round 1's schedule: two rounds per pass, exchange per pass
    cases = [(GOLD, 10, two), (GOLD, 10, {"vars_per_pass": 1, "tail_log": 0}), (GOLD, 11, dict(two, tail_log=4)), (389, 9, two),
             (GOLD, 3, two), (GOLD, world.bit_length() - 1, two), (5, 8, {"vars_per_pass": 1, "tail_log": 2}),
             # the default schedule: five-round passes on the shards, gathered when only the pending challenges are left
             (GOLD, 10, {}), (GOLD, 14, {}), (389, 13, {}), (GOLD, 3, {}), (GOLD, 4, {}), (GOLD, 12, {"grid_max_vars": 3}),
             (GOLD, 12, {"grid_sharded": 0, "tail_log": 5}), (GOLD, 13, {"first_pass_vars": 3})]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + world + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
    for rank, out in results:
        assert not isinstance(out, str), out
        for (p, n, opts, ok, nar, ng) in out:
            assert ok, (rank, p, n, opts)
            assert ng == 1 or n < world.bit_length(), (n, opts, ng)     # exactly one gather per sharded proof on a host transport
        # with tail_log 0 and n = 10 the two-round schedule used several limb all-reduces
        assert out[0][4] >= 3


def test_shard_plan_and_limbs():
    pkg = load_package()
    D = pkg.distributed
    assert D.shard_range(28, 3, 8) == (3 << 25, 1 << 25)
    assert D.shard_range(3, 7, 8) == (7, 1)
    with pytest.raises(ValueError):
        D.shard_range(2, 0, 8)
    with pytest.raises(ValueError):
        D.shard_range(10, 0, 3)
    p = 2**64 - 2**32 + 1
    vals = [0, 1, p - 1, 2**32, 2**32 - 1, 0x123456789ABCDEF0 % p]
    limbs = D.split_limbs(vals)
    assert all(int(x) < 2**32 for x in limbs)
    assert D.recombine_limbs(limbs, p) == vals
    # eight ranks worth of the worst case cannot wrap a u64 slot and recombines mod p
    total = np.zeros_like(limbs)
    for _ in range(8):
        total += D.split_limbs([p - 1] * len(vals))
    assert D.recombine_limbs(total, p) == [(8 * (p - 1)) % p] * len(vals)


def test_bench_self_launch_relays_exit_code_without_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its rank processes itself (a child launcher; the parent never
    imports torch or touches a GPU) and relays their exit code: here, without a GPU, every rank refuses to run (the HIP
    path has no CPU fallback), so the parent must exit non-zero and print no JSON line"""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--num-vars", "12", "--steps", "1", "--warmup", "0",
                          "--cpu-num-vars", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    from conftest import has_gpu
    if has_gpu():
        return      # on a GPU box the run is real: covered by tests/test_gpu_00_multiprocess.py
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "needs a GPU" in out.stderr


def model_local_prove(o, plan_proof, p, n, world, opts, pyref):
    """The ONE-PROCESS form (sc_ctx_create_multi, transport "local"): the N shards live in one process, every planned launch runs
    on every shard (the oracle in place of the kernels), the shards' grids are added in the process - no collective - and the
    plan's last step is the HOST TAIL: every shard folds its pending challenges, the folded shards are the whole tables and the
    host serves every remaining round from them, as engine/abi_prover.inc: host_tail / host_round do."""
    from conftest import load_package
    D = load_package().distributed
    L = o.lib
    add, sub, mul = (lambda x, y: L.sco_add(o.fp, x, y)), (lambda x, y: L.sco_sub(o.fp, x, y)), \
        (lambda x, y: L.sco_mul(o.fp, x, y))
    g = world.bit_length() - 1
    shards = []
    for rank in range(world):
        start, length = D.shard_range(n, rank, world)
        shards.append([o.generate_range(pyref.SEED_A, start, length), o.generate_range(pyref.SEED_B, start, length)])
    ch = [o.challenge(pyref.SEED_R, j + 1) for j in range(n)]
    steps = list(plan_proof(n, world, "local", **opts))
    # the library's own default hand-over size (sc_plan_options_init), not a literal that ages
    from conftest import load_package as _lp
    _L = _lp()._lib
    _d = _L.ScPlanOptions()
    _lp().load().sc_plan_options_init(ctypes.byref(_d), ctypes.sizeof(_d))
    tail_default = int(_d.host_tail_log)
    pending, cache, evals = [], None, []
    host = False
    for j in range(n):
        if j:
            pending.append(ch[j - 1])
        covered = cache is not None and 0 <= j - cache[1] < cache[0] and len(pending) == j - cache[1]
        if not covered and not host:
            step = steps.pop(0)
            kf, ks = step["kf"], step["ks"]
            assert kf == len(pending) and step["log_in"] == int(shards[0][0].size).bit_length() - 1, (step, len(pending))
            if step["action"] == "host_tail":
                # every shard is down to <= 2^host_tail_log entries (<= 32 with the option off): the host takes over for good
                limit = max(opts.get("host_tail_log", tail_default), 5) if world > 1 else opts.get("host_tail_log", tail_default)
                assert ks == n - j and not steps and (step["log_in"] <= limit or step["log_in"] == kf), (step, j)
                host = True
            else:
                # (gram_pass: four rounds from the first read; wfold_pass: the streaming fold with a grid pass's cells - to this
                # model they are passes that fold kf challenges and serve ks rounds like any other)
                assert step["action"] in ("pass", "grid_pass", "gram_pass", "wfold_pass") and step["sharded"] == (world > 1)
                assert step["action"] != "gram_pass" or (kf == 0 and ks == 4)
                S = None
                for sh in shards:
                    if kf:
                        sh[0], sh[1] = o.fix_variables(sh[0], pending), o.fix_variables(sh[1], pending)
                    part = [int(x) for x in o.gridk_sums(sh[0], sh[1], ks)]
                    S = part if S is None else [add(x, y) for x, y in zip(S, part)]
                pending = []
                cache = (ks, j, S)
        if host:
            evals.append(host_round(o, shards, pending, sub, add))
            shards = [list(HOST_TABLES[0])]
            pending = []
            continue
        ks, j0, S = cache
        grid, cells = list(S), len(S)
        for r in pending:
            r2 = mul(r, r)
            cells //= 3
            grid = [add(add(grid[c], mul(r, sub(sub(grid[cells + c], grid[c]), grid[2 * cells + c]))), mul(r2, grid[2 * cells + c]))
                    for c in range(cells)]
        rest = cells // 3
        h = []
        for x in range(3):
            t = 0
            for c in range(rest):
                digits, d = [], c
                while d:
                    digits.append(d % 3)
                    d //= 3
                if 2 not in digits:
                    t = add(t, grid[x * rest + c])
            h.append(t)
        t = add(h[1], h[2])
        evals.append([h[0], h[1], sub(add(t, t), h[0])])
    assert not steps, steps
    return evals, ch


@pytest.mark.parametrize("world", [1, 2, 8])
def test_local_plan_is_sufficient(world):
    """the planner's schedule for ONE process over N devices - no gather, no rank pass, the host tail at the end - executed
    with the oracle in place of the kernels reproduces the full-table oracle transcript (the engine itself:
    tests/test_gpu_multi.py)"""
    import pyref
    from oracle import Oracle
    pkg = load_package()
    GOLD = 2**64 - 2**32 + 1
    g = world.bit_length() - 1
    for (p, n, opts) in [(GOLD, max(g, 1), {}), (GOLD, g + 1, {}), (GOLD, g + 4, {}), (GOLD, 10, {}), (GOLD, 14, {}), (389, 13, {}),
                         (GOLD, 12, {"grid_max_vars": 3}), (GOLD, 11, {"vars_per_pass": 1}), (GOLD, 12, {"grid_pass": 0}),
                         (5, 9, {"grid_sharded": 0}), (GOLD, 13, {"first_pass_vars": 2, "grid_log": 6}),
                         # the matrix-core first pass and the streaming fold behind it, asked for on small shards ...
                         (GOLD, 15 + g, {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1, "wfold5_min_log": 12}),
                         # ... and where the DEFAULT plan takes them: 2^21-entry tables / shards, hand-over at 2^11 or 2^12 entries
                         (GOLD, 21 + g, {})]:
        o = Oracle(p)
        evals, ch = model_local_prove(o, pkg.schedule.plan_proof, p, n, world, opts, pyref)
        if n >= 21:
            plan = [s["action"] for s in pkg.schedule.plan_proof(n, world, "local", **opts)]
            assert plan[0] == "gram_pass" and "wfold_pass" in plan and plan[-1] == "host_tail", plan
        full = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), np.array(ch, dtype=np.uint64))
        assert full["status"] == 0 and evals == [[int(x) for x in row] for row in full["evals"]], (world, p, n, opts)
