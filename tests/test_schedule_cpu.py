"""The prover's schedule as pure host logic (sc_plan_proof = the planner the engine runs at every pass): invariants over
every (num_vars, world, transport, options) the library accepts - checked WITHOUT a GPU.  The GPU suite checks that
the launches of real proofs are exactly these plans (tests/test_gpu_schedule.py)."""
import itertools

import pytest

from conftest import load_package


@pytest.fixture(scope="module")
def plan():
    return load_package().schedule.plan_proof


OPTION_SETS = [
    {}, {"grid_pass": 0}, {"vars_per_pass": 1}, {"first_pass_vars": 1}, {"first_pass_vars": 2}, {"first_pass_vars": 3},
    {"grid_max_vars": 1}, {"grid_max_vars": 3}, {"grid_log": 3}, {"grid_log": 26}, {"grid_sharded": 0}, {"grid_sharded": 0, "tail_log": 4},
    {"grid_sharded": 0, "tail_log": 0, "grid_pass": 0}, {"use_mailbox": 0}, {"grid_max_vars": 4, "grid_log": 12, "first_pass_vars": 2},
    {"gram_log": 0}, {"gram_log": 14}, {"first_pass_vars": 4}, {"first_pass_vars": 4, "grid_log": 10}, {"gram_log": 20, "grid_pass": 0},
    {"host_tail_log": 0}, {"host_tail_log": 3}, {"host_tail_log": 7, "grid_max_vars": 2}, {"host_tail_log": 0, "grid_pass": 0}, {"host_tail_log": 10},
    {"wfold_log": 0}, {"wfold_log": 25}, {"wfold_always": 1}, {"wfold_always": 1, "wfold_min_log": 12, "first_pass_vars": 4},
    {"wfold_always": 1, "wfold_min_log": 12, "grid_log": 26, "gram_log": 0}, {"wfold_always": 1, "host_tail_log": 0},
    {"wfold5_min_log": 12}, {"wfold5_min_log": 12, "wfold_min_log": 12, "first_pass_vars": 4, "wfold_always": 1, "host_tail_log": 0},
    {"tail_log": 26}, {"tail_log": 24, "wfold_always": 1}, {"tail_log": 30, "wfold_always": 1, "grid_log": 14}, {"tail_log": 0},
    {"first_pass_vars": 4, "wfold_min_log": 12, "wfold5_min_log": 12, "grid_log": 8}, {"grid_log": 12, "wfold5_min_log": 14, "wfold_min_log": 14, "gram_log": 14},
]


def check(steps, n, world, transport, opts):
    g = world.bit_length() - 1
    served = sum(s["ks"] for s in steps)
    assert served == n, (served, n)
    cur_log, kf, sharded = n - g, 0, transport != "none" and not (transport == "local" and world == 1)
    gmax = opts.get("grid_max_vars", 5)
    htl = opts.get("host_tail_log", 12)
    prev = None
    served_before = 0
    for s in steps:
        assert s["log_in"] == cur_log and s["sharded"] == (sharded if s["action"] != "gather" else True), (s, cur_log, sharded)
        if s["action"] == "gather":
            assert sharded and world > 1 or transport != "none"
            cur_log += g
            sharded = False
            continue
        assert s["kf"] == kf, (s, kf)
        if transport == "local" and world > 1:
            # the devices of one multi-device handle: never a gather or a rank pass; the host finishes the device bits
            assert s["action"] in ("pass", "grid_pass", "host_tail", "gram_pass", "wfold_pass")
        if s["action"] == "host_tail":
            # the host finishes: always the last step, it serves every round that is left; only a whole prover or the shards of a
            # multi-device handle hand over (the shards of the other transports live in other processes), and only tables the
            # option admits - written by a five-round pass, or <= 32 entries of any pass - or, on a handle, shards that hold
            # nothing but their pending challenges (wherever those are)
            local = transport == "local" and sharded
            assert s is steps[-1] and s["ks"] == n - sum(x["ks"] for x in steps[:-1]) and kf <= 5
            assert opts.get("use_mailbox", 1) == 1 or (local and cur_log == kf)
            assert not sharded or local
            limit = min(max(htl, 5) if local else htl, 12)
            handed = prev is not None and prev["kf"] > 0 and cur_log <= limit and (prev["action"] == "grid_pass" or cur_log <= 5)
            assert handed or (local and cur_log == kf), (s, prev, limit)
            cur_log, sharded = 0, False
        elif s["action"] == "rank_pass":
            assert transport == "peer" and sharded and cur_log == kf and s["ks"] == g and 1 <= g <= 3 and kf <= 5
            cur_log, sharded = g, False
        elif s["action"] == "gram_pass":
            # the four-round first pass on the matrix cores: first launch only, the default two-round schedule only, tables or shards
            # of >= 2^gram_log entries (first_pass_vars = 4: any size from 2^14), sharded provers where they take grid passes
            assert s is steps[0] and kf == 0 and s["ks"] == 4 and cur_log >= 14 and (not sharded or opts.get("grid_sharded", 1) == 1)
            assert opts.get("vars_per_pass", 2) == 2 and opts.get("use_mailbox", 1) == 1 and opts.get("first_pass_vars", 0) in (0, 4)
            assert opts.get("first_pass_vars", 0) == 4 or (opts.get("gram_log", 21) > 0 and cur_log >= opts.get("gram_log", 21))
        elif s["action"] == "wfold_pass":
            # the fold behind a four-round first pass that serves five rounds: round 4, four pending challenges, tables (shards) of
            # 2^max(12, wfold_min_log) .. 2^wfold_log entries that keep six variables, and a grid pass must be able to take the
            # five challenges it leaves (the folded table of THAT pass <= 2^grid_log); sharded provers where they take grid passes
            # ... or a grid pass with five challenges to fold over such a table, in the same kernel's (5, ks) form (never the pass that
            # hands over to the host: those tables are small)
            assert max(12, opts.get("wfold_min_log", 21) if kf == 4 else opts.get("wfold5_min_log", 24)) <= cur_log <= opts.get("wfold_log", 40)
            assert opts.get("grid_pass", 1) == 1 and opts.get("use_mailbox", 1) == 1 and opts.get("vars_per_pass", 2) == 2
            assert not sharded or opts.get("grid_sharded", 1) == 1
            nxt = next(x for x in steps[steps.index(s) + 1:] if x["action"] != "gather")
            if kf == 4:
                assert s["ks"] == 5 and served_before == 4 and cur_log - kf >= 6 and n - served_before >= 6 and gmax >= 5
                assert nxt["action"] in ("grid_pass", "wfold_pass") and nxt["kf"] == 5
                # (the pass behind it: a grid pass, or - folded table too large for one - the streaming form)
                assert nxt["log_in"] - 5 <= opts.get("grid_log", 20) or nxt["action"] == "wfold_pass"
            else:
                # a grid pass's job on a large table - or, where the folded table is too large for a grid pass (whole provers and
                # the shards of a handle), the only kernel that folds five challenges there
                assert kf == 5 and 3 <= s["ks"] <= gmax and cur_log >= kf + s["ks"]
                assert cur_log - kf <= opts.get("grid_log", 20) or not sharded or transport == "local"
                assert nxt["action"] != "host_tail"
            cur_log -= kf
        elif s["action"] == "pass":
            assert (kf <= 3 or (kf == 4 and s["ks"] == 2 and cur_log >= 12)) and 1 <= s["ks"] <= 3      # (pass_kernel<4,2>: whole tiles)
            assert (s["ks"] < 3 or kf == 0) and cur_log >= kf + s["ks"]
            assert s["ks"] <= max(opts.get("vars_per_pass", 2), 1) or (kf == 0 and s["ks"] == 3)
            cur_log -= kf
        else:
            assert s["action"] == "grid_pass" and kf <= 5 and 1 <= s["ks"] <= gmax and cur_log >= kf + s["ks"]
            assert cur_log - kf <= opts.get("grid_log", 20) and opts.get("grid_pass", 1) == 1 and opts.get("use_mailbox", 1) == 1
            assert not sharded or opts.get("grid_sharded", 1) == 1
            cur_log -= kf
        if s["action"] in ("pass", "grid_pass") and s is not steps[-1] and s["kf"] > 0 and opts.get("use_mailbox", 1) == 1:
            # a pass that COULD hand over (rule above) does: the next step is the host tail
            local = transport == "local" and sharded
            limit = min(max(htl, 5) if local else htl, 12)
            if (not sharded or local) and cur_log <= limit and (s["action"] == "grid_pass" or cur_log <= 5):
                assert steps[steps.index(s) + 1]["action"] == "host_tail", (s, steps)
        kf = s["ks"]
        served_before += s["ks"]
        prev = s
    # whatever is left after the last launch are the variables its cached grid serves
    assert cur_log >= 0


@pytest.mark.parametrize("opts", OPTION_SETS, ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()) or "default")
def test_plan_invariants(plan, opts):
    for world, transport in [(1, "none"), (1, "peer"), (1, "rccl"), (2, "peer"), (2, "rccl"), (4, "host"), (8, "peer"), (8, "rccl"), (8, "host"),
                             (1, "local"), (2, "local"), (4, "local"), (8, "local")]:
        g = world.bit_length() - 1
        for n in range(max(g, 1), 41):
            steps = plan(n, world, transport, **opts)     # none of these combinations may be refused
            check(steps, n, world, transport, opts)


def test_known_schedules(plan):
    def sig(steps):
        return [(s["action"], s["kf"], s["ks"], s["log_in"]) for s in steps]
    W525 = "grid_pass"        # (the (5, ks) form of the streaming kernel: from 2^24-entry tables, wfold5_min_log)

    # the headline: n = 28 on one GPU (bench.py config.schedule of every run): four rounds from the matrix-core pass, five from the
    # fold behind it (wfold_pass_kernel), three grid passes of 27 / 243 / 81 cells, and the host finishes from the 2^11-entry
    # tables the fifth launch leaves: FIVE launches (round 4: seven; the 27-cell first pass, gram_log = 0, needs seven)
    # (the third launch folds five challenges over 2^24-entry tables: the streaming kernel's (5, 3) form)
    assert sig(plan(28)) == [("gram_pass", 0, 4, 28), ("wfold_pass", 4, 5, 28), ("wfold_pass", 5, 3, 24), ("grid_pass", 3, 5, 19),
                             ("grid_pass", 5, 4, 16), ("host_tail", 4, 7, 11)]
    assert sig(plan(28, wfold_log=0)) == [("gram_pass", 0, 4, 28), ("pass", 4, 2, 28), ("pass", 2, 2, 24), ("grid_pass", 2, 4, 22),
                                          ("grid_pass", 4, 5, 20), ("grid_pass", 5, 4, 16), ("host_tail", 4, 7, 11)]
    # (round 5's first half: the host took over at 2^10 entries wherever the even split happened to pass that size)
    assert sig(plan(28, wfold_log=0, host_tail_log=10)) == [("gram_pass", 0, 4, 28), ("pass", 4, 2, 28), ("pass", 2, 2, 24), ("grid_pass", 2, 5, 22),
                                                             ("grid_pass", 5, 5, 20), ("grid_pass", 5, 5, 15), ("host_tail", 5, 5, 10)]
    assert sig(plan(28, host_tail_log=0)) == [("gram_pass", 0, 4, 28), ("wfold_pass", 4, 5, 28), ("wfold_pass", 5, 5, 24), ("grid_pass", 5, 5, 19),
                                              ("grid_pass", 5, 5, 14), ("grid_pass", 5, 4, 9)]
    assert sig(plan(28, gram_log=0)) == [("pass", 0, 3, 28), ("pass", 3, 2, 28), ("pass", 2, 2, 25), ("pass", 2, 2, 23), ("grid_pass", 2, 3, 21),
                                         ("grid_pass", 3, 5, 19), ("grid_pass", 5, 4, 16), ("host_tail", 4, 7, 11)]
    # the shard of an 8-GPU run as a proof of its own: FOUR launches (round 4: seven; first half of round 5: five)
    assert sig(plan(25)) == [("gram_pass", 0, 4, 25), ("wfold_pass", 4, 5, 25), (W525, 5, 5, 21), ("grid_pass", 5, 4, 16), ("host_tail", 4, 7, 11)]
    assert sig(plan(25, wfold_log=0)) == [("gram_pass", 0, 4, 25), ("pass", 4, 2, 25), ("grid_pass", 2, 3, 21), ("grid_pass", 3, 5, 19),
                                          ("grid_pass", 5, 4, 16), ("host_tail", 4, 7, 11)]
    # the wfold pass is taken where it saves a launch (25, 27, 28, 29) or replaces a grid pass over the whole table (21 .. 24), not
    # where pass_kernel<4,2> does the same in as many launches (26) or no grid pass could take its five challenges (30)
    assert [n for n in range(14, 41) if any(x["action"] == "wfold_pass" for x in plan(n))] == list(range(21, 41))
    # from n = 30 the pass behind it folds to more than 2^grid_log entries: the streaming form is that pass (5 launches where
    # pass_kernel<4,2> and the two-round passes behind it need 7 and 8)
    assert sig(plan(30)) == [("gram_pass", 0, 4, 30), ("wfold_pass", 4, 5, 30), ("wfold_pass", 5, 5, 26), ("grid_pass", 5, 5, 21), ("grid_pass", 5, 4, 16),
                             ("host_tail", 4, 7, 11)]
    assert (len(plan(30, wfold_log=0)), len(plan(31)), len(plan(31, wfold_log=0))) == (8, 6, 9)
    for n, saved in ((25, 1), (26, 1), (27, 1), (28, 1), (29, 2), (30, 2), (31, 3)):
        assert len(plan(n)) == len(plan(n, wfold_log=0)) - saved
    # the hand-over: <= 2^12 entries (host_tail_log), aimed at 2^11 unless the larger one saves a launch - n = 17, 21, 26
    assert [n for n in range(1, 34) if plan(n)[-1]["action"] == "host_tail" and plan(n)[-1]["log_in"] == 12] == [17, 21, 26, 31]
    for n in (17, 21, 26, 31):
        assert len(plan(n)) == len(plan(n, host_tail_log=11)) - 1
    assert sig(plan(26)) == [("gram_pass", 0, 4, 26), ("wfold_pass", 4, 5, 26), ("grid_pass", 5, 5, 22), ("grid_pass", 5, 4, 17), ("host_tail", 4, 8, 12)]
    assert all(len(plan(n)) == len(plan(n, host_tail_log=11)) for n in range(1, 34) if n not in (17, 21, 26, 31))
    assert sig(plan(20))[0] == ("grid_pass", 0, 4, 20) and sig(plan(21))[0] == ("gram_pass", 0, 4, 21)
    assert sig(plan(20, first_pass_vars=4))[:2] == [("gram_pass", 0, 4, 20), ("grid_pass", 4, 5, 20)]
    # BASELINE config 4: n = 28 over 8 ranks, peer transport - seven launches, seven exchanges, no gather
    s8 = plan(28, 8, "peer")
    assert sig(s8) == [("gram_pass", 0, 4, 25), ("pass", 4, 2, 25), ("grid_pass", 2, 5, 21), ("grid_pass", 5, 5, 19), ("grid_pass", 5, 5, 14),
                       ("grid_pass", 5, 4, 9), ("rank_pass", 4, 3, 4)]
    assert all(s["sharded"] for s in s8)
    assert sig(plan(28, 8, "peer", host_tail_log=0)) == sig(s8)     # (the shards of a peer / RCCL / host plane never hand over)
    assert sig(plan(28, 8, "peer", gram_log=0)) == [("pass", 0, 3, 25), ("pass", 3, 2, 25), ("grid_pass", 2, 5, 22), ("grid_pass", 5, 5, 20),
                                                    ("grid_pass", 5, 5, 15), ("grid_pass", 5, 5, 10), ("rank_pass", 5, 3, 5)]
    # the same over RCCL (and host callbacks), where every sharded pass costs a collective: three sharded launches, ONE all-gather of
    # the 2^16-entry shards (tail_log), two launches on the whole table and the host - five launches, three all-reduces (round 5's
    # first half: eight and six).  tail_log = 0: the shards go on down to their pending challenges as on the peer plane
    r8 = plan(28, 8, "rccl")
    assert sig(r8) == [("gram_pass", 0, 4, 25), ("wfold_pass", 4, 5, 25), ("grid_pass", 5, 4, 21), ("gather", 0, 0, 16), ("grid_pass", 4, 5, 19),
                       ("grid_pass", 5, 5, 15), ("host_tail", 5, 5, 10)]
    assert [x["sharded"] for x in r8] == [True, True, True, True, False, False, False] and sig(plan(28, 8, "host")) == sig(r8)
    r80 = plan(28, 8, "rccl", tail_log=0)
    assert sig(r80)[:6] == sig(s8)[:6] and sig(r80)[6:] == [("gather", 0, 0, 4), ("grid_pass", 4, 3, 7)]
    # ONE process over 8 devices (sc_ctx_create_multi): the n = 25 schedule on every device - FOUR launches - then every launcher
    # thread folds the four pending challenges of the 2^11 entries per table its device handed over and the host serves the ten
    # rounds that are left - no gather
    l8 = plan(28, 8, "local")
    assert sig(l8)[:4] == sig(plan(25))[:4] and sig(l8)[4:] == [("host_tail", 4, 10, 11)] and all(x["sharded"] for x in l8)
    # (option off: round 4's rule - the host takes over from <= 32 entries per table and device)
    l8o = plan(28, 8, "local", host_tail_log=0)
    assert sig(l8o) == [("gram_pass", 0, 4, 25), ("pass", 4, 2, 25), ("grid_pass", 2, 4, 21), ("grid_pass", 4, 5, 19), ("grid_pass", 5, 5, 15),
                        ("grid_pass", 5, 5, 10), ("host_tail", 5, 3, 5)]
    assert sig(plan(3, 8, "local")) == [("host_tail", 0, 3, 0)]      # one entry per device: the host serves every round
    assert sig(plan(28, 1, "local")) == sig(plan(28))                  # one device behind the handle: the plain schedule
    # two rounds per pass with a gather at 2^16-entry shards (grid_sharded 0): round 1's sharded schedule
    t8 = plan(28, 8, "rccl", grid_sharded=0)
    assert [s["action"] for s in t8].count("gather") == 1 and t8[[s["action"] for s in t8].index("gather")]["log_in"] == 16
    # small proofs are grid passes alone, five rounds per launch
    assert sig(plan(12)) == [("grid_pass", 0, 4, 12), ("grid_pass", 4, 4, 12), ("host_tail", 4, 4, 8)]
    assert sig(plan(12, host_tail_log=0)) == [("grid_pass", 0, 4, 12), ("grid_pass", 4, 4, 12), ("grid_pass", 4, 4, 8)]
    assert len(plan(20)) == 4 and len(plan(5)) == 1 and sig(plan(10)) == [("grid_pass", 0, 5, 10), ("grid_pass", 5, 5, 10)]
    # aiming at the hand-over saves a launch where the even split passes the limit late: n = 16 in two launches
    assert sig(plan(16)) == [("grid_pass", 0, 5, 16), ("grid_pass", 5, 4, 16), ("host_tail", 4, 7, 11)]
    assert len(plan(16, host_tail_log=10)) == 4


def test_new_planner_never_needs_more_launches(plan):
    """aiming at the hand-over and the five-round fold only ever remove launches: against the even split with the hand-over at
    2^10 entries (round 5's first half) and against the device alone, for whole provers and the shards of a handle"""
    def launches(steps):
        return sum(1 for s in steps if s["action"] != "host_tail")
    for world, transport in [(1, "none"), (2, "local"), (8, "local")]:
        g = world.bit_length() - 1
        for n in range(max(g, 1), 41):
            new = launches(plan(n, world, transport))
            assert new <= launches(plan(n, world, transport, wfold_log=0, host_tail_log=10)), (n, world)
            assert new <= launches(plan(n, world, transport, wfold_log=0)), (n, world)
            assert new <= launches(plan(n, world, transport, host_tail_log=0)), (n, world)
    # the rank transports never hand over while sharded: the wfold pass must not cost them a launch either
    for world, transport in [(2, "peer"), (8, "peer"), (8, "rccl"), (4, "host")]:
        for n in range(world.bit_length() - 1, 41):
            assert launches(plan(n, world, transport)) <= launches(plan(n, world, transport, wfold_log=0)), (n, world, transport)


def test_random_option_sets_are_never_refused(plan):
    """the option space the GPU fuzz draws from (tools/fuzz_diff.py) and beyond, on the planner alone: every combination the library
    accepts yields a plan (no state without a kernel - found by the fuzz once: a streaming pass that left four challenges to a
    table too small for pass_kernel<4,2>) and the plan keeps the invariants above"""
    import random
    rng = random.Random(20261004)
    for _ in range(6000):
        opts = {}
        if rng.random() < 0.85:
            opts = {"vars_per_pass": rng.choice([1, 2, 2, 2]), "first_pass_vars": rng.choice([0, 0, 1, 2, 3, 4, 4]), "grid_pass": rng.choice([0, 1, 1, 1]),
                    "grid_log": rng.choice([0, 3, 6, 9, 12, 16, 20, 26]), "grid_max_vars": rng.randint(1, 5), "tail_log": rng.choice([0, 2, 5, 9, 14, 16, 24]),
                    "gram_log": rng.choice([0, 14, 15, 17, 21, 28]), "host_tail_log": rng.choice([0, 2, 5, 8, 10, 11, 12]),
                    "wfold_log": rng.choice([0, 16, 40, 40]), "wfold_min_log": rng.choice([12, 12, 14, 21]), "wfold_always": rng.randint(0, 1),
                    "wfold5_min_log": rng.choice([12, 12, 15, 24]), "grid_sharded": rng.choice([0, 1, 1]), "use_mailbox": rng.choice([0, 1, 1, 1, 1])}
            for k in rng.sample(sorted(opts), rng.randint(0, 6)):      # (some at their defaults)
                del opts[k]
        world, transport = rng.choice([(1, "none"), (1, "none"), (1, "local"), (2, "local"), (8, "local"), (2, "peer"), (8, "peer"), (4, "rccl"), (8, "host")])
        n = rng.randint(max(world.bit_length() - 1, 1), 40)
        steps = plan(n, world, transport, **opts)
        check(steps, n, world, transport, opts)


def test_plan_argument_checks(plan):
    pkg = load_package()
    for bad in [dict(num_vars=2, world=8, transport="peer"), dict(num_vars=10, world=3, transport="peer"),
                dict(num_vars=10, world=2, transport="none"), dict(num_vars=10, world=1, transport="none", grid_max_vars=6),
                dict(num_vars=10, world=1, transport="none", vars_per_pass=3), dict(num_vars=10, world=1, transport="none", host_tail_log=13),
                dict(num_vars=10, world=1, transport="none", wfold_log=5), dict(num_vars=10, world=1, transport="none", wfold_min_log=3)]:
        with pytest.raises(pkg.SumcheckHipError) as ei:
            plan(**bad)
        assert ei.value.code == 1
    with pytest.raises(KeyError):
        plan(10, no_such_option=1)


def test_plan_options_struct_is_versioned():
    """sc_plan_options carries the size the CALLER compiled; the library writes and reads nothing beyond it, fields a
    caller's (older, shorter) struct lacks take their defaults, and an uninitialised struct is refused"""
    import ctypes
    pkg = load_package()
    lib, L = pkg._lib.load(), pkg._lib
    assert lib.sc_abi_version() == L.ABI_VERSION == 6
    assert not hasattr(lib, "sc_plan_options_default")      # the old symbol is gone: a stale caller fails at load time
    full = L.ScPlanOptions()
    lib.sc_plan_options_init(ctypes.byref(full), ctypes.sizeof(full))
    assert full.struct_size == ctypes.sizeof(full) == 60 and full.host_tail_log == 12 and full.gram_log == 21
    assert (full.wfold_log, full.wfold_min_log, full.wfold_always, full.wfold5_min_log) == (40, 21, 0, 24)

    class Old(ctypes.Structure):      # a caller built before host_tail_log existed, with a guard word behind its struct
        _fields_ = [("struct_size", ctypes.c_uint32)] + [(k, ctypes.c_int32) for k in (
            "vars_per_pass", "first_pass_vars", "grid_pass", "grid_log", "grid_max_vars", "grid_sharded", "tail_log", "use_mailbox", "gram_log")] + \
            [("guard", ctypes.c_int32)]
    old = Old()
    old.guard = 0x5A5A5A5A
    lib.sc_plan_options_init(ctypes.cast(ctypes.byref(old), ctypes.POINTER(L.ScPlanOptions)), 40)
    assert old.struct_size == 40 and old.guard == 0x5A5A5A5A and old.gram_log == 21
    old.guard = 0          # (would read as host_tail_log = 0 if the library looked past struct_size)
    steps = (L.ScPlanStep * 64)()
    n = ctypes.c_size_t()
    assert lib.sc_plan_proof(ctypes.cast(ctypes.byref(old), ctypes.POINTER(L.ScPlanOptions)), 28, 1, 0, steps, 64, ctypes.byref(n)) == 0
    assert L.PLAN_ACTIONS[steps[n.value - 1].action] == "host_tail"          # the default, not the guard word
    blank = L.ScPlanOptions()
    assert lib.sc_plan_proof(ctypes.byref(blank), 28, 1, 0, steps, 64, ctypes.byref(n)) == 1      # SC_ERR_ARG: never initialised


# ---- the rule chain against a table: a dynamic programme over the kernels that exist ----------------------
# plan_pass decides by a chain of predicates (engine/abi_prover.inc) that has grown an exception per round.  What the chain is FOR
# is simple to state: serve all n rounds with the launches that exist, with as few launches as possible up to the hand-over to the
# host (a launch costs ~13 us of fixed time + ~7 us between two launches, profiles/r05_wfold_ab.txt, profiles/r06_finish_cost.txt -
# more than anything a choice of kernel changes on the tables where there is a choice).  The table below lists, per kernel, the
# states (kf pending challenges, ks rounds served, 2^L entries read) it accepts at DEFAULT options; a DP over (round, kf, L) finds
# the fewest launches any chain of them needs.  The planner must never need more - for every n up to 40, whole provers and the
# shards of 2 / 4 / 8-device handles - and, where several chains tie, must hand over at <= 2^11 entries unless 2^12 saves a launch.

GRAM_LOG, GRID_LOG, WFOLD_MIN, WFOLD5_MIN, TAIL_HARD, TAIL_SMALL = 21, 20, 21, 24, 12, 5


def _launches(j, kf, L, local):
    """every (kind, ks) one launch can be at this state, default options"""
    out = []
    if j == 0 and kf == 0 and L >= GRAM_LOG:
        out.append(("gram_pass", 4))
    if j == 4 and kf == 4 and L >= max(12, WFOLD_MIN):
        out.append(("wfold_pass", 5))
    if kf == 5 and L >= 12 and (L >= WFOLD5_MIN or L - 5 > GRID_LOG):
        out += [("wfold_pass", ks) for ks in (3, 4, 5)]
    if 1 <= L - kf <= GRID_LOG and kf <= 5:
        out += [("grid_pass", ks) for ks in range(1, 6)]
    if kf == 0 and j == 0:
        out += [("pass", ks) for ks in (1, 2, 3)]
    elif kf <= 3:
        out += [("pass", ks) for ks in (1, 2)]
    elif kf == 4 and L >= 12:
        out.append(("pass", 2))
    return [(k, ks) for k, ks in out if ks <= L - kf]        # (the folded table keeps the variables the rounds are about)


def _min_launches(n, g):
    """fewest launches from the start to the hand-over (or the last round), and the smallest hand-over size that count allows"""
    import functools
    local = g > 0

    @functools.lru_cache(maxsize=None)
    def best(j, kf, L):
        # -> (launches, hand-over log or 0)
        if j >= n - g and L == kf:
            return (0, 0)                                    # a handle's shards hold only their pending challenges: the host takes them
        if j >= n:
            return (0, 0)
        res = (10**6, 0)
        for kind, ks in _launches(j, kf, L, local):
            out_log, j2 = L - kf, j + ks
            if j2 > n - g and not (j2 == n - g):
                continue
            if j2 > n:
                continue
            cand = None
            hands = kf > 0 and j2 < n and ((kind == "grid_pass" and out_log <= TAIL_HARD) or out_log <= TAIL_SMALL)
            if hands:
                cand = (1, out_log)
            else:
                sub = best(j2, ks, out_log)
                cand = (1 + sub[0], sub[1])
            if cand[0] < res[0] or (cand[0] == res[0] and max(cand[1], 11) < max(res[1], 11)):
                res = cand
        return res

    return best(0, 0, n - g)


@pytest.mark.parametrize("world", [1, 2, 8])
def test_default_plans_are_launch_minimal(plan, world):
    g = world.bit_length() - 1
    transport = "none" if world == 1 else "local"
    worse = []
    for n in range(max(g, 1), 41):
        steps = [s for s in plan(n, world, transport) if s["action"] != "host_tail"]
        tail = [s for s in plan(n, world, transport) if s["action"] == "host_tail"]
        want, hand_log = _min_launches(n, g)
        if len(steps) > want:
            worse.append((n, len(steps), want))
        # every launch of the plan is a row of the table
        j, kf, L = 0, 0, n - g
        for s in steps:
            assert (s["action"], s["ks"]) in _launches(j, kf, L, g > 0), (n, s)
            j, kf, L = j + s["ks"], s["ks"], L - s["kf"]
        # the hand-over: at <= 2^11 entries unless the table allows a shorter chain only with 2^12
        if tail and tail[0]["log_in"] > 11 and steps:
            assert len(steps) == want and hand_log == 12, (n, tail, want, hand_log)
    assert not worse, "plans with more launches than the kernels need (n, planned, minimal): %r" % worse
