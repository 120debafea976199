"""One rank of the multi-process transport tests (launched by torch.distributed.run; every rank on GPU 0 - or, with
SC_WORKER_DISTINCT_DEVICES=1 on a box that has them, rank r on GPU r: tests/test_gpu_multi_device.py).
SC_PEER_WORKER_MODE = parity | faults | widened | rccl_death | headline.  Not collected by pytest.
SC_WORKER_TRANSPORT = peer (default: the in-kernel exchange over HIP IPC) | rccl (Transport::kRccl: ncclAllReduce behind every
sharded pass, ncclAllGather at the tail - on a one-GPU box through tests/rccl_double, selected by SC_RCCL_LIBRARY, because
RCCL itself refuses two ranks on one device).

The library's bound on an in-kernel wait for a peer (peer_spin_ms, default 2 s) is the skew it tolerates between the
ranks' launches of the same pass; the ranks of a real run call in lockstep.  Here every rank does seconds of oracle
work of its own between proofs, so each proof starts behind a control-plane barrier - the "all ranks are about to
launch" handshake a caller owes the library."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


TRANSPORT = os.environ.get("SC_WORKER_TRANSPORT", "peer")
DISTINCT = os.environ.get("SC_WORKER_DISTINCT_DEVICES", "0") == "1"
DEVICE = int(os.environ.get("LOCAL_RANK", "0")) if DISTINCT else 0


def attach(pkg, ctx, rank, world):
    if TRANSPORT == "rccl":
        pkg.distributed.attach_rccl(ctx, rank, world)
        assert ctx.get_option("transport") == 1
    else:
        pkg.distributed.attach_peer(ctx, rank, world)          # default peer_spin_ms: the connect's handshake absorbed the start-up lag
        assert ctx.get_option("transport") == 3
    assert ctx.rank_world() == (rank, world) and ctx.get_option("comm_nranks") == world


def shard_tables(pkg, ctx, pyref, n, rank, world, seeds=None):
    D = pkg.distributed
    sa, sb = seeds or (pyref.SEED_A, pyref.SEED_B)
    start, length = D.shard_range(n, rank, world)
    nl = length.bit_length() - 1
    a = pkg.DenseMultilinearExtension.generate(ctx, sa, nl, start=start)
    b = pkg.DenseMultilinearExtension.generate(ctx, sb, nl, start=start)
    return a, b, nl


def parity(pkg, dist, pyref, Oracle, rank, world):
    D = pkg.distributed
    for p in (pyref.GOLDILOCKS, 389):
        o = Oracle(p)
        ctx = pkg.Context(pkg.Field(p), device=DEVICE)
        attach(pkg, ctx, rank, world)
        # grid_sharded 1: the shards go on with five-round passes (cells exchanged inside the kernel) down to one entry;
        # 0: two-round passes with the exchange, gather at tail_log, unsharded tail
        # WF: the matrix-core first pass and wfold_pass_kernel - (4, 5), then (5, ks) - on the (small) shards; its cells cross the ranks
        # like a grid pass's
        WF = {"first_pass_vars": 4, "wfold_min_log": 12, "wfold_always": 1, "wfold5_min_log": 12}
        WF_OFF = {"first_pass_vars": 0, "wfold_min_log": 21, "wfold_always": 0, "wfold5_min_log": 24}
        for n, tail_log, gs, extra in [(1, 0, 1, None), (2, 0, 1, None), (5, 0, 1, None), (12, 0, 1, None), (12, 5, 0, None), (12, 0, 0, None),
                                       (16, 12, 1, None), (16, 12, 0, None), (20, 16, 1, None), (20, 16, 0, None), (22, 16, 1, None),
                                       (21, 0, 1, WF), (22, 0, 1, WF)]:
            if n < world.bit_length() - 1:
                continue            # fewer entries than ranks
            ctx.set_option("tail_log", tail_log)
            ctx.set_option("grid_sharded", gs)
            for k, v in (extra or WF_OFF).items():
                ctx.set_option(k, v)
            a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world)
            g = pkg.matrix_multiplication.G(a, b)
            assert g.num_vars() == n
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            dist.barrier()
            c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
            log = ctx.launch_log(reset=True)
            ctx.set_option("time_kernels", 0)
            if extra:
                assert [r["kind"] for r in log][:3] == ["gram_pass", "wfold_pass", "wfold_pass"], log
            if gs and nl >= 6 and TRANSPORT == "peer":      # the shard's own variables are served five at a time, then one small launch for the rank bits
                assert [r["kind"] for r in log].count("grid_pass") >= 2 and log[-1]["log_in"] <= 5 + world.bit_length() - 1, log
            final = g.evaluate([int(x) for x in ch])
            hs = g.hypercube_sum()
            e0 = g.round_evals() if nl >= 1 else None
            oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
            ref = o.prove(oa, ob, ch)
            assert ref["status"] == 0
            assert c1 == ref["c_1"], (p, n, tail_log, "c_1")
            assert np.array_equal(evals, ref["evals"]), (p, n, tail_log, "round polynomials")
            assert final == ref["final_eval"], (p, n, "evaluate")
            assert hs == ref["c_1"]
            if e0 is not None:
                assert e0 == [int(x) for x in ref["evals"][0]]
            del a, b, g
        # ranks fed different challenges must fail loudly (digest in the exchange), on every rank
        n = 10
        a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world)
        g = pkg.matrix_multiplication.G(a, b)
        ctx.set_option("tail_log", 0)
        ctx.set_option("grid_sharded", 1)
        dist.barrier()
        if TRANSPORT == "peer":      # (the digest travels in the in-kernel exchange: the peer plane's check)
            try:
                pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R + (7 if rank == 1 else 0))
                raise AssertionError("different challenges were accepted")
            except pkg.SumcheckHipError as e:
                assert e.code == 5 and "different challenges" in str(e), e
        dist.barrier()
        del a, b, g
        ctx.close()
    print("%s-OK rank %d" % (TRANSPORT.upper(), rank), flush=True)


def widened(pkg, dist, pyref, Oracle, rank, world):
    """BASELINE config 5's pieces over the peer transport between PROCESSES: sharded G::new (+ proof), the sharded GKR W
    prover with sharded wiring, the sharded triangle prover - vector all-reduces and gathers that go through the arenas
    in chunks (arena_log = 8 here: 256 words per rank and chunk), bit-exact against the oracle on every rank"""
    import random
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_gkr import make_circuit, random_circuit
    D = pkg.distributed
    p = pyref.GOLDILOCKS
    o = Oracle(p)
    F = pkg.Field(p)
    g_ = world.bit_length() - 1
    ctx = pkg.Context(F, device=DEVICE)
    if TRANSPORT == "peer":
        ctx.set_option("arena_log", 8)
    attach(pkg, ctx, rank, world)
    # ---- sharded G::new at n = 8 (2^16-entry matrices; the f_a all-reduce moves 2 x 2^8 limbs) + the proof on it
    n = 8
    pt = np.array([o.challenge(pyref.SEED_PT, j) for j in range(2 * n)], dtype=np.uint64)
    fa, fb = o.g_new(n, o.generate(11, 2 * n), o.generate(12, 2 * n), pt)
    ch = np.array([o.challenge(pyref.SEED_R, j + 1) for j in range(n)], dtype=np.uint64)
    ref = o.prove(fa, fb, ch)
    start, length = D.shard_range(2 * n, rank, world)
    nl = length.bit_length() - 1
    At = pkg.DenseMultilinearExtension.generate(ctx, 11, nl, start=start)
    Bt = pkg.DenseMultilinearExtension.generate(ctx, 12, nl, start=start)
    dist.barrier()
    g = pkg.matrix_multiplication.G.new_from_tables(ctx, n, At, Bt, [int(x) for x in pt])
    s0, l0 = D.shard_range(n, rank, world)
    assert np.array_equal(g.f_a.to_evaluations(), fa[s0:s0 + l0]) and np.array_equal(g.f_b.to_evaluations(), fb[s0:s0 + l0])
    dist.barrier()
    c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
    del g, At, Bt
    # ---- GKR layers: sharded wiring, dense W prover on the shards (P / L all-reduce, add_r / mul_r gather), generic sums
    gp = pkg.gkr_protocol
    rng = random.Random(4242)
    for ks in ([5, 4], [6, 7], [4, 9]):
        layers = random_circuit(rng, ks)
        circuit = make_circuit(pkg, layers, 1 << ks[-1])
        inputs = [F.from_int(rng.randrange(p)) for _ in range(1 << ks[-1])]
        evaluation = circuit.evaluate(F, inputs)
        k_i, k_next = ks
        r_i = [F.from_int(rng.randrange(p)) for _ in range(k_i)]
        oadd, omul = o.wiring_fixed(layers[0], k_next, r_i)
        ow = np.array(evaluation[1], dtype=np.uint64)
        chw = [F.from_int(rng.randrange(p)) for _ in range(2 * k_next)]
        refw = o.w_prove(oadd, omul, ow, ow, chw)
        assert refw["status"] == 0
        dist.barrier()
        w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i)
        n_loc = oadd.size // world
        assert np.array_equal(w.add_i.to_evaluations(), oadd[rank * n_loc:(rank + 1) * n_loc])
        dist.barrier()
        assert w.round_evals() == [int(x) for x in refw["evals"][0]]
        dist.barrier()
        eng = w.native_prover()
        assert eng.c1() == refw["c_1"], ks
        for j in range(2 * k_next):
            assert eng.round_evals(chw[j - 1] if j else F.one, j) == [int(x) for x in refw["evals"][j]], (ks, j)
        assert w.evaluate(chw) == refw["final_eval"]
        del eng, w
    # ---- triangle counting, 64 vertices: adjacency rows sharded, the matrix square split across the ranks and gathered
    k = 6
    nv = 1 << k
    words = np.array([F.one if rng.random() < 0.4 else 0 for _ in range(nv * nv)], dtype=np.uint64)
    cht = [F.from_int(rng.randrange(p)) for _ in range(3 * k)]
    reft = o.tri_prove(words, k, cht)
    assert reft["status"] == 0
    rows = nv // world
    t = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * k - g_, words[rank * rows * nv:(rank + 1) * rows * nv])
    tg = pkg.triangle_counting.G(t, t, t, k)
    dist.barrier()
    eng = pkg.triangle_counting._NativeTriProver(tg)
    assert eng.c1() == reft["c_1"]
    for j in range(3 * k):
        assert eng.round_evals(cht[j - 1] if j else F.one, j) == [int(x) for x in reft["evals"][j]], j
    del eng, tg, t
    dist.barrier()
    ctx.close()
    print("WIDENED-OK rank %d" % rank, flush=True)


def faults(pkg, dist, pyref, Oracle, rank, world):
    D = pkg.distributed
    scp = pkg.sum_check_protocol
    p = pyref.GOLDILOCKS
    o = Oracle(p)
    F = pkg.Field(p)

    def reference(n, seeds=None):
        sa, sb = seeds or (pyref.SEED_A, pyref.SEED_B)
        ch = np.array([o.challenge(pyref.SEED_R, j + 1) for j in range(n)], dtype=np.uint64)
        return o.prove(o.generate(sa, n), o.generate(sb, n), ch), ch

    # ---- 1. a late rank: every sharded launch of one rank delayed by 1 .. 100 ms -------------------------------------
    ctx = pkg.Context(F, device=DEVICE)
    D.attach_peer(ctx, rank, world)
    for n, gs, delay_rank, delay in [(14, 1, 0, 1), (14, 0, world - 1, 7), (18, 1, world - 1, 30), (12, 0, 0, 100), (20, 1, 1 % world, 3)]:
        ctx.set_option("grid_sharded", gs)
        ctx.set_option("tail_log", 4)
        ctx.set_option("dbg_delay_ms", delay if rank == delay_rank else 0)
        a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world)
        g = pkg.matrix_multiplication.G(a, b)
        ref, ch = reference(n)
        dist.barrier()
        t0 = time.perf_counter()
        c1, evals, chn = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        took = time.perf_counter() - t0
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"]) and np.array_equal(chn, ch), (n, gs, delay)
        assert g.evaluate([int(x) for x in ch]) == ref["final_eval"]
        assert took < 10.0, took
        del a, b, g
    ctx.set_option("dbg_delay_ms", 0)

    # ---- 2. interleaved sharded provers on one context, gather in between (two-round schedule keeps gathered tables
    #         alive for several rounds: they must be the prover's own copies, not the arena a later gather reuses) ------
    ctx.set_option("grid_sharded", 0)
    ctx.set_option("grid_pass", 0)
    ctx.set_option("first_pass_vars", 2)
    ctx.set_option("tail_log", 6)
    sizes = (9, 11)
    provers, refs, chs = [], [], []
    for k, n in enumerate(sizes):
        a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world, seeds=(51 + k, 61 + k))
        g = pkg.matrix_multiplication.G(a, b)
        ref, ch = reference(n, seeds=(51 + k, 61 + k))
        refs.append(ref)
        chs.append(ch)
        dist.barrier()
        provers.append((scp.Prover.new(g), g))
        assert provers[-1][0].c_1() == ref["c_1"]
    pts = (0, F.one, F.add(F.one, F.one))
    for j in range(max(sizes)):
        for k, (pr, g) in enumerate(provers):
            if j < sizes[k]:
                dist.barrier()
                poly = pr.round(int(chs[k][j - 1]) if j else F.one, j)
                assert [poly.evaluate(x) for x in pts] == [int(x) for x in refs[k]["evals"][j]], (k, j)
                if j % 3 == 1:   # another sharded collective on the same context between the rounds
                    assert g.hypercube_sum() == refs[k]["c_1"]
    del provers
    dist.barrier()
    ctx.close()

    # ---- 3. gathers longer than the arena go in chunks --------------------------------------------------------------
    ctx = pkg.Context(F, device=DEVICE)
    ctx.set_option("arena_log", 6)                 # 64 words per rank and chunk
    D.attach_peer(ctx, rank, world)
    ctx.set_option("grid_sharded", 0)
    ctx.set_option("tail_log", 11)                 # gather 2^11-entry shards: 32 chunks per table
    n = 11 + world.bit_length() - 1 + 2
    a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world)
    g = pkg.matrix_multiplication.G(a, b)
    ref, ch = reference(n)
    dist.barrier()
    c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
    del a, b, g
    dist.barrier()
    ctx.close()

    # ---- 4. a rank out of step (skips an exchange tag): every rank fails with SC_ERR_RCCL within the bound -----------
    for gs in (1, 0):
        ctx = pkg.Context(F, device=DEVICE)
        ctx.set_option("peer_spin_ms", 400)
        D.attach_peer(ctx, rank, world)
        ctx.set_option("grid_sharded", gs)
        ctx.set_option("tail_log", 4)
        n = 14
        a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world)
        g = pkg.matrix_multiplication.G(a, b)
        ref, ch = reference(n)
        dist.barrier()
        c1, evals, _ = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)       # in step: fine
        assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
        if rank == world - 1:
            ctx.set_option("dbg_skip_tag", 1)
        dist.barrier()
        t0 = time.perf_counter()
        try:
            pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
            raise AssertionError("a rank out of step went unnoticed")
        except pkg.SumcheckHipError as e:
            assert e.code == 3 and "did not arrive within 400 ms" in str(e), e
        took = time.perf_counter() - t0
        assert took < 5.0, took                                                    # bounded, not a hang
        try:                                                                       # the context is unusable afterwards and says so
            g.hypercube_sum()
            raise AssertionError("poisoned context accepted a call")
        except pkg.SumcheckHipError as e:
            assert e.code == 5, e
        dist.barrier()
        del a, b, g
        ctx.close()
    print("FAULTS-OK rank %d" % rank, flush=True)


def rccl_death(pkg, dist, pyref, Oracle, rank, world):
    """a rank that dies: the last rank leaves the job between two proofs (os._exit: no destructor, no goodbye); the collective of
    every surviving rank must FAIL - SC_ERR_RCCL within the double's bound (SC_RCCL_DOUBLE_TIMEOUT_MS), a poisoned-or-refusing
    context afterwards - and never hang or return a transcript"""
    p = pyref.GOLDILOCKS
    o = Oracle(p)
    ctx = pkg.Context(pkg.Field(p), device=DEVICE)
    attach(pkg, ctx, rank, world)
    n = 14
    bound_ms = int(os.environ.get("SC_WORKER_RCCL_TIMEOUT_MS", "0"))
    if bound_ms:      # the library's own bound on a collective that never completes (the real librccl; the double in its async mode)
        ctx.set_option("rccl_timeout_ms", bound_ms)
        assert ctx.get_option("rccl_timeout_ms") == bound_ms
    a, b, _ = shard_tables(pkg, ctx, pyref, n, rank, world)
    g = pkg.matrix_multiplication.G(a, b)
    dist.barrier()
    c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)       # everyone is alive: the oracle's transcript
    ref = o.prove(o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n), ch)
    assert c1 == ref["c_1"] and np.array_equal(evals, ref["evals"])
    dist.barrier()
    if rank == world - 1:
        print("RCCL-DEATH rank %d leaves" % rank, flush=True)
        sys.stdout.flush()
        os._exit(0)
    t0 = time.perf_counter()
    try:
        pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        raise AssertionError("a proof finished although a rank is dead")
    except pkg.SumcheckHipError as e:
        assert e.code == 3, e                                                   # SC_ERR_RCCL
        if bound_ms and os.environ.get("SC_RCCL_DOUBLE_ASYNC_HANG") == "1":
            assert "rccl_timeout_ms" in str(e) and "aborted" in str(e), e       # ended by the library's bound, not by the stand-in
    took = time.perf_counter() - t0
    assert took < 15.0 + 1e-3 * bound_ms, took                                  # bounded (1.5 s in the double, bound_ms in the library), not a hang
    try:                                                                        # and every later collective fails at once
        t0 = time.perf_counter()
        pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
        raise AssertionError("a proof finished although a rank is dead")
    except pkg.SumcheckHipError as e:
        assert e.code in (3, 5), e
        assert time.perf_counter() - t0 < 5.0
    print("RCCL-DEATH-OK rank %d" % rank, flush=True)
    sys.stdout.flush()
    os._exit(0)      # (no control-plane barrier: a rank is gone)


def headline(pkg, dist, pyref, Oracle, rank, world):
    """BASELINE configs[3] as it is stated: the n = 28 hypercube over the ranks' devices, one process per GPU, the sums crossing
    the ranks on the chosen data plane - compared round for round with the CPU oracle's transcript (rank 0 computes it with the
    all-cores form of the reference-shaped prover and hands it to the others over the control plane)."""
    import torch
    p = pyref.GOLDILOCKS
    o = Oracle(p)
    ctx = pkg.Context(pkg.Field(p), device=DEVICE)
    attach(pkg, ctx, rank, world)
    devs = [None] * world
    dist.all_gather_object(devs, (DEVICE, torch.cuda.device_count()))
    if DISTINCT:
        assert len({d for d, _ in devs}) == world, devs                          # one GPU per rank, all different
    for n in [int(x) for x in os.environ.get("SC_WORKER_NUM_VARS", "20,28").split(",")]:
        ref = [None]
        if rank == 0:
            oa, ob = o.generate(pyref.SEED_A, n), o.generate(pyref.SEED_B, n)
            ch = np.array([o.challenge(pyref.SEED_R, j + 1) for j in range(n)], dtype=np.uint64)
            c1, ev = o.prover_run_mt(oa, ob, ch)
            del oa, ob
            ref = [(int(c1), ev.tolist(), ch.tolist())]
        dist.broadcast_object_list(ref, src=0)
        c1_ref, ev_ref, ch_ref = ref[0]
        a, b, nl = shard_tables(pkg, ctx, pyref, n, rank, world)
        g = pkg.matrix_multiplication.G(a, b)
        for rep in range(2):                                                       # cold, then warm (pool, communicator, tickets reused)
            ctx.set_option("time_kernels", 1)
            ctx.launch_log(reset=True)
            dist.barrier()
            c1, evals, ch = pkg.matrix_multiplication.prove(ctx, g, pyref.SEED_R)
            log = ctx.launch_log(reset=True)
            ctx.set_option("time_kernels", 0)
            assert [int(x) for x in ch] == ch_ref and c1 == c1_ref, (n, rep, "c_1")
            bad = [j for j in range(n) if [int(x) for x in evals[j]] != ev_ref[j]]
            assert not bad, (n, rep, "rounds that differ from the oracle", bad)
            plan = pkg.schedule.plan_proof(n, world, "rccl" if TRANSPORT == "rccl" else "peer")
            # (the launch log files the peer plane's rank pass under the five-round passes' kind)
            assert [(r["kind"], r["kf"], r["ks"]) for r in log] == \
                [("grid_pass" if s["action"] == "rank_pass" else s["action"], s["kf"], s["ks"]) for s in plan if s["action"] not in ("host_tail", "gather")], (log, plan)
        assert g.evaluate(ch_ref) is not None
        del a, b, g
        dist.barrier()
    assert ctx.get_option("comm_nranks") == world
    ctx.close()
    print("HEADLINE-OK rank %d device %d" % (rank, DEVICE), flush=True)


def main():
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import pyref
    from oracle import Oracle
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(DEVICE)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    mode = os.environ.get("SC_PEER_WORKER_MODE", "parity")
    {"parity": parity, "faults": faults, "widened": widened, "rccl_death": rccl_death, "headline": headline}[mode](pkg, dist, pyref, Oracle, rank, world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
