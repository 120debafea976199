#!/usr/bin/env python3
"""bench.py - field mul-adds/s of the sumcheck prover (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload prover|mle]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
(both forms work for N > 1: without WORLD_SIZE in the environment bench.py starts the N rank processes itself, as a
child launcher, before it has touched the GPU)

--workload prover (default).  One "step" = one complete interactive sumcheck proof of the synthetic
n-variable instance of BASELINE.md section 3 (g = a*b, two 2^n-entry Goldilocks tables resident in HBM
before the timed region): Prover::new + n rounds, the host drawing each challenge only after that
round's sums were read back - the timed region of the reference's criterion bench
(matrix-multiplication/benches/mm_benchmark.rs:88-96).  n = 28 (BASELINE.json configs[3], the
configuration the metric is quoted on; 4 GiB of tables, fits one GPU).
N > 1: strong scaling - the same 2^28 hypercube sharded by its top log2(N) index bits, one process per
GPU, one exchange of the round sums per device pass.  Both in-library data planes are timed, each in its own
barrier-bracketed region of exactly K proofs: the in-kernel exchange through peer-mapped inboxes and RCCL
(one ncclAllReduce of the pass's split limbs, the collective BASELINE.json names); `value` is the faster one,
`config.transports` carries both with the rank count each transport reports (ncclCommCount for RCCL).

--workload mle.  BASELINE.json configs[1]: multilinear-extensions evaluate + fix_variable on ONE table
of 2^n entries (n = 24 by default; --num-vars 28 for the large shape).  One step = evaluate (LE),
evaluate (BE = vsbw_/cti_multilinear_from_evaluations), fix_variables of k = 1, 3 and n/2 low variables.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`
objects.  Every byte count in `roofline` comes from the launches that really ran (sc_ctx_launch_log:
kernel kind, input size, HIP-event duration) - `frac` is bytes actually moved through HBM by the
dominant kernel / its measured duration / 8 TB/s and is <= 1 by construction; SURVEY section 8d's
one-round-per-pass byte model is reported separately as `sec8d_credited_frac`.  The CPU baseline and
every correctness check use oracle/ as the checker only; the measured path is libsumcheck_hip.so.
"""
import argparse
import json
import os
import statistics
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the pool's driver only supports dmabuf IPC (RCCL)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def lagrange_at(F, e, r):
    inv2 = F.inv(F.two)
    l0 = F.mul(F.mul(F.sub(r, F.one), F.sub(r, F.two)), inv2)
    l1 = F.neg(F.mul(r, F.sub(r, F.two)))
    l2 = F.mul(F.mul(r, F.sub(r, F.one)), inv2)
    return F.add(F.add(F.mul(l0, int(e[0])), F.mul(l1, int(e[1]))), F.mul(l2, int(e[2])))


def check_identities(F, c1, evals, ch, final_eval):
    """the verifier's checks, sum-check-protocol/src/lib.rs:286, :316-318, :303"""
    claim = c1
    for j in range(len(evals)):
        if F.add(int(evals[j][0]), int(evals[j][1])) != claim:
            return "round %d: g_j(0)+g_j(1) != previous claim" % j
        claim = lagrange_at(F, evals[j], int(ch[j]))
    if claim != final_eval:
        return "g_n(r_n) != g(r)"
    return None


def kernel_name(rec):
    k = rec["kind"]
    if k == "pass":
        return "sc::pass_kernel<GoldilocksMont,%d,%d> on 2^%d-entry tables" % (rec["kf"], rec["ks"], rec["log_in"])
    if k == "tail_pass":
        return "sc::grid_pass3_kernel<GoldilocksMont> (kf=%d, ks=3; small_pass3_kernel with mid_pass=0) on 2^%d-entry tables" % (rec["kf"], rec["log_in"])
    if k == "grid_pass":
        return "sc::wgrid_pass_kernel<GoldilocksMont,ks> (kf=%d, ks=%d) on 2^%d-entry tables" % (rec["kf"], rec["ks"], rec["log_in"])
    if k == "tail_resident":
        return "sc::tail_resident_kernel<GoldilocksMont> from 2^%d-entry tables (%d rounds)" % (rec["log_in"], rec["ks"])
    if k == "evaluate":
        return "sc::evaluate_kernel<GoldilocksMont> on a 2^%d-entry table" % rec["log_in"]
    if k == "fold":
        return "sc::fold_kernel<GoldilocksMont,%d> on a 2^%d-entry table" % (rec["kf"], rec["log_in"])
    if k == "fix_low":
        return "sc::fix_low_kernel<GoldilocksMont> (%d variables) on a 2^%d-entry table" % (rec["kf"], rec["log_in"])
    return "%s kf=%d ks=%d 2^%d" % (k, rec["kf"], rec["ks"], rec["log_in"])


def aggregate_launches(log, steps_sampled):
    """group the timed launches by (kind, kf, ks, log_in): calls per step, mean duration, bytes"""
    groups = {}
    for r in log:
        key = (r["kind"], r["kf"], r["ks"], r["log_in"])
        g = groups.setdefault(key, {"rec": r, "n": 0, "ms": 0.0})
        g["n"] += 1
        g["ms"] += r["ms"]
    out = []
    for key, g in groups.items():
        r = g["rec"]
        nbytes = r["bytes_read"] + r["bytes_written"]
        avg_ms = g["ms"] / g["n"]
        out.append({"kernel": kernel_name(r), "launches_per_step": g["n"] / max(steps_sampled, 1),
                    "avg_us": avg_ms * 1e3, "bytes_per_launch": nbytes,
                    "GBps": nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                    "ms_per_step": g["ms"] / max(steps_sampled, 1), "_key": key})
    out.sort(key=lambda x: -x["ms_per_step"])
    return out


def load_traffic(key):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        return json.load(f).get(key)


def attach_plane(plane, pkg, ctx, rank, world, dist):
    """Join this rank's context to one data plane ("peer" | "rccl" | "host") and prove a small sharded instance through
    it.  Every rank executes the same control-plane collectives whatever happens locally (a one-sided failure must not
    leave the others waiting in a rendezvous); returns (ok_on_every_rank, reason)."""
    import torch
    D = pkg.distributed

    def agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def bail():
        errs = [None] * world
        dist.all_gather_object(errs, err)
        return False, next((e for e in errs if e), "a rank failed to attach")

    ok, err = True, ""
    if os.environ.get("SC_BENCH_FAIL_TRANSPORT_RANK") in (str(rank), "all"):
        ok, err = False, "injected transport failure (SC_BENCH_FAIL_TRANSPORT_RANK)"   # test hook
    if plane == "peer":
        handle = None
        if ok:
            try:
                handle = ctx.comm_peer_export(rank, world)
            except Exception as e:  # pragma: no cover - depends on the node
                ok, err = False, str(e)
        handles = [None] * world
        dist.all_gather_object(handles, handle)
        if ok and all(h is not None for h in handles):
            try:
                ctx.comm_peer_connect(handles)      # hello handshake + the library's own exchange / gather self-test
            except Exception as e:  # pragma: no cover
                ok, err = False, str(e)
        else:
            ok = False
    elif plane == "rccl":
        uid = None
        if rank == 0 and ok:
            try:
                uid = pkg.Context.rccl_unique_id()
            except Exception as e:  # pragma: no cover
                ok, err = False, str(e)
        box = [uid]
        dist.broadcast_object_list(box, src=0)
        ok = ok and box[0] is not None
        if not agree(ok):             # nobody enters ncclCommInitRank unless everybody does (it blocks until all ranks arrive)
            return bail()
        try:
            ctx.comm_init_rccl(box[0], rank, world)
        except Exception as e:  # pragma: no cover - e.g. two ranks on one device: RCCL refuses duplicate GPUs
            ok, err = False, str(e)
    else:
        ar, ag = D.torch_collectives()
        ctx.comm_init_host(rank, world, ar, ag)
    if not agree(ok):
        return bail()
    # a small sharded proof; every rank must produce the same transcript
    mine, why = None, ""
    try:
        mm, syn = pkg.matrix_multiplication, pkg.synthetic
        start, length = D.shard_range(16, rank, world)
        a, b = syn.tables(ctx, length.bit_length() - 1, start=start)
        g = mm.G(a, b)
        dist.barrier()
        c1, evals, ch = mm.prove(ctx, g, syn.SEED_R)
        problem = check_identities(ctx.field, c1, evals, ch, g.evaluate([int(x) for x in ch]))
        if problem:
            raise RuntimeError("sharded self-test: " + problem)
        mine = (c1, evals.tobytes())
    except Exception as e:  # pragma: no cover - depends on the node
        why = str(e)
    seen = [None] * world
    dist.all_gather_object(seen, (mine, why))
    if any(s[0] is None for s in seen) or any(s[0] != seen[0][0] for s in seen):
        return False, next((s[1] for s in seen if s[1]), "ranks disagree on the self-test transcript")
    return True, ""


def planes_to_time(world):
    """data planes a run times.  N > 1: BOTH in-library planes by default - the in-kernel peer exchange and RCCL (the
    all-reduce per pass BASELINE.json names) - each in its own timed region of exactly K steps; the headline is the
    faster one and config.transports carries both.  SC_BENCH_TRANSPORT=peer|rccl|host restricts the run to one."""
    want = os.environ.get("SC_BENCH_TRANSPORT", "")
    if world == 1:
        if os.environ.get("SC_BENCH_FORCE_RCCL") == "1" or want in ("rccl", "peer"):
            return [want or "rccl"]     # diagnostic: the sharded code path with one rank on a single-GPU box
        return ["none"]
    if want:
        if want not in ("peer", "rccl", "host"):
            raise SystemExit("SC_BENCH_TRANSPORT must be peer, rccl or host")
        return [want]
    return ["peer", "rccl"]


def cpu_sample_size(requested):
    if requested >= 0:
        return requested
    avail_gib = 0.0
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    avail_gib = int(line.split()[1]) / 2**20
    except OSError:
        pass
    # ~10-30 s of single-core work: n = 28 needs ~14 GiB of host memory, n = 27 ~7 GiB
    return 28 if avail_gib > 48 else (27 if avail_gib > 20 else 26)


def context_options():
    """SC_BENCH_OPTIONS="first_ring=0,ring_log=24": library options for A/B runs (reported in config.options)"""
    out = []
    for item in os.environ.get("SC_BENCH_OPTIONS", "").split(","):
        if item.strip():
            k, v = item.split("=")
            out.append((k.strip(), int(v)))
    return out


def time_plane(args, pkg, torch, dist, rank, world, local_rank, plane, n):
    """One data plane end to end: context, this rank's shards, exactly W untimed and K timed proofs (barrier +
    synchronize on both sides, max over ranks), the verifier-identity gate on the last transcript.  Returns a dict;
    {"ok": False, "error": ...} when the plane could not be set up on every rank."""
    mm, D, syn = pkg.matrix_multiplication, pkg.distributed, pkg.synthetic
    F = pkg.Field(pkg.GOLDILOCKS)

    def barrier():
        if dist is not None:
            dist.barrier()

    ctx = pkg.Context(F, device=local_rank)
    ctx.set_option("vars_per_pass", args.vars_per_pass)
    for k, v in context_options():
        ctx.set_option(k, v)
    label = plane
    if plane != "none":
        if world == 1:
            (D.attach_peer if plane == "peer" else D.attach_rccl)(ctx, 0, 1)
            label = "%s(world=1, diagnostic)" % plane
        else:
            ok, why = attach_plane(plane, pkg, ctx, rank, world, dist)
            if not ok:
                ctx.close()
                return {"ok": False, "plane": plane, "error": why[:300]}
    comm_nranks = ctx.get_option("comm_nranks")
    start, length = D.shard_range(n, rank, world)
    nl = length.bit_length() - 1
    a, b = syn.tables(ctx, nl, start=start)
    g = mm.G(a, b)
    assert g.num_vars() == n

    barrier()                                 # the ranks enter every proof together (peer_spin_ms bounds their skew)
    for _ in range(args.warmup):              # exactly W untimed steps
        mm.prove(ctx, g, syn.SEED_R)

    # HIP events around every kernel (on the library's stream) on every fourth timed step: the event
    # records cost ~25 us per proof, so sampling keeps the probe from moving `value` by more than
    # ~0.3 %; the sampled launches are inside the timed region
    timed_every = 4
    ctx.kernel_time(reset=True)
    ctx.launch_log(reset=True)
    step_ms = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps_with_events = 0
    for i in range(args.steps):
        sample = (i % timed_every) == 0
        if sample:
            ctx.set_option("time_kernels", 1)
            steps_with_events += 1
        ts = time.perf_counter()
        c1, evals, ch = mm.prove(ctx, g, syn.SEED_R)     # synchronous: returns after the last round's sums
        step_ms.append((time.perf_counter() - ts) * 1e3)
        if sample:
            ctx.set_option("time_kernels", 0)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = ctx.kernel_time(reset=True)
    log = ctx.launch_log(reset=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness gate (outside the timed region) ----------------------------------------
    final_eval = g.evaluate([int(x) for x in ch])
    problem = check_identities(F, c1, evals, ch, final_eval)
    if problem:
        raise SystemExit("PARITY FAILURE at n=%d (%s): %s" % (n, label, problem))
    return {"ok": True, "plane": plane, "label": label, "ctx": ctx, "tables": (a, b, g), "elapsed": elapsed, "step_ms": step_ms,
            "n_launch": n_launch, "kernel_ms": kernel_ms, "log": log, "steps_with_events": steps_with_events,
            "timed_every": timed_every, "comm_nranks": comm_nranks, "transcript": (c1, evals.tobytes())}


def run_prover(args, pkg, torch, dist, rank, world, local_rank):
    import numpy as np
    mm, syn = pkg.matrix_multiplication, pkg.synthetic
    n = args.num_vars if args.num_vars > 0 else 28
    F = pkg.Field(pkg.GOLDILOCKS)

    runs, failed = [], {}
    for plane in planes_to_time(world):
        r = time_plane(args, pkg, torch, dist, rank, world, local_rank, plane, n)
        if not r["ok"]:
            failed[plane] = r["error"]
            if rank == 0:
                sys.stderr.write("bench.py: data plane '%s' is not usable here: %s\n" % (plane, r["error"]))
            continue
        if runs:                      # one plane's tables at a time in HBM
            if r["transcript"] != runs[0]["transcript"]:
                raise SystemExit("PARITY FAILURE: the %s and %s data planes produced different transcripts" % (runs[0]["plane"], plane))
            if r["elapsed"] < runs[0]["elapsed"]:
                runs[0], r = r, runs[0]
            r.pop("tables")
            r.pop("ctx").close()
        runs.append(r)
    if not runs:
        # a scaling run must never silently measure something else: no in-library data plane, no line
        raise SystemExit("bench.py: no data-plane transport could be set up on every rank (%s); set SC_BENCH_TRANSPORT=host "
                         "to run over the host (gloo) transport on purpose" % "; ".join("%s: %s" % kv for kv in failed.items()))
    best = runs[0]
    ctx = best["ctx"]
    a, b, g = best.pop("tables")
    elapsed, step_ms, log = best["elapsed"], best["step_ms"], best["log"]
    n_launch, kernel_ms, steps_with_events, timed_every = best["n_launch"], best["kernel_ms"], best["steps_with_events"], best["timed_every"]
    transport = best["label"]
    transports = {r["plane"]: {"ms_per_step": r["elapsed"] / args.steps * 1e3, "ms_per_step_median": statistics.median(r["step_ms"]),
                               "comm_nranks": r["comm_nranks"]} for r in runs}
    for plane, why in failed.items():
        transports[plane] = {"ms_per_step": None, "error": why}

    muladds = 5 * 2**n - 7
    alg_bytes = 64 * 2**n - 96
    value = muladds * args.steps / elapsed
    kernels = aggregate_launches(log, steps_with_events)
    per_step = len(log) // max(steps_with_events, 1)
    schedule = [[r["kind"], r["kf"], r["ks"], r["log_in"]] for r in log[:per_step]]    # the launches of one proof, in order
    kernel_ms_per_step = kernel_ms / max(steps_with_events, 1)
    moved = sum(k["bytes_per_launch"] * k["launches_per_step"] for k in kernels)    # this rank's launches
    ms_per_step = elapsed / args.steps * 1e3
    median_ms = statistics.median(step_ms)
    unsampled = [m for i, m in enumerate(step_ms) if i % timed_every]

    result = None
    if rank == 0:
        dom = kernels[0] if kernels else None
        first_pass = next((r["ks"] for r in log if r["kind"] == "pass" and r["kf"] == 0), 0)
        tkey = "n%d_gpus%d_vpp%d_first%d" % (n, world, args.vars_per_pass, first_pass)
        tj = load_traffic(tkey)
        traffic, traffic_step, traffic_check = None, None, "no PMC record for %s in profiles/traffic.json" % tkey
        if tj:
            traffic_step = tj.get("hbm_bytes_per_step")
            per_kernel = tj.get("kernels", {})
            if dom and dom["kernel"] in per_kernel:
                traffic = per_kernel[dom["kernel"]]["hbm_bytes_per_launch"]
            if traffic_step and moved:
                dev = abs(moved - traffic_step) / traffic_step
                traffic_check = ("ok: launches of this run move %.5g B/step, PMC %.5g (%.2f %% apart)" % (moved, traffic_step, dev * 100)
                                 if dev <= 0.02 else
                                 "MISMATCH: launches of this run move %.5g B/step but profiles/traffic.json holds %.5g" % (moved, traffic_step))
                if dev > 0.02:
                    traffic, traffic_step = None, None
                    sys.stderr.write("bench.py: " + traffic_check + "\n")
        result = {
            "metric": "field mul-adds/sec in sumcheck prover, n=%d vars" % n,
            "value": value,
            "unit": "field mul-adds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_median": median_ms,
            "value_at_median": muladds / (median_ms * 1e-3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "full sumcheck prover, g=a*b, n=%d, Goldilocks p=2^64-2^32+1, hypercube sharded by top "
                            "index bits over %d GPU(s), %s" % (
                                n, world, ("%s exchange per pass" % transport) if world > 1 else "no collective"),
                "num_vars": n,
                "field_mul_adds_per_step": muladds,
                "algorithmic_bytes_per_step": alg_bytes,
                "vars_per_pass": args.vars_per_pass,
                "first_pass_vars": first_pass,
                "parallelism": "hypercube-shard x%d" % world,
                "parity_gate": "verifier identities at n=%d ok" % n + ("; every timed data plane gave the same transcript" if len(runs) > 1 else ""),
                "transport": transport,
                # every data plane this run timed (its own barrier-bracketed region of exactly `steps` proofs; the headline is
                # the fastest); comm_nranks = the ranks the plane spans as the transport reports it (ncclCommCount for RCCL)
                "transports": transports,
                "options": dict(context_options()), "schedule": schedule,
                "ms_per_step_median_unsampled": statistics.median(unsampled) if unsampled else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom["kernel"] if dom else None,
                "achieved": dom["GBps"] if dom else None,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": dom["GBps"] / HBM_PEAK_GBS if dom and dom["GBps"] else None,
                "traffic": traffic,
                "bytes_per_launch": dom["bytes_per_launch"] if dom else None,
                "avg_launch_us": dom["avg_us"] if dom else None,
                "per_gpu": True,
                "step": {
                    "bytes_moved": moved,
                    "kernel_ms": kernel_ms_per_step,
                    "launches": n_launch / max(steps_with_events, 1),
                    "frac_of_kernel_time": moved / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms_per_step else None,
                    "frac_of_wall_time": moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "pmc_traffic": traffic_step,
                    "traffic_check": traffic_check,
                    "sec8d_algorithmic_bytes": alg_bytes / world,
                    "sec8d_credited_frac": (alg_bytes / world) / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms_per_step else None,
                },
                "kernels": [{k: v for k, v in kk.items() if k != "_key"} for kk in kernels],
                "steps_sampled": steps_with_events,
                "note": "achieved/frac: the dominant kernel's HBM bytes per launch (inputs read once + outputs written "
                        "once, from the launch log of this run) / its mean HIP-event duration, on ONE GPU (rank 0's launches over "
                        "its own shard: a per-GPU fraction of the per-GPU peak). step.*: all launches of "
                        "a proof on that GPU. sec8d_credited_frac divides SURVEY 8d's one-round-per-pass byte model (64*2^n-96) by "
                        "kernel time; the multi-round schedule moves fewer bytes, so that figure can exceed 1 and is "
                        "not a roofline fraction.",
            },
        }

    # ---- CPU baseline: the reference-shaped port, 1 core, bounded sample (N = 1 only) -------
    nc = cpu_sample_size(args.cpu_num_vars)
    if rank == 0 and world == 1 and nc > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle   # the checker / CPU baseline: only this leg touches oracle/
        import pyref
        assert (pyref.SEED_A, pyref.SEED_B, pyref.SEED_R) == (syn.SEED_A, syn.SEED_B, syn.SEED_R)
        o = Oracle(pkg.GOLDILOCKS)
        oa, ob = o.generate(pyref.SEED_A, nc), o.generate(pyref.SEED_B, nc)
        och = np.array([o.challenge(pyref.SEED_R, j + 1) for j in range(nc)], dtype=np.uint64)
        tc = time.perf_counter()
        c1_cpu, ev_cpu = o.prover_run(oa, ob, och)
        cpu_s = time.perf_counter() - tc
        tc = time.perf_counter()
        c1_mt, ev_mt = o.prover_run_mt(oa, ob, och)           # second, clearly labelled row: all host cores
        cpu_mt_s = time.perf_counter() - tc
        if c1_mt != c1_cpu or not np.array_equal(ev_mt, ev_cpu):
            raise SystemExit("oracle: multi-threaded and single-threaded runs disagree")
        # the same sample through the GPU path must agree bit for bit
        del a, b, g
        ga, gb = syn.tables(ctx, nc)
        c1_gpu, ev_gpu, ch_gpu = mm.prove(ctx, mm.G(ga, gb), syn.SEED_R)
        if c1_gpu != c1_cpu or not np.array_equal(ev_gpu, ev_cpu) or not np.array_equal(ch_gpu, och):
            raise SystemExit("PARITY FAILURE: GPU and CPU oracle disagree at n=%d" % nc)
        result["config"]["parity_gate"] += "; bit-exact vs CPU oracle at n=%d ok" % nc
        result["cpu_baseline"] = {
            "value": (5 * 2**nc - 7) / cpu_s,
            "unit": "field mul-adds/s",
            "cores": 1,
            "host_cores_total": os.cpu_count(),
            "kind": "port",
            "sample": "same synthetic workload at n=%d (%d mul-adds, %.1f s): oracle/sc_oracle.c sco_prover_run, the "
                      "reference-shaped single-thread C restatement (clone + multiply + sum for c_1, copy-fold-copy "
                      "per table per round, separate sum pass); the Rust reference cannot be built here" % (
                          nc, 5 * 2**nc - 7, cpu_s),
            "all_cores": {"value": (5 * 2**nc - 7) / cpu_mt_s, "cores": os.cpu_count(), "seconds": cpu_mt_s,
                          "note": "same port with the element loops split over OpenMP threads (BASELINE.md CPU-ref-allT)"},
        }
    elif rank == 0:
        result["cpu_baseline"] = None
    return result


def run_mle(args, pkg, torch, dist, rank, world, local_rank):
    """BASELINE config 2: evaluate + fix_variable on one 2^n-entry table (single GPU)."""
    import numpy as np
    if world != 1:
        raise SystemExit("--workload mle is a single-GPU workload (run --gpus N as N replicas by hand)")
    syn = pkg.synthetic
    n = args.num_vars if args.num_vars > 0 else 24
    F = pkg.Field(pkg.GOLDILOCKS)
    ctx = pkg.Context(F, device=local_rank)
    t = pkg.DenseMultilinearExtension.generate(ctx, syn.SEED_A, n)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle    # challenge derivation + the checker (after the timed region)
    o = Oracle(pkg.GOLDILOCKS)
    pt = [int(o.challenge(syn.SEED_PT, j)) for j in range(n)]
    ks = [1, 3, n // 2]
    ops = [("evaluate_le", n), ("evaluate_be", n)] + [("fix_variables_k%d" % k, k) for k in ks]

    def step():
        outs = [t.evaluate(pt, pkg.ORDER_LE), t.evaluate(pt, pkg.ORDER_BE)]
        outs += [t.fix_variables(pt[:k]) for k in ks]
        ctx.synchronize()          # fix_variables returns once its launches are in the library's stream
        return outs

    ctx.set_option("time_kernels", 1)     # five launches per step, each hundreds of microseconds: probe cost is negligible
    for _ in range(args.warmup):
        step()
    ctx.launch_log(reset=True)
    ctx.kernel_time(reset=True)
    step_ms = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        outs = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.set_option("time_kernels", 0)
    n_launch, kernel_ms = ctx.kernel_time(reset=True)
    log = ctx.launch_log(reset=True)
    if os.environ.get("SC_BENCH_DEBUG"):
        sys.stderr.write("step_ms: %s\n" % " ".join("%.3f" % x for x in step_ms))

    muladds = 2 * (2**n - 1) + sum(2**n - 2**(n - k) for k in ks)
    alg_bytes = 2 * 8 * 2**n + sum(8 * 2**n + 8 * 2**(n - k) for k in ks)
    kernels = aggregate_launches(log, args.steps)
    moved = sum(k["bytes_per_launch"] * k["launches_per_step"] for k in kernels)
    kernel_ms_per_step = kernel_ms / args.steps
    ms_per_step = elapsed / args.steps * 1e3
    dom = next((k for k in kernels if k["_key"][0] == "evaluate"), kernels[0])

    # parity (outside the timed region): oracle at sizes it finishes in seconds, properties above
    nchk = min(n, 24)
    parity = []
    tc = pkg.DenseMultilinearExtension.generate(ctx, syn.SEED_A, nchk)
    ot = o.generate(syn.SEED_A, nchk)
    ptc = pt[:nchk]
    optc = np.array(ptc, dtype=np.uint64)
    if tc.evaluate(ptc, pkg.ORDER_LE) != o.evaluate(ot, optc):
        raise SystemExit("PARITY FAILURE: evaluate (LE) differs from the oracle at n=%d" % nchk)
    if tc.evaluate(ptc, pkg.ORDER_BE) != o.vsbw(ot, optc):
        raise SystemExit("PARITY FAILURE: evaluate (BE) differs from vsbw_multilinear_from_evaluations at n=%d" % nchk)
    for k in (1, 3, nchk // 2):
        if not np.array_equal(tc.fix_variables(ptc[:k]).to_evaluations(), o.fix_variables(ot, optc[:k])):
            raise SystemExit("PARITY FAILURE: fix_variables k=%d differs from the oracle at n=%d" % (k, nchk))
    parity.append("evaluate LE/BE + fix_variables k=1,3,%d bit-exact vs CPU oracle at n=%d" % (nchk // 2, nchk))
    # size-independent property at the benchmarked size: fixing k variables then evaluating the rest
    # equals the full evaluate
    for k, folded in zip(ks, outs[2:]):
        if folded.evaluate(pt[k:], pkg.ORDER_LE) != outs[0]:
            raise SystemExit("PARITY FAILURE: fix_variables(k=%d) then evaluate != evaluate at n=%d" % (k, n))
    parity.append("fix(k) o evaluate == evaluate at n=%d" % n)

    tkey = "mle_n%d" % n
    tj = load_traffic(tkey)
    traffic = None
    if tj and dom["kernel"] in tj.get("kernels", {}):
        traffic = tj["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
    result = {
        "metric": "field mul-adds/sec, multilinear-extensions evaluate + fix_variable, n=%d" % n,
        "value": muladds * args.steps / elapsed,
        "unit": "field mul-adds/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_median": statistics.median(step_ms),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "multilinear-extensions evaluate (LE and BE) + fix_variables k=1,3,n/2 on one 2^%d-entry Goldilocks "
                        "table (BASELINE configs[1])" % n,
            "num_vars": n,
            "ops": [name for name, _ in ops],
            "field_mul_adds_per_step": muladds,
            "algorithmic_bytes_per_step": alg_bytes,
            "parity_gate": "; ".join(parity),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dom["kernel"],
            "achieved": dom["GBps"],
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": dom["GBps"] / HBM_PEAK_GBS if dom["GBps"] else None,
            "traffic": traffic,
            "bytes_per_launch": dom["bytes_per_launch"],
            "avg_launch_us": dom["avg_us"],
            "step": {"bytes_moved": moved, "kernel_ms": kernel_ms_per_step, "launches": n_launch / args.steps,
                     "frac_of_kernel_time": moved / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms_per_step else None,
                     "frac_of_wall_time": moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "sec8d_algorithmic_bytes": alg_bytes},
            "kernels": [{k: v for k, v in kk.items() if k != "_key"} for kk in kernels],
            "note": "algorithmic bytes of every op = one read of the table + its output (SURVEY 8d, config 2); each op is "
                    "one launch, so bytes moved = algorithmic bytes",
        },
    }
    # CPU baseline: the oracle's reference-shaped single-thread evaluate / fix on a bounded sample
    nc = min(n, 24)
    tc0 = time.perf_counter()
    o.evaluate(ot[: 1 << nc], optc[:nc])
    o.vsbw(ot[: 1 << nc], optc[:nc])
    for k in (1, 3, nc // 2):
        o.fix_variables(ot[: 1 << nc], optc[:k])
    cpu_s = time.perf_counter() - tc0
    cpu_muladds = 2 * (2**nc - 1) + sum(2**nc - 2**(nc - k) for k in (1, 3, nc // 2))
    result["cpu_baseline"] = {
        "value": cpu_muladds / cpu_s, "unit": "field mul-adds/s", "cores": 1, "host_cores_total": os.cpu_count(),
        "kind": "port",
        "sample": "the same five operations at n=%d (%.1f s): oracle/sc_oracle.c, reference-shaped single-thread C "
                  "restatement of DenseMultilinearExtension::fix_variables / evaluate and vsbw_multilinear_from_evaluations" % (nc, cpu_s),
    }
    return result


def self_launch(n_ranks):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run ... bench.py <same flags>` as a
    child process (one rank per GPU), relay its stdout and return its exit code"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")      # what the launcher would set itself, without its warning on stderr
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:                    # stderr is inherited
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["prover", "mle"], default=os.environ.get("SC_BENCH_WORKLOAD", "prover"))
    ap.add_argument("--num-vars", type=int, default=int(os.environ.get("SC_BENCH_N", "0")),
                    help="0 = the workload's BASELINE size (prover 28, mle 24)")
    ap.add_argument("--cpu-num-vars", type=int, default=int(os.environ.get("SC_BENCH_CPU_N", "-1")),
                    help="size of the bounded CPU-baseline sample (0 disables; -1 = as large as host memory allows, <= 28)")
    ap.add_argument("--vars-per-pass", type=int, default=2)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N rank processes ourselves.  This process has not touched the GPU
        # (torch is not even imported yet) and never will: the ranks run in a CHILD launcher, whose stdout (rank 0's one
        # JSON line) and exit code are relayed.
        return self_launch(args.gpus)

    import torch
    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if os.environ.get("SC_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0  # diagnostic: several ranks share GPU 0
    torch.cuda.set_device(local_rank)

    pkg = ge.load_package()
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)  # control plane only

    run = run_prover if args.workload == "prover" else run_mle
    result = run(args, pkg, torch, dist, rank, world, local_rank)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
