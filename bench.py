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

--workload gkr | gnew | triangle.  The callers either side of the path (BASELINE configs[4] names gkr and the
matrix-multiplication prover): one GKR layer's W sumcheck at k = 13, matrix_multiplication::G::new at n = 14, the
triangle-counting prover at k = 10 - each with its own roofline object from the launch log, a parity gate inside the run
and the oracle's reference-shaped CPU run of a bounded sample as cpu_baseline.

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
import re
import statistics
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the pool's driver only supports dmabuf IPC (RCCL)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ABANDONED = []   # contexts whose stream may be stuck in a collective that never completes: never destroyed; the process
                 # leaves through os._exit once rank 0's line is out
FIELD = {"p": None, "name": "GoldilocksMont", "label": "Goldilocks p=2^64-2^32+1"}   # set by main() from --field
GENERIC_P = 2**64 - 59   # --field generic: the largest 64-bit prime, through the kernels every modulus but Goldilocks' takes
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
INT8_PEAK_TOPS = 5000.0  # dense int8 matrix-core peak: 2 x the ~2.5 PF dense bf16 figure (MI355X_MICROARCH.md, matrix cores)


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
    return _kernel_name(rec).replace("GoldilocksMont", FIELD["name"])      # the field policy the kernels were instantiated with


def _kernel_name(rec):
    k = rec["kind"]
    if k == "pass":
        return "sc::pass_kernel<GoldilocksMont,%d,%d> on 2^%d-entry tables" % (rec["kf"], rec["ks"], rec["log_in"])
    if k == "grid_pass":
        return "sc::wgrid_pass_kernel<GoldilocksMont,ks> (kf=%d, ks=%d) on 2^%d-entry tables" % (rec["kf"], rec["ks"], rec["log_in"])
    if k == "wfold_pass":
        return "sc::wfold_pass_kernel<GoldilocksMont,%d,%d> on 2^%d-entry tables" % (rec["kf"], rec["ks"], rec["log_in"])
    if k == "gram_pass":      # field-agnostic: exact integer limb products on the int8 matrix cores (kernels/gram.hpp)
        return "sc::gram_pass_kernel<%d> (rounds 1..%d from one read) on 2^%d-entry tables" % (rec["ks"], rec["ks"], rec["log_in"])
    if k == "gram_finish":
        return "sc::gram_finish_kernel<GoldilocksMont,%d> (partials -> %d cells) behind the 2^%d-entry pass" % (rec["ks"], 3 ** rec["ks"], rec["log_in"])
    if k == "evaluate":
        return "sc::evaluate_kernel<GoldilocksMont> on a 2^%d-entry table" % rec["log_in"]
    if k == "fold":
        return "sc::fold_kernel<GoldilocksMont,%d> on a 2^%d-entry table" % (rec["kf"], rec["log_in"])
    if k == "fix_low":
        return "sc::fix_low_kernel<GoldilocksMont> (%d variables) on a 2^%d-entry table" % (rec["kf"], rec["log_in"])
    if k == "gkr":
        return "sc::gkr_phase1_kernel<GoldilocksMont> (P and L over 2^%d rows of c) on 2^%d-entry add/mul tables" % (rec["kf"], rec["log_in"])
    if k == "coldot":
        return "sc::coldot_kernel<GoldilocksMont> (2^%d rows) on a 2^%d-entry table" % (rec["kf"], rec["log_in"])
    if k == "matsq":
        return "sc::matsq_tiled_kernel<GoldilocksMont> on a 2^%d-entry adjacency table" % rec["log_in"]
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


PROVING_FAILED = []   # data planes that came up and then failed while proving: the run still prints its line, and exits non-zero


def all_agree(torch, dist, ok):
    """True iff `ok` on every rank (one control-plane all-reduce that EVERY rank executes, whatever happened locally)"""
    if dist is None:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


def load_traffic(key):
    """the PMC record of one workload from profiles/traffic.json (the builder's rocprofv3 --pmc runs; this run cannot collect its
    own: the profiler has to wrap the program from its start) - or None when the record was measured on other kernel sources than
    the ones in this tree (tools/provenance.py: a fingerprint of thaler-study_amd/csrc stored with every record)"""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        rec = json.load(f).get(key)
    if rec is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import provenance
    prov = rec.get("provenance") or {}
    now = provenance.csrc_fingerprint()
    rec["_stale"] = None if prov.get("csrc_sha16") == now else (
        "profiles/traffic.json[%s] was measured on kernel sources %s (%s), this tree has %s: PMC bytes withheld until "
        "tools/refresh_profiles.sh is run again" % (key, prov.get("csrc_sha16") or "without a fingerprint", prov.get("tag") or rec.get("source", "?")[:3], now))
    return rec


def traffic_source(rec):
    """`roofline.traffic_source`: which run the PMC bytes belong to"""
    prov = rec.get("provenance") or {}
    return ("profiles/traffic.json: `%s` on the builder's GPU box, %s, commit %s, kernel sources %s (= this tree's); FETCH_SIZE x 2 KiB + "
            "WRITE_SIZE x 1 KiB per MI355X_MICROARCH.md; not collected in this run, handed out only if this run's launch-log bytes agree "
            "within 2 %%" % (prov.get("command"), prov.get("date"), prov.get("commit") or "not recorded", prov.get("csrc_sha16")))


def self_pmc(args, n, schedule):
    """FIRST-HAND HBM bytes for `roofline.traffic`: after the timed region, two child runs of this very workload under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md's HBM section
    prescribes; the program itself after `--`), on THIS box - a profiler has to wrap a program from its start, so it is a child
    process (started, not exec'ed: this process holds the GPU).  Returns (bytes per launch of one proof in schedule order, what was
    run) or (None, why not).  Off: SC_BENCH_SELF_PMC=0; by default only for the headline shape (n = 28, one GPU)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not rocprof:
        return None, "rocprofv3 not found on this box"
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SC_BENCH_SELF_PMC="0", SC_BENCH_RAMP_MS="0", SC_BENCH_SECONDARY="0", TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-num-vars", "0", "--num-vars", str(n),
             "--vars-per-pass", str(args.vars_per_pass), "--field", args.field]
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="sc_pmc_", dir="/tmp")
        try:
            r = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=float(os.environ.get("SC_BENCH_SELF_PMC_TIMEOUT", "240")))
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "the %s pass failed (rc %d): %s" % (counter, r.returncode, (r.stderr or r.stdout)[-200:])
            rows = [x for x in csv.DictReader(open(max(files, key=os.path.getmtime))) if x["Counter_Name"] == counter and "pass_kernel<" in x["Kernel_Name"]]
        except Exception as e:      # a profiler that does not start or does not end: the record in profiles/ is used instead
            return None, "the %s pass: %s" % (counter, str(e)[:200])
        finally:
            shutil.rmtree(out, ignore_errors=True)
        # the launches of the LAST proof of the child's trace: a proof starts with the matrix-core pass or a pass that folds nothing
        steps, cur = [], []
        for x in rows:
            name = x["Kernel_Name"]
            m = re.search(r"[^_]pass_kernel<sc::\w+, (\d), (\d)", " " + name)
            first = "gram_pass_kernel<" in name or (m is not None and m.group(1) == "0" and "wgrid" not in name and "wfold" not in name)
            if first and cur:
                steps.append(cur)
                cur = []
            cur.append(float(x["Counter_Value"]))
        if cur:
            steps.append(cur)
        if not steps or len(steps[-1]) != len(schedule):
            return None, "the %s pass shows %d launches in its last proof, this run's schedule has %d" % (counter, len(steps[-1]) if steps else 0, len(schedule))
        vals[counter] = steps[-1]
    per_launch = [f * 2 * 1024 + w * 1024 for f, w in zip(vals["FETCH_SIZE"], vals["WRITE_SIZE"])]
    return per_launch, "`rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py %s`" % " ".join(child[2:])


def reference_probe():
    """could the reference itself (Rust) be timed beside the GPU on this box?  It needs cargo, an
    offline registry holding the arkworks crates, and the reference's sources - which bench.py may not read at run time on the
    GPU box (they are not there).  Returns what was found; the CPU baseline stays the C port unless all three exist."""
    import glob
    import shutil
    cargo = shutil.which("cargo")
    home = os.environ.get("CARGO_HOME") or os.path.expanduser("~/.cargo")
    crates = glob.glob(os.path.join(home, "registry", "src", "*", "ark-poly-*")) + glob.glob(os.path.join(ROOT, "rust", "vendor", "ark-poly*"))
    src = os.environ.get("SC_REFERENCE_DIR")
    have_src = bool(src) and os.path.isfile(os.path.join(src, "matrix-multiplication", "Cargo.toml"))
    out = {"cargo": cargo or "absent", "rustc": shutil.which("rustc") or "absent", "offline_ark_crates": bool(crates),
           "reference_sources": src if have_src else "absent (SC_REFERENCE_DIR unset; never on the GPU box)",
           "usable": bool(cargo) and bool(crates) and have_src}
    if out["usable"] and os.environ.get("SC_BENCH_REFERENCE", "1") == "1":
        # all three exist (never on this pool): the reference's OWN criterion bench (matrix-multiplication/benches/mm_benchmark.rs:
        # Prover::new + rounds over G for num_vars 2..15), run from its sources where they lie, bounded; its report is recorded as it
        # comes.  It is the reference at ITS sizes - G's tables are private, no public constructor takes two 2^28-entry tables - so the
        # n = 28 row of cpu_baseline stays the port either way.
        import subprocess
        try:
            r = subprocess.run([cargo, "bench", "--offline", "--manifest-path", os.path.join(src, "Cargo.toml"), "-p", "matrix-multiplication"],
                               capture_output=True, text=True, timeout=float(os.environ.get("SC_BENCH_REFERENCE_TIMEOUT", "600")))
            out["criterion"] = {"returncode": r.returncode, "report_tail": [l for l in r.stdout.splitlines() if "time:" in l or "thrpt:" in l][-28:]}
        except Exception as e:      # a bench that does not build or does not end is a finding, not a failure of this run
            out["criterion"] = {"error": str(e)[:300]}
    return out


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
        # ncclCommInitRank blocks until every rank has arrived and has been seen to hang on broken fabrics: it runs in a
        # thread this rank can walk away from (the run then ends through os._exit, see ABANDONED)
        import threading
        box2 = {}

        def init():
            try:
                if os.environ.get("SC_BENCH_TEST_RCCL_HANG") == "1":      # test hook: an init that never returns
                    time.sleep(3600)
                ctx.comm_init_rccl(box[0], rank, world)
                box2["ok"] = True
            except Exception as e:  # pragma: no cover - e.g. two ranks on one device: RCCL refuses duplicate GPUs
                box2["err"] = str(e)

        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(float(os.environ.get("SC_BENCH_RCCL_INIT_TIMEOUT", "180")))
        if th.is_alive():
            ok, err = False, "ncclCommInitRank did not return within its time limit"
            ABANDONED.append(ctx)
        elif "err" in box2:
            ok, err = False, box2["err"]
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
        if want not in ("peer", "rccl", "host", "inproc"):
            raise SystemExit("SC_BENCH_TRANSPORT must be peer, rccl, host or inproc")
        return [want]
    # "inproc": ONE process (rank 0) drives all N devices through one multi-device handle (sc_ctx_create_multi) - the form the
    # reference's single-process caller uses; the other ranks wait at the barriers of its timed region
    return ["peer", "rccl", "inproc"]


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
    {"ok": False, "error": ...} when the plane could not be set up on every rank or failed while proving on any.
    Every rank runs the same sequence of control-plane collectives whatever happens to it locally: a failure is
    caught where it happens, the rank keeps walking through the barriers, and the ranks agree on the outcome at the end.
    plane "inproc": rank 0 alone proves, over ALL `world` devices through one multi-device handle; the other ranks only take
    part in the barriers, so the region is timed exactly like the others."""
    mm, D, syn = pkg.matrix_multiplication, pkg.distributed, pkg.synthetic
    F = pkg.Field(FIELD["p"])
    inproc = plane == "inproc"
    working = (rank == 0) or not inproc        # does this rank launch anything on this plane?

    def barrier():
        if dist is not None:
            dist.barrier()

    ctx, label, comm_nranks, err = None, plane, world, None
    a = b = g = None
    try:
        if inproc and os.environ.get("SC_BENCH_FAIL_TRANSPORT_RANK") in (str(rank), "all"):
            raise pkg.SumcheckHipError(2, "injected transport failure (SC_BENCH_FAIL_TRANSPORT_RANK)")   # test hook
        if inproc:
            if working:
                ndev = 1 if os.environ.get("SC_BENCH_SINGLE_DEVICE") == "1" else max(torch.cuda.device_count(), 1)
                devices = [d % ndev for d in range(world)]
                ctx = pkg.Context(F, devices=devices)
                label = "inproc (one process, one handle over devices %s)" % devices
        else:
            ctx = pkg.Context(F, device=local_rank)
        if ctx is not None:
            ctx.set_option("vars_per_pass", args.vars_per_pass)
            for k, v in context_options():
                ctx.set_option(k, v)
    except pkg.SumcheckHipError as e:
        err = "context: %s" % e
    if not all_agree(torch, dist, err is None):
        if ctx is not None:
            ctx.close()
        return {"ok": False, "plane": plane, "error": (err or "failed on another rank")[:300]}
    if plane not in ("none", "inproc"):
        if world == 1:
            (D.attach_peer if plane == "peer" else D.attach_rccl)(ctx, 0, 1)
            label = "%s(world=1, diagnostic)" % plane
        else:
            ok, why = attach_plane(plane, pkg, ctx, rank, world, dist)
            if not ok:
                if ctx not in ABANDONED:
                    ctx.close()
                return {"ok": False, "plane": plane, "error": why[:300]}
    step_ms, steps_with_events, timed_every = [], 0, 4
    c1 = evals = ch = None
    try:
        if working:
            comm_nranks = ctx.get_option("comm_nranks")
            if inproc:
                a, b = syn.tables(ctx, n)                  # whole tables: the handle splits them over its devices
            else:
                start, length = D.shard_range(n, rank, world)
                a, b = syn.tables(ctx, length.bit_length() - 1, start=start)
            g = mm.G(a, b)
            assert g.num_vars() == n
    except pkg.SumcheckHipError as e:
        err = "tables: %s" % e

    barrier()                                 # the ranks enter every proof together (peer_spin_ms bounds their skew)
    # ---- setup, untimed: bring the device to its steady clocks -------------------------------------------------------------
    # An MI355X that has idled for as little as 50 ms runs its first ~10-20 proofs below its steady clocks: the VALU-bound first
    # pass takes 770-815 us instead of 665-680 and a proof 1.92-2.00 ms instead of 1.81 (profiles/r04_clock_ramp.txt).  A run of
    # W = 5 + K = 20 proofs (46 ms) is over before the ramp is, so its number was a property of the power state the run began
    # in, not of the kernels.  Like the reference's own harness (criterion warms up for 3 s before it samples:
    # matrix-multiplication/benches/mm_benchmark.rs), the bench first keeps the device busy for SC_BENCH_RAMP_MS (default 80 ms)
    # with the workload itself - setup, before the W warm-up steps, reported in config.clock_ramp with the COLD figure beside
    # it.  Every rank runs the same number of proofs (the count is agreed through the control plane).
    ramp = None
    ramp_ms = float(os.environ.get("SC_BENCH_RAMP_MS", "80"))
    # the FIRST proof of this process on this context (a caller of the reference's loop proves once,
    # mm_benchmark.rs:88-96, and sees this number, not the steady one): with the context prewarmed for this size (option "prewarm":
    # code object on the device, pool blocks, resident-grid queries - setup, like building the tables) unless SC_BENCH_PREWARM=0
    first_proof = {"prewarm": os.environ.get("SC_BENCH_PREWARM", "1") == "1", "ms": None}
    if working and err is None:
        try:
            if first_proof["prewarm"]:
                tp = time.perf_counter()
                ctx.set_option("prewarm", n)
                first_proof["prewarm_ms"] = (time.perf_counter() - tp) * 1e3
            ctx.synchronize()
            tr = time.perf_counter()
            mm.prove(ctx, g, syn.SEED_R)
            first_proof["ms"] = (time.perf_counter() - tr) * 1e3
        except pkg.SumcheckHipError as e:
            err = "failed while proving (first proof): %s" % e
    if ramp_ms > 0:
        cold = 0.0
        if working and err is None:
            try:
                tr = time.perf_counter()
                mm.prove(ctx, g, syn.SEED_R)
                mm.prove(ctx, g, syn.SEED_R)
                cold = (time.perf_counter() - tr) / 2 * 1e3
            except pkg.SumcheckHipError as e:
                err = "failed while proving (clock ramp): %s" % e
        est = cold
        if dist is not None:
            t = torch.tensor([est], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            est = float(t.item())
        count = min(4000, int(ramp_ms / est) + 1) if est > 0 else 0
        if working and err is None:
            try:
                for _ in range(count):
                    mm.prove(ctx, g, syn.SEED_R)
            except pkg.SumcheckHipError as e:
                err = "failed while proving (clock ramp): %s" % e
        ramp = {"untimed_ms": ramp_ms, "proofs": count + 2, "cold_ms_per_proof": cold, "first_proof_ms": first_proof["ms"],
                "first_proof": first_proof,
                "note": "setup before the W warm-up steps: the workload itself, run until the device is at its steady clocks "
                        "(SC_BENCH_RAMP_MS=0 switches it off); first_proof_ms = the very first proof of this process on its fresh "
                        "context (prewarmed for this size unless SC_BENCH_PREWARM=0); cold_ms_per_proof = the two proofs after it, "
                        "still below the steady clocks"}
        barrier()
    if working and err is None:
        try:
            for _ in range(args.warmup):      # exactly W untimed steps
                mm.prove(ctx, g, syn.SEED_R)
            # HIP events around every kernel (on the library's stream) on every fourth timed step: the event
            # records cost ~25 us per proof, so sampling keeps the probe from moving `value` by more than
            # ~0.3 %; the sampled launches are inside the timed region
            ctx.kernel_time(reset=True)
            ctx.launch_log(reset=True)
        except pkg.SumcheckHipError as e:
            err = "failed while proving (warm-up): %s" % e
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if working and err is None:
        try:
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
        except pkg.SumcheckHipError as e:
            err = "failed while proving: %s" % e
    try:
        torch.cuda.synchronize()
    except RuntimeError as e:  # pragma: no cover - a faulted device
        err = err or "device synchronize: %s" % e
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not all_agree(torch, dist, err is None):
        # a plane that came up but failed while proving (a collective that never completed, a peer that fell out of step): the
        # other planes' measurements survive it; its context may hold a stream that never drains and is not destroyed
        if ctx is not None:
            ABANDONED.append(ctx)
        PROVING_FAILED.append(plane)
        return {"ok": False, "plane": plane, "error": (err or "failed while proving on another rank")[:300]}
    if not working:
        return {"ok": True, "plane": plane, "label": label, "ctx": None, "tables": None, "elapsed": elapsed, "step_ms": [], "n_launch": 0,
                "kernel_ms": 0.0, "log": [], "steps_with_events": 0, "timed_every": timed_every, "comm_nranks": world, "transcript": None}
    n_launch, kernel_ms = ctx.kernel_time(reset=True)
    log = ctx.launch_log(reset=True)

    # ---- correctness gate (outside the timed region) ----------------------------------------
    final_eval = g.evaluate([int(x) for x in ch])
    problem = check_identities(F, c1, evals, ch, final_eval)
    if problem:
        raise SystemExit("PARITY FAILURE at n=%d (%s): %s" % (n, label, problem))
    return {"ok": True, "plane": plane, "label": label, "ctx": ctx, "tables": (a, b, g), "elapsed": elapsed, "step_ms": step_ms,
            "n_launch": n_launch, "kernel_ms": kernel_ms, "log": log, "steps_with_events": steps_with_events,
            "timed_every": timed_every, "comm_nranks": comm_nranks, "transcript": (c1, evals.tobytes()), "ramp": ramp, "first_proof": first_proof}


def run_prover(args, pkg, torch, dist, rank, world, local_rank):
    import numpy as np
    mm, syn = pkg.matrix_multiplication, pkg.synthetic
    n = args.num_vars if args.num_vars > 0 else 28
    F = pkg.Field(FIELD["p"])

    runs, failed = [], {}
    for plane in planes_to_time(world):
        r = time_plane(args, pkg, torch, dist, rank, world, local_rank, plane, n)     # (never raises SumcheckHipError: see there)
        if not r["ok"]:
            failed[plane] = r["error"]
            if rank == 0:
                sys.stderr.write("bench.py: data plane '%s' is not usable here: %s\n" % (plane, r["error"]))
            continue
        if dist is not None and plane == "inproc":
            # only rank 0 holds this plane's transcript and launch log: the ranks compare on rank 0's word
            box = [r["transcript"]]
            dist.broadcast_object_list(box, src=0)
            if rank != 0:
                r["transcript"] = box[0]
        if runs:                      # one plane's tables at a time in HBM
            same = r["transcript"] == runs[0]["transcript"]
            if not same:
                raise SystemExit("PARITY FAILURE: the %s and %s data planes produced different transcripts" % (runs[0]["plane"], plane))
            # the headline is the fastest plane - never one whose collectives were served by a stand-in for librccl
            # (SC_RCCL_LIBRARY: tests/rccl_double on one-GPU boxes): that plane is timed and reported, not ranked
            def stand_in(pl):
                return pl == "rccl" and bool(os.environ.get("SC_RCCL_LIBRARY"))
            if (r["elapsed"] < runs[0]["elapsed"] and not stand_in(plane)) or (stand_in(runs[0]["plane"]) and not stand_in(plane)):
                runs[0], r = r, runs[0]
            r.pop("tables")
            if r["ctx"] is not None:
                r.pop("ctx").close()
        runs.append(r)
    if not runs:
        # a scaling run must never silently measure something else: no in-library data plane, no line
        raise SystemExit("bench.py: no data-plane transport could be set up on every rank (%s); set SC_BENCH_TRANSPORT=host "
                         "to run over the host (gloo) transport on purpose" % "; ".join("%s: %s" % kv for kv in failed.items()))
    best = runs[0]
    ctx = best["ctx"]
    a, b, g = best.pop("tables") or (None, None, None)     # (None on the ranks that only waited for an "inproc" plane)
    elapsed, step_ms, log = best["elapsed"], best["step_ms"], best["log"]
    n_launch, kernel_ms, steps_with_events, timed_every = best["n_launch"], best["kernel_ms"], best["steps_with_events"], best["timed_every"]
    transport = best["label"]
    transports = {r["plane"]: {"ms_per_step": r["elapsed"] / args.steps * 1e3, "ms_per_step_median": statistics.median(r["step_ms"]) if r["step_ms"] else None,
                               "comm_nranks": r["comm_nranks"]} for r in runs}
    for plane, why in failed.items():
        transports[plane] = {"ms_per_step": None, "error": why}
    if "rccl" in transports:
        # which library served the collectives of the RCCL plane: librccl, or whatever SC_RCCL_LIBRARY names (on a one-GPU box the
        # test double of tests/rccl_double - a functional run of the N > 1 control flow, not a measurement of RCCL)
        lib = os.environ.get("SC_RCCL_LIBRARY")
        transports["rccl"]["library"] = ("SC_RCCL_LIBRARY=" + lib + (" (test double: not RCCL, never the headline)" if "double" in os.path.basename(lib) else "")) if lib else "librccl"

    muladds = 5 * 2**n - 7
    alg_bytes = 64 * 2**n - 96
    value = muladds * args.steps / elapsed
    kernels = aggregate_launches(log, steps_with_events)
    per_step = len(log) // max(steps_with_events, 1)
    schedule = [[r["kind"], r["kf"], r["ks"], r["log_in"]] for r in log[:per_step]]    # the launches of one proof, in order
    kernel_ms_per_step = kernel_ms / max(steps_with_events, 1)
    moved = sum(k["bytes_per_launch"] * k["launches_per_step"] for k in kernels)    # this rank's launches
    ms_per_step = elapsed / args.steps * 1e3
    median_ms = statistics.median(step_ms) if step_ms else None
    unsampled = [m for i, m in enumerate(step_ms) if i % timed_every]
    # the first folding pass of a 2^28-entry proof has two speeds, box to box and context to context (DESIGN_HISTORY.md section 11,
    # experiments/r03_fold_pass_two_modes.md, experiments/r04_vmm_placement.md): say which one this run drew
    fold_pass_mode = None
    fp_rec = next((k for k in kernels if ((k["_key"][0] == "pass" and k["_key"][1] in (3, 4) and k["_key"][2] == 2) or
                                          (k["_key"][0] == "wfold_pass" and k["_key"][1] == 4)) and k["_key"][3] >= 27), None)
    if fp_rec and fp_rec["GBps"]:
        fold_pass_mode = {"mode": "fast" if fp_rec["GBps"] >= 5900.0 else "slow", "GBps": fp_rec["GBps"], "avg_us": fp_rec["avg_us"],
                          "note": "the first folding pass (wfold_pass_kernel<4,5> behind the four-round first pass; pass_kernel<4,2> / <3,2> in the other schedules) on the caller's tables: 'fast' >= 5.9 TB/s (reads at 6.9 + writes at 4.6 TB/s add up), "
                                  "'slow' is what a plain 8:1 read/write stream gets on the same box (5.4-5.8 TB/s)"}

    result = None
    if rank == 0:
        dom = kernels[0] if kernels else None
        first_pass = next((r["ks"] for r in log if r["kind"] in ("pass", "gram_pass") and r["kf"] == 0), 0)
        tkey = "n%d_gpus%d_vpp%d_first%d" % (n, world, args.vars_per_pass, first_pass) + ("" if FIELD["name"] == "GoldilocksMont" else "_generic")
        tj = load_traffic(tkey)
        traffic, traffic_step, traffic_check = None, None, "no PMC record for %s in profiles/traffic.json" % tkey
        if tj and tj["_stale"]:
            traffic_check = tj["_stale"]
            sys.stderr.write("bench.py: " + traffic_check + "\n")
        elif tj:
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
        # first-hand counters where this is the headline shape (or asked for): two child runs under rocprofv3 --pmc on THIS box
        own_traffic, own_source, traffic_record = None, None, traffic
        want_own = os.environ.get("SC_BENCH_SELF_PMC", "1")
        if world == 1 and dom and want_own != "0" and (want_own == "force" or (n == 28 and args.vars_per_pass == 2)):
            own, how = self_pmc(args, n, schedule)
            if own:
                idx = next((i for i, sc in enumerate(schedule) if tuple(sc) == tuple(dom["_key"])), None)
                if idx is not None and abs(sum(own) - moved) / moved <= 0.02:
                    own_traffic, own_source = own[idx], ("this run's box, after the timed region: two child passes of " + how +
                                                         "; FETCH_SIZE x 2 KiB + WRITE_SIZE x 1 KiB per MI355X_MICROARCH.md (gfx950 reports half of a wide read)")
                    traffic_step = sum(own)
                    traffic_check = "ok: launches of this run move %.5g B/step, this box's PMC %.5g (%.2f %% apart)" % (moved, sum(own), abs(sum(own) - moved) / moved * 100)
                else:
                    own_source = "own PMC passes disagree with the launch log (%.5g vs %.5g B/step): withheld" % (sum(own), moved)
            else:
                own_source = "own PMC passes not available: " + how
            if own_traffic is not None:
                traffic = own_traffic
        result = {
            "metric": "field mul-adds/sec in sumcheck prover, n=%d vars" % n + ("" if FIELD["name"] == "GoldilocksMont" else " (generic modulus)"),
            "value": value,
            "unit": "field mul-adds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_median": median_ms,
            "value_at_median": muladds / (median_ms * 1e-3) if median_ms else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "full sumcheck prover, g=a*b, n=%d, %s, hypercube sharded by top "
                            "index bits over %d GPU(s), %s" % (
                                n, FIELD["label"], world, ("%s exchange per pass" % transport) if world > 1 else "no collective"),
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
                "clock_ramp": best.get("ramp"),
                "first_proof_ms": (best.get("first_proof") or {}).get("ms"),
                "cold_ms_per_proof": (best.get("ramp") or {}).get("cold_ms_per_proof"),
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
                "traffic_source": own_source if own_traffic is not None else (traffic_source(tj) if traffic else traffic_check),
                "traffic_record": {"hbm_bytes_per_launch": traffic_record, "source": traffic_source(tj) if traffic_record else None,
                                   "own_pmc": own_source} if own_source else None,
                "bytes_per_launch": dom["bytes_per_launch"] if dom else None,
                "avg_launch_us": dom["avg_us"] if dom else None,
                "per_gpu": True,
                "step": {
                    "bytes_moved": moved,
                    "kernel_ms": kernel_ms_per_step,
                    "launches": n_launch / max(steps_with_events, 1),
                    "frac_of_kernel_time": moved / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms_per_step else None,
                    "frac_of_wall_time": moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "fold_pass_mode": fold_pass_mode,
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

    # ---- CPU baseline: the reference-shaped port, 1 core, bounded sample - on rank 0's host cores, at every N (the CPU
    # prover does not shard: a line of an N-GPU run carries the same baseline an N = 1 run on this box would) ----------------
    nc = cpu_sample_size(args.cpu_num_vars)
    probe = reference_probe() if rank == 0 else None
    if rank == 0 and nc > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from oracle import Oracle   # the checker / CPU baseline: only this leg touches oracle/
        import pyref
        assert (pyref.SEED_A, pyref.SEED_B, pyref.SEED_R) == (syn.SEED_A, syn.SEED_B, syn.SEED_R)
        o = Oracle(FIELD["p"])
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
        del oa, ob
        if world == 1:
            # the same sample through the GPU path must agree bit for bit
            del a, b, g
            ga, gb = syn.tables(ctx, nc)
            c1_gpu, ev_gpu, ch_gpu = mm.prove(ctx, mm.G(ga, gb), syn.SEED_R)
            if c1_gpu != c1_cpu or not np.array_equal(ev_gpu, ev_cpu) or not np.array_equal(ch_gpu, och):
                raise SystemExit("PARITY FAILURE: GPU and CPU oracle disagree at n=%d" % nc)
            result["config"]["parity_gate"] += "; bit-exact vs CPU oracle at n=%d ok" % nc
        elif nc == n:
            # N > 1: the sharded proofs that were timed ARE this instance (same n, same seeds): their transcript - identical on
            # every plane, checked above - against the oracle's, bit for bit
            if best["transcript"] != (c1_cpu, ev_cpu.tobytes()):
                raise SystemExit("PARITY FAILURE: the sharded GPU transcript and the CPU oracle disagree at n=%d over %d ranks" % (n, world))
            result["config"]["parity_gate"] += "; the timed %d-rank transcript is bit-exact vs the CPU oracle at n=%d" % (world, n)
        result["cpu_baseline"] = {
            "value": (5 * 2**nc - 7) / cpu_s,
            "unit": "field mul-adds/s",
            "cores": 1,
            "host_cores_total": os.cpu_count(),
            "kind": "port",
            "sample": "same synthetic workload at n=%d (%d mul-adds, %.1f s): oracle/sc_oracle.c sco_prover_run, the "
                      "reference-shaped single-thread C restatement (clone + multiply + sum for c_1, copy-fold-copy "
                      "per table per round, separate sum pass)%s; the Rust reference itself: %s" % (
                          nc, 5 * 2**nc - 7, cpu_s,
                          "" if world == 1 else "; timed on rank 0's host cores while the other ranks wait - the CPU prover is one process whatever N is",
                          "cargo %s, offline arkworks crates %s, reference sources %s -> not buildable here" % (
                              "present" if probe["cargo"] != "absent" else "absent", "present" if probe["offline_ark_crates"] else "absent",
                              "present" if probe["usable"] or not probe["reference_sources"].startswith("absent") else "absent")),
            "reference_probe": probe,
            "all_cores": {"value": (5 * 2**nc - 7) / cpu_mt_s, "cores": os.cpu_count(), "seconds": cpu_mt_s,
                          "note": "same port with the element loops split over OpenMP threads (BASELINE.md CPU-ref-allT)"},
        }
        # ---- BASELINE configs[1] and configs[2] beside the headline: driver-visible, each with its own
        # in-run oracle gate; after the timed region, a few seconds in all
        if world == 1 and n == 28 and os.environ.get("SC_BENCH_SECONDARY", "1") == "1":
            del ga, gb
            result["config"]["secondary"] = secondary_configs(pkg, ctx, o, np)
    elif rank == 0:
        result["cpu_baseline"] = None
        result["config"]["reference_probe"] = probe
    return result


def secondary_configs(pkg, ctx, o, np):
    """configs[2] (full prover, n = 26) and configs[1] (multilinear-extensions evaluate + fix_variable, n = 24) on the same context,
    each timed over a handful of steps (median wall time, HIP-event kernel time and the bytes its launches move -> a fraction of
    the 8 TB/s peak on bytes actually moved) and compared bit for bit with the CPU oracle in this run."""
    import statistics
    mm, syn = pkg.matrix_multiplication, pkg.synthetic
    out = {}

    def measure(step, reps, warm=3):
        for _ in range(warm):
            step()
        ctx.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = step()
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ctx.set_option("time_kernels", 1)
        ctx.kernel_time(reset=True)
        ctx.launch_log(reset=True)
        r = step()
        ctx.synchronize()
        n_launch, kernel_ms = ctx.kernel_time(reset=True)
        log = ctx.launch_log(reset=True)
        ctx.set_option("time_kernels", 0)
        moved = sum(x["bytes_read"] + x["bytes_written"] for x in log)
        return r, {"ms_per_step_median": statistics.median(ts), "launches": n_launch, "kernel_ms": kernel_ms, "bytes_moved": moved,
                   "frac_of_kernel_time": moved / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms else None,
                   "frac_of_wall_time": moved / (statistics.median(ts) * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # configs[2]: "full sumcheck prover n=26 on 1 MI355X, bit-exact round polys vs CPU verifier"
    n2 = 26
    a, b = syn.tables(ctx, n2)
    g = mm.G(a, b)
    (c1, evals, ch), m = measure(lambda: mm.prove(ctx, g, syn.SEED_R), reps=15)
    oa, ob = o.generate(syn.SEED_A, n2), o.generate(syn.SEED_B, n2)
    tc = time.perf_counter()
    c1_cpu, ev_cpu = o.prover_run_mt(oa, ob, np.asarray(ch, dtype=np.uint64))
    gate_s = time.perf_counter() - tc
    if c1 != c1_cpu or not np.array_equal(evals, ev_cpu):
        raise SystemExit("PARITY FAILURE: GPU and CPU oracle disagree at n=%d (secondary config)" % n2)
    problem = check_identities(ctx.field, c1, evals, ch, g.evaluate([int(x) for x in ch]))
    if problem:
        raise SystemExit("PARITY FAILURE at n=%d (secondary config): %s" % (n2, problem))
    m.update({"workload": "full sumcheck prover, n=26, Goldilocks (BASELINE configs[2])", "field_mul_adds_per_s": (5 * 2**n2 - 7) / (m["ms_per_step_median"] * 1e-3),
              "parity_gate": "bit-exact vs CPU oracle (all-cores form, %.2f s) and the verifier's identities at n=%d ok" % (gate_s, n2)})
    out["prover_n26"] = m
    del a, b, g, oa, ob
    # configs[1]: "multilinear-extensions evaluate + fix_variable, n=24 (2^24 evals) on 1 MI355X"
    n1 = 24
    t = pkg.DenseMultilinearExtension.generate(ctx, syn.SEED_A, n1)
    pt = [int(o.challenge(syn.SEED_PT, j)) for j in range(n1)]
    ks = [1, 3, n1 // 2]

    def mle_step():
        outs = [t.evaluate(pt, pkg.ORDER_LE), t.evaluate(pt, pkg.ORDER_BE)]
        outs += [t.fix_variables(pt[:k]) for k in ks]
        ctx.synchronize()
        return outs

    outs, m = measure(mle_step, reps=15)
    ot = o.generate(syn.SEED_A, n1)
    if outs[0] != o.evaluate(ot, pt) or outs[1] != o.vsbw(ot, pt):
        raise SystemExit("PARITY FAILURE: evaluate disagrees with the CPU oracle at n=%d (secondary config)" % n1)
    if not np.array_equal(outs[2 + len(ks) - 1].to_evaluations(), o.fix_variables(ot, pt[:ks[-1]])):
        raise SystemExit("PARITY FAILURE: fix_variables disagrees with the CPU oracle at n=%d (secondary config)" % n1)
    if not np.array_equal(outs[2].to_evaluations()[:4096], o.fix_variables(ot, pt[:1])[:4096]):
        raise SystemExit("PARITY FAILURE: fix_variables k=1 disagrees with the CPU oracle at n=%d (secondary config)" % n1)
    m.update({"workload": "multilinear-extensions evaluate (LE, BE) + fix_variables k=1,3,12 on one 2^24-entry table (BASELINE configs[1]); a step = the five calls",
              "field_mul_adds_per_step": 2 * (2**n1 - 1) + sum(2**n1 - 2**(n1 - k) for k in ks),
              "parity_gate": "evaluate LE / BE, fix_variables k=12 (whole) and k=1 (first 4096 entries) bit-exact vs CPU oracle at n=%d ok" % n1})
    out["mle_n24"] = m
    return out


def run_mle(args, pkg, torch, dist, rank, world, local_rank):
    """BASELINE config 2: evaluate + fix_variable on one 2^n-entry table (single GPU)."""
    import numpy as np
    if world != 1:
        raise SystemExit("--workload mle is a single-GPU workload (run --gpus N as N replicas by hand)")
    syn = pkg.synthetic
    n = args.num_vars if args.num_vars > 0 else 24
    F = pkg.Field(FIELD["p"])
    ctx = pkg.Context(F, device=local_rank)
    t = pkg.DenseMultilinearExtension.generate(ctx, syn.SEED_A, n)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle    # challenge derivation + the checker (after the timed region)
    o = Oracle(FIELD["p"])
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

    # several points per pass (sc_table_evaluate_many; outside the timed steps, own HIP-event timing): what ONE evaluation costs
    # when m of them share a read of the table - the amortisation VERDICT r03 asked for beside the ~0.6 ceiling of a single
    # n = 24 evaluate, whose launch carries ~8-10 us that do not shrink with the table
    many = []
    rngp = np.random.default_rng(9)
    for m_pts in (1, 2, 4, 8, 16):
        pts_m = [[int(o.challenge(syn.SEED_PT + 31 * j, i)) for i in range(n)] for j in range(m_pts)]
        t.evaluate_many(pts_m)
        ctx.set_option("time_kernels", 1)
        ctx.launch_log(reset=True)
        reps_m = 10
        tw = time.perf_counter()
        for _ in range(reps_m):
            vals = t.evaluate_many(pts_m)
        wall_us = (time.perf_counter() - tw) / reps_m * 1e6
        lg = [r for r in ctx.launch_log(reset=True) if r["kind"] == "evaluate"]
        ctx.set_option("time_kernels", 0)
        if vals[0] != t.evaluate(pts_m[0], pkg.ORDER_LE) or vals[-1] != t.evaluate(pts_m[-1], pkg.ORDER_LE):
            raise SystemExit("PARITY FAILURE: evaluate_many(m=%d) differs from single evaluations at n=%d" % (m_pts, n))
        us = statistics.median(r["ms"] for r in lg) * 1e3
        many.append({"points": m_pts, "launch_us": us, "us_per_point": us / m_pts, "wall_us_per_call": wall_us,
                     "table_GBps": 8 * 2**n / (us * 1e-6) / 1e9,
                     "frac_of_peak_per_point": m_pts * 8 * 2**n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
    parity.append("evaluate_many(m = 1, 2, 4, 8, 16) == single evaluations at n=%d" % n)

    tkey = "mle_n%d" % n
    tj = load_traffic(tkey)
    traffic = None
    if tj and not tj["_stale"] and dom["kernel"] in tj.get("kernels", {}):
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
            "evaluate_many": {"rows": many, "note": "sc_table_evaluate_many: m points in ONE pass over the table (not part of the "
                              "timed step); frac_of_peak_per_point credits every point with a full read of the table - what m single "
                              "evaluations would have to reach to match - so it may exceed 1"},
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


def roofline_from_log(log, steps, dominant_kind, traffic_key, note, kernel_ms, n_launch, ms_per_step, alg_bytes):
    """the `roofline` object of a single-GPU workload whose every launch is timed: dominant kernel = the group of
    `dominant_kind` launches with the largest share of the step"""
    kernels = aggregate_launches(log, steps)
    moved = sum(k["bytes_per_launch"] * k["launches_per_step"] for k in kernels)
    kernel_ms_per_step = kernel_ms / steps
    dom = next((k for k in kernels if k["_key"][0] == dominant_kind), kernels[0])
    tj = load_traffic(traffic_key)
    traffic = None
    if tj and not tj["_stale"] and dom["kernel"] in tj.get("kernels", {}):
        traffic = tj["kernels"][dom["kernel"]]["hbm_bytes_per_launch"]
    return {
        "bound": "hbm",
        "kernel": dom["kernel"],
        "achieved": dom["GBps"],
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": dom["GBps"] / HBM_PEAK_GBS if dom["GBps"] else None,
        "traffic": traffic,
        "traffic_source": traffic_source(tj) if traffic else (tj["_stale"] if tj else "no PMC record for %s in profiles/traffic.json" % traffic_key),
        "bytes_per_launch": dom["bytes_per_launch"],
        "avg_launch_us": dom["avg_us"],
        "step": {"bytes_moved": moved, "kernel_ms": kernel_ms_per_step, "launches": n_launch / steps,
                 "frac_of_kernel_time": moved / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms_per_step else None,
                 "frac_of_wall_time": moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "sec8d_algorithmic_bytes": alg_bytes},
        "kernels": [{k: v for k, v in kk.items() if k != "_key"} for kk in kernels],
        "note": note,
    }


def timed_steps(args, torch, ctx, step):
    """W untimed + K timed calls of step() with every launch event-timed; returns (elapsed_s, step_ms, n_launch, kernel_ms, log, last)"""
    ctx.set_option("time_kernels", 1)
    for _ in range(args.warmup):
        step()
    ctx.launch_log(reset=True)
    ctx.kernel_time(reset=True)
    step_ms, last = [], None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        last = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.set_option("time_kernels", 0)
    n_launch, kernel_ms = ctx.kernel_time(reset=True)
    log = ctx.launch_log(reset=True)
    return elapsed, step_ms, n_launch, kernel_ms, log, last


def widened_line(args, metric, value, elapsed, step_ms, config, roofline, cpu):
    return {"metric": metric, "value": value, "unit": "field mul-adds/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": statistics.median(step_ms), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": config, "roofline": roofline,
            "cpu_baseline": cpu}


def random_layer(pkg, rng, k):
    gp = pkg.gkr_protocol
    n = 1 << k
    kinds = ("add", "mul")
    return gp.Circuit([gp.CircuitLayer([gp.Gate(kinds[rng.getrandbits(1)], [rng.randrange(n), rng.randrange(n)]) for _ in range(n)])], n)


def quadratic_at(F, e, r):
    return lagrange_at(F, e, r)


def run_gkr(args, pkg, torch, dist, rank, world, local_rank):
    """BASELINE configs[4]'s inner loop on one GPU: the sumcheck of ONE GKR layer (gkr-protocol/src/lib.rs:373-456 drives
    SumCheckProver<F, W<F>>) with 2^k gates over 2^k values - add_i(r_i,.,.) / mul_i(r_i,.,.) are 4^k-entry tables
    (k = 13: 2 x 512 MiB), built before the timed region as the reference builds W in start_round.  One step = Prover::new
    + 2k rounds, every challenge drawn on the host after its round's sums were read back."""
    import random
    import numpy as np
    if world != 1:
        raise SystemExit("--workload gkr is a single-GPU workload")
    gp = pkg.gkr_protocol
    k = args.num_vars if args.num_vars > 0 else 13
    F = pkg.Field(FIELD["p"])
    ctx = pkg.Context(F, device=local_rank)
    rng = random.Random(2026)
    circuit = random_layer(pkg, rng, k)
    inputs = [F.from_int(rng.randrange(F.p)) for _ in range(1 << k)]
    evaluation = circuit.evaluate(F, inputs)
    r_i = [F.from_int(rng.randrange(F.p)) for _ in range(k)]
    w = gp.start_round_w(ctx, circuit, evaluation, 0, r_i)
    seed = pkg.synthetic.SEED_R

    elapsed, step_ms, n_launch, kernel_ms, log, last = timed_steps(args, torch, ctx, lambda: gp.prove_w(ctx, w, seed))
    c1, evals, ch = last

    # ---- parity gates (outside the timed region) -----------------------------------------------
    wi = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, k, np.array(evaluation[0], dtype=np.uint64))
    if c1 != wi.evaluate(r_i):
        raise SystemExit("PARITY FAILURE: c_1 != W_i~(r_i) at k=%d" % k)
    problem = check_identities(F, c1, evals, ch, w.evaluate([int(x) for x in ch]))
    if problem:
        raise SystemExit("PARITY FAILURE at k=%d: %s" % (k, problem))
    sparse = gp.SparseLayerProver(ctx, circuit, evaluation, 0, r_i)          # the gate-list prover: same rounds
    if sparse.c1() != c1 or any(sparse.round_evals(int(ch[j - 1]) if j else F.one, j) != [int(x) for x in evals[j]] for j in range(2 * k)):
        raise SystemExit("PARITY FAILURE: dense and sparse W provers disagree at k=%d" % k)
    parity = ["k=%d: c_1 == W_i~(r_i), verifier identities, final W::evaluate, dense == sparse prover every round" % k]
    # The layer's last message: restrict_poly(b*, c*, W_{i+1}) (gkr-protocol/src/lib.rs:291-321, sent by round_msg :439-456) - the k + 1
    # evaluations of W~ on the line through the two halves of the challenge vector.  ONE pass over the table since round 4
    # (sc_table_evaluate_many); the k + 1 single evaluations it replaced are timed beside it.  Outside the timed steps.
    wt = w.w_b
    bpt, cpt = [int(x) for x in ch[:k]], [int(x) for x in ch[k:]]
    reps, t_one, t_many = 30, [], []
    q_ref = gp.restrict_poly(bpt, cpt, wt)
    line_pts = [[F.add(bi, F.mul(F.from_int(j), F.sub(ci, bi))) for bi, ci in zip(bpt, cpt)] for j in range(k + 1)]
    for _ in range(reps):
        t0 = time.perf_counter()
        gp.restrict_poly(bpt, cpt, wt)
        t_one.append((time.perf_counter() - t0) * 1e6)
        t0 = time.perf_counter()
        singles = [wt.evaluate(pt) for pt in line_pts]
        t_many.append((time.perf_counter() - t0) * 1e6)
    if [q_ref.evaluate(F.from_int(j)) for j in range(k + 1)] != singles:
        raise SystemExit("PARITY FAILURE: restrict_poly disagrees with the k + 1 single evaluations at k=%d" % k)
    parity.append("restrict_poly(t = 0..k) == the k + 1 single W~ evaluations")
    restrict = {"one_pass_us_median": statistics.median(t_one), "k_plus_1_single_evaluations_us_median": statistics.median(t_many),
                "points": k + 1, "note": "host wall clock incl. the Python wrapper; the second figure has k + 1 wrapper calls"}
    # the same layer shape at a size the oracle finishes in seconds, bit for bit; that run is the CPU baseline
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    o = Oracle(FIELD["p"])
    kc = min(k, args.cpu_num_vars if args.cpu_num_vars > 0 else 12)
    crng = random.Random(7)
    ccirc = random_layer(pkg, crng, kc)
    cev = ccirc.evaluate(F, [F.from_int(crng.randrange(F.p)) for _ in range(1 << kc)])
    cr = [F.from_int(crng.randrange(F.p)) for _ in range(kc)]
    cw = gp.start_round_w(ctx, ccirc, cev, 0, cr)
    gc1, gev, gch = gp.prove_w(ctx, cw, seed)
    oadd, omul = cw.add_i.to_evaluations(), cw.mul_i.to_evaluations()
    ow = np.array(cev[1], dtype=np.uint64)
    tc = time.perf_counter()
    ref = o.w_prove(oadd, omul, ow, ow, [int(x) for x in gch])
    cpu_s = time.perf_counter() - tc
    if ref["status"] != 0 or ref["c_1"] != gc1 or not np.array_equal(ref["evals"], gev):
        raise SystemExit("PARITY FAILURE: GPU W prover and the CPU oracle disagree at k=%d" % kc)
    parity.append("bit-exact vs CPU oracle (sco_w_prove) at k=%d" % kc)

    # algorithmic work of a layer in the two-phase form: P, L: 2 mul-adds per table entry; add(r_b,.), mul(r_b,.): 2 per
    # entry; two product sumchecks on 2^(k+1)-entry tables.  Bytes: both 4^k-entry tables read twice
    muladds = lambda kk: 4 * 4**kk + 2 * (5 * 2**(kk + 1) - 7)      # noqa: E731
    alg_bytes = 32 * 4**k
    note = ("dominant kernel = gkr_phase1_kernel: one streaming pass over add_i and mul_i (2 x 8 x 4^k bytes) producing P and L; "
            "bytes from the launch log of this run, duration = HIP events on the library's stream")
    roof = roofline_from_log(log, args.steps, "gkr", "gkr_k%d" % k, note, kernel_ms, n_launch, elapsed / args.steps * 1e3, alg_bytes)
    config = {"workload": "GKR layer sumcheck (W round polynomial, dense two-phase prover), 2^%d gates over 2^%d values, add/mul tables "
                          "of 4^%d entries, Goldilocks (BASELINE configs[4] inner loop on 1 GPU)" % (k, k, k),
              "k": k, "num_vars": 2 * k, "field_mul_adds_per_step": muladds(k), "algorithmic_bytes_per_step": alg_bytes,
              "restrict_poly": restrict,
              "parity_gate": "; ".join(parity),
              "schedule": [[r["kind"], r["kf"], r["ks"], r["log_in"]] for r in log[: len(log) // args.steps]]}
    cpu = {"value": muladds(kc) / cpu_s, "unit": "field mul-adds/s", "cores": 1, "host_cores_total": os.cpu_count(), "kind": "port",
           "sample": "one layer at k=%d (%.1f s): oracle/sc_oracle.c sco_w_prove, the reference-shaped single-thread restatement of "
                     "W::to_univariate / fix_variables (walks all remaining evaluations every round), credited with the same "
                     "algorithmic mul-add count" % (kc, cpu_s)}
    return widened_line(args, "field mul-adds/sec, GKR layer sumcheck (W), k=%d" % k, muladds(k) * args.steps / elapsed, elapsed, step_ms,
                        config, roof, cpu)


def run_gnew(args, pkg, torch, dist, rank, world, local_rank):
    """BASELINE configs[4]'s polynomial construction on one GPU: matrix_multiplication::G::new
    (matrix-multiplication/src/lib.rs:77-92) on two 2^n x 2^n matrices (n = 14: 2 GiB each) at a random point: f_a =
    MLE(A).relabel(0,n,n).fix_variables(point[..n]) as ONE column-dot pass, f_b = MLE(B).fix_variables(point[n..]) as ONE
    segment-dot pass.  One step = one G::new."""
    import numpy as np
    if world != 1:
        raise SystemExit("--workload gnew is a single-GPU workload")
    mm, syn = pkg.matrix_multiplication, pkg.synthetic
    n = args.num_vars if args.num_vars > 0 else 14
    F = pkg.Field(FIELD["p"])
    ctx = pkg.Context(F, device=local_rank)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    o = Oracle(FIELD["p"])
    A = pkg.DenseMultilinearExtension.generate(ctx, 11, 2 * n)
    B = pkg.DenseMultilinearExtension.generate(ctx, 12, 2 * n)
    pt = [int(o.challenge(syn.SEED_PT, t)) for t in range(2 * n)]

    def step():
        g = mm.G.new_from_tables(ctx, n, A, B, pt)
        return g

    elapsed, step_ms, n_launch, kernel_ms, log, g = timed_steps(args, torch, ctx, step)

    # ---- parity (outside the timed region) -----------------------------------------------------
    side = 1 << n
    bits = lambda v: [F.one if (v >> t) & 1 else F.zero for t in range(n)]   # noqa: E731
    fa, fb = g.f_a.to_evaluations(), g.f_b.to_evaluations()
    for z in (0, 1, side - 1, 0x1234 & (side - 1)):
        if int(fa[z]) != A.evaluate(bits(z) + pt[:n]) or int(fb[z]) != B.evaluate(pt[n:] + bits(z)):
            raise SystemExit("PARITY FAILURE: G::new entry %d is not the matrix MLE at (z, r) at n=%d" % (z, n))
    c1, evals, ch = mm.prove(ctx, g, syn.SEED_R)
    problem = check_identities(F, c1, evals, ch, g.evaluate([int(x) for x in ch]))
    if problem:
        raise SystemExit("PARITY FAILURE at n=%d: %s" % (n, problem))
    parity = ["n=%d: f_a[z] == A~(r1, z), f_b[z] == B~(z, r2) at sampled z; the proof on (f_a, f_b) passes the verifier identities" % n]
    nc = min(n, args.cpu_num_vars if args.cpu_num_vars > 0 else 12)
    oa, ob = o.generate(11, 2 * nc), o.generate(12, 2 * nc)
    opt = np.array(pt[:nc] + pt[n:n + nc], dtype=np.uint64)
    tc = time.perf_counter()
    ofa, ofb = o.g_new(nc, oa, ob, opt)
    cpu_s = time.perf_counter() - tc
    Ac = pkg.DenseMultilinearExtension.generate(ctx, 11, 2 * nc)
    Bc = pkg.DenseMultilinearExtension.generate(ctx, 12, 2 * nc)
    gc = mm.G.new_from_tables(ctx, nc, Ac, Bc, [int(x) for x in opt])
    if not np.array_equal(gc.f_a.to_evaluations(), ofa) or not np.array_equal(gc.f_b.to_evaluations(), ofb):
        raise SystemExit("PARITY FAILURE: GPU G::new and the CPU oracle disagree at n=%d" % nc)
    parity.append("bit-exact vs CPU oracle (sco_g_new) at n=%d" % nc)

    muladds = lambda nn: 2 * 4**nn                     # noqa: E731  every entry of A and of B enters one multiply-add
    alg_bytes = 16 * 4**n + 16 * 2**n
    note = ("dominant kernel = coldot_kernel (f_a: one pass over A against eq(point[..n]), no transposed copy); fix_low_kernel is the "
            "f_b half; bytes from the launch log of this run")
    roof = roofline_from_log(log, args.steps, "coldot", "gnew_n%d" % n, note, kernel_ms, n_launch, elapsed / args.steps * 1e3, alg_bytes)
    config = {"workload": "matrix_multiplication::G::new on two 2^%d x 2^%d Goldilocks matrices (2^%d entries each) at a random point "
                          "(BASELINE configs[4] polynomial construction on 1 GPU)" % (n, n, 2 * n),
              "n": n, "num_vars": 2 * n, "field_mul_adds_per_step": muladds(n), "algorithmic_bytes_per_step": alg_bytes,
              "parity_gate": "; ".join(parity),
              "schedule": [[r["kind"], r["kf"], r["ks"], r["log_in"]] for r in log[: len(log) // args.steps]]}
    cpu = {"value": muladds(nc) / cpu_s, "unit": "field mul-adds/s", "cores": 1, "host_cores_total": os.cpu_count(), "kind": "port",
           "sample": "G::new at n=%d (%.1f s): oracle/sc_oracle.c sco_g_new, the reference-shaped single-thread restatement (relabel = "
                     "full-table swap pass, then n one-variable folds with per-call table copies)" % (nc, cpu_s)}
    return widened_line(args, "field mul-adds/sec, matrix_multiplication::G::new, n=%d" % n, muladds(n) * args.steps / elapsed, elapsed,
                        step_ms, config, roof, cpu)


def run_triangle(args, pkg, torch, dist, rank, world, local_rank):
    """triangle_counting::G on one GPU: Prover<F, G> for G::new_adj_matrix on a 2^k-vertex graph
    (k = 10), all 3k rounds.  One step = Prover::new (the n^3 matrix square) + 3k rounds."""
    import numpy as np
    if world != 1:
        raise SystemExit("--workload triangle is a single-GPU workload")
    tc_mod = pkg.triangle_counting
    k = args.num_vars if args.num_vars > 0 else 10
    F = pkg.Field(FIELD["p"])
    ctx = pkg.Context(F, device=local_rank)
    seed = pkg.synthetic.SEED_R

    def graph(kk, s):
        nn = 1 << kk
        upper = np.triu(np.random.default_rng(s).random((nn, nn)) < 0.25, 1)
        return upper | upper.T

    def table(m, kk):
        ev = np.where(m.flatten(), np.uint64(F.one), np.uint64(0)).astype(np.uint64)
        t = pkg.DenseMultilinearExtension.from_evaluations_vec(ctx, 2 * kk, ev)
        return tc_mod.G(t, t, t, kk)

    m = graph(k, 77)
    g = table(m, k)
    elapsed, step_ms, n_launch, kernel_ms, log, last = timed_steps(args, torch, ctx, lambda: tc_mod.prove(ctx, g, seed))
    c1, evals, ch = last

    a = m.astype(np.int64)
    tri = int(np.trace(a @ a @ a)) // 6
    if F.to_int(c1) != (6 * tri) % F.p:
        raise SystemExit("PARITY FAILURE: c_1 != 6 x triangles at k=%d" % k)
    problem = check_identities(F, c1, evals, ch, g.evaluate([int(x) for x in ch]))
    if problem:
        raise SystemExit("PARITY FAILURE at k=%d: %s" % (k, problem))
    parity = ["k=%d: c_1 == 6 x %d triangles (numpy), verifier identities, final G::evaluate" % (k, tri)]
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import Oracle
    o = Oracle(FIELD["p"])
    kc = min(k, args.cpu_num_vars if args.cpu_num_vars > 0 else 9)
    mc = graph(kc, 78)
    gc = table(mc, kc)
    gc1, gev, gch = tc_mod.prove(ctx, gc, seed)
    oadj = np.where(mc.flatten(), np.uint64(F.one), np.uint64(0)).astype(np.uint64)
    t0 = time.perf_counter()
    ref = o.tri_prove(oadj, kc, [int(x) for x in gch])
    cpu_s = time.perf_counter() - t0
    if ref["status"] != 0 or ref["c_1"] != gc1 or not np.array_equal(ref["evals"], gev):
        raise SystemExit("PARITY FAILURE: GPU triangle prover and the CPU oracle disagree at k=%d" % kc)
    parity.append("bit-exact vs CPU oracle (sco_tri_prove) at k=%d" % kc)

    # n^3 multiply-adds of the matrix square, then three product sumchecks (x: 4^k-entry tables, y and z: 2^k), one
    # column-dot and two k-variable folds over the 4^k-entry table
    muladds = lambda kk: 8**kk + (5 * 4**kk - 7) + 2 * (5 * 2**kk - 7) + 3 * 4**kk      # noqa: E731
    alg_bytes = 8 * 4**k * 6
    # the dominant launch group of the proof: the matrix square (int8 MFMA on a 0/1 adjacency table: priced against the dense
    # int8 matrix-core peak), or - once that is a few per cent of the proof - the largest streaming pass (HBM)
    groups = aggregate_launches(log, args.steps)
    dom_kind = groups[0]["_key"][0]
    note = ("dominant launch group by time: %s.  matsq = matsq_bytes_kernel (table -> int8 bytes and transpose) + matsq_mfma_kernel "
            "(v_mfma_i32_32x32x32_i8, exact counts) + the skipped generic fallback, timed as one group; bytes from the launch log" % dom_kind)
    roof = roofline_from_log(log, args.steps, dom_kind if dom_kind != "matsq" else groups[min(1, len(groups) - 1)]["_key"][0],
                             "triangle_k%d" % k, note, kernel_ms, n_launch, elapsed / args.steps * 1e3, alg_bytes)
    ms_rec = [r for r in log if r["kind"] == "matsq"]
    if ms_rec:
        us = sum(r["ms"] for r in ms_rec) / len(ms_rec) * 1e3
        tops = 2 * 8**k / (us * 1e-6) / 1e12
        roof["matsq"] = {"kernel": "sc::matsq_bytes_kernel + sc::matsq_mfma_kernel<GoldilocksMont> on a 2^%d x 2^%d 0/1 matrix" % (k, k),
                         "bound": "mfma", "avg_launch_us": us, "int8_multiply_adds": 8**k, "achieved": tops, "peak": INT8_PEAK_TOPS,
                         "unit": "TOP/s (int8 multiply-add = 2 ops)", "frac": tops / INT8_PEAK_TOPS}
        if dom_kind == "matsq":
            roof.update({"bound": "mfma", "kernel": roof["matsq"]["kernel"], "achieved": tops, "peak": INT8_PEAK_TOPS,
                         "unit": "TFLOP/s", "frac": tops / INT8_PEAK_TOPS, "avg_launch_us": us, "bytes_per_launch": None, "traffic": None})
    config = {"workload": "triangle_counting::G prover, 2^%d vertices (adjacency MLE of 2^%d entries), Goldilocks" % (k, 2 * k),
              "k": k, "num_vars": 3 * k, "field_mul_adds_per_step": muladds(k), "algorithmic_bytes_per_step": alg_bytes,
              "parity_gate": "; ".join(parity),
              "schedule": [[r["kind"], r["kf"], r["ks"], r["log_in"]] for r in log[: len(log) // args.steps]]}
    cpu = {"value": muladds(kc) / cpu_s, "unit": "field mul-adds/s", "cores": 1, "host_cores_total": os.cpu_count(), "kind": "port",
           "sample": "one proof at k=%d (%.1f s): oracle/sc_oracle.c sco_tri_prove, the reference-shaped single-thread restatement "
                     "(walks all 2^(3k) evaluations every round), credited with the same algorithmic mul-add count" % (kc, cpu_s)}
    return widened_line(args, "field mul-adds/sec, triangle_counting::G prover, k=%d" % k, muladds(k) * args.steps / elapsed, elapsed,
                        step_ms, config, roof, cpu)


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
    ap.add_argument("--workload", choices=["prover", "mle", "gkr", "gnew", "triangle"], default=os.environ.get("SC_BENCH_WORKLOAD", "prover"))
    ap.add_argument("--num-vars", type=int, default=int(os.environ.get("SC_BENCH_N", "0")),
                    help="0 = the workload's BASELINE size (prover n=28, mle n=24, gkr k=13, gnew n=14, triangle k=10)")
    ap.add_argument("--cpu-num-vars", type=int, default=int(os.environ.get("SC_BENCH_CPU_N", "-1")),
                    help="size of the bounded CPU-baseline sample (0 disables; -1 = as large as host memory allows, <= 28)")
    ap.add_argument("--vars-per-pass", type=int, default=2)
    ap.add_argument("--field", default=os.environ.get("SC_BENCH_FIELD", "goldilocks"),
                    help="goldilocks (BASELINE's 64-bit prime field, the headline) | generic (p = 2^64-59 through the kernels of "
                         "every other modulus: the reference's Fp64<MontBackend<T,1>>) | an odd prime below 2^64")
    args = ap.parse_args()
    if args.field == "goldilocks":
        FIELD["p"] = 2**64 - 2**32 + 1
    else:
        FIELD["p"] = GENERIC_P if args.field == "generic" else int(args.field, 0)
        FIELD["name"] = "MontGeneric"
        FIELD["label"] = "generic-modulus Montgomery field p=%d%s" % (FIELD["p"], " (2^64-59)" if FIELD["p"] == GENERIC_P else "")

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

    run = {"prover": run_prover, "mle": run_mle, "gkr": run_gkr, "gnew": run_gnew, "triangle": run_triangle}[args.workload]
    result = run(args, pkg, torch, dist, rank, world, local_rank)
    if rank == 0:
        print(json.dumps(result), flush=True)
    abandoned = bool(ABANDONED)
    if dist is not None:          # ... on ANY rank: then nobody waits in a final barrier for a rank that has to leave
        flag = torch.tensor([1 if abandoned else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        abandoned = int(flag.item()) == 1
    # a plane that failed WHILE PROVING is a failure of the run even though the other planes' line is out
    code = 3 if PROVING_FAILED else 0
    if abandoned:                 # something may never return (a stuck collective): leave without destroying it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return code


if __name__ == "__main__":
    sys.exit(main() or 0)
