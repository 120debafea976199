"""Turn rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

usage: make_profile_summary.py <tag> <workload: prover|mle> <stats_dir> <fetch_dir> <write_dir> <bench_json> [n]

Writes profiles/<tag>_<workload>_kernel_stats.csv, profiles/<tag>_<workload>_summary.md and the PMC traffic
records bench.py reads (profiles/traffic.json: bytes per step and per launch of each kernel, keyed by the
kernel names bench.py prints)."""
import csv
import glob
import json
import provenance
import os
import re
import sys

tag, workload, d_stats, d_fetch, d_write, bench_json = sys.argv[1:7]
n = int(sys.argv[7]) if len(sys.argv) > 7 else (28 if workload == "prover" else 24)
FIELD = sys.argv[8] if len(sys.argv) > 8 else "GoldilocksMont"      # the field policy of the kernels (MontGeneric: bench.py --field generic)
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def one(d, pat):
    return max(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


bench = None
for line in open(bench_json):
    if line.startswith("{"):
        bench = json.loads(line)

# ---- kernel stats (rocprofv3 --kernel-trace --stats)
stats = list(csv.DictReader(open(one(d_stats, "*_kernel_stats.csv"))))
with open(os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, workload)), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

# ---- per-launch trace: the hot-path kernels in dispatch order
HOT = ("pass_kernel<", "wgrid_pass_kernel<", "evaluate_kernel<", "fold_kernel<", "fix_low_kernel<", "fold_be_kernel<", "coldot_kernel<")


def is_hot(name):
    return any(h in name for h in HOT)


def describe(name):
    """(kind, kf, ks) of a hot kernel from its demangled name"""
    m = re.search(r"gram_pass_kernel<(?:sc::\w+, )?(\d)(?:, \w+)?>", name)
    if m:
        return "gram_pass", 0, int(m.group(1))
    m = re.search(r"gram_finish_kernel<sc::\w+, (\d)>", name)
    if m:
        return "gram_finish", 0, int(m.group(1))
    if "wgrid_pass_kernel<" in name:
        return "grid_pass", -1, -1       # kf, ks from bench.py's schedule
    m = re.search(r"wfold_pass_kernel<sc::\w+, (\d), (\d), \w+>", name)
    if m:
        return "wfold_pass", int(m.group(1)), int(m.group(2))
    m = re.search(r"pass_kernel<sc::\w+, (\d), (\d)(?:, \d+)?>", name)
    if m:
        return "pass", int(m.group(1)), int(m.group(2))
    m = re.search(r"fold_kernel<sc::\w+, (\d)", name)
    if m:
        return "fold", int(m.group(1)), 0
    for k in ("evaluate", "fix_low", "fold_be", "coldot"):
        if k + "_kernel<" in name:
            return k, 0, 0
    return "other", 0, 0


if workload == "prover":   # a proof is passes only (the evaluate launches behind it are bench.py's parity gate)
    HOT = ("pass_kernel<", "wgrid_pass_kernel<")   # ("pass_kernel<" also matches gram_pass_kernel<)
trace = [r for r in csv.DictReader(open(one(d_stats, "*_kernel_trace.csv"))) if is_hot(r["Kernel_Name"])]


def split_steps(rows):
    """one list per step: a prover step starts with a kf = 0 pass, an mle step with evaluate_kernel"""
    steps, cur = [], []
    for r in rows:
        kind, kf, ks = describe(r["Kernel_Name"])
        start = ((kind == "pass" and kf == 0) or kind == "gram_pass") if workload == "prover" else (kind == "evaluate" and (not cur or describe(cur[-1]["Kernel_Name"])[0] != "evaluate"))
        if start and cur:
            steps.append(cur)
            cur = []
        cur.append(r)
    if cur:
        steps.append(cur)
    return steps


steps = split_steps(trace)
# the stats run is `--steps 10 --warmup 2`, the counter runs `--steps 2 --warmup 1`: take the LAST TIMED step
# (after it bench.py runs its parity checks, which launch the same kernels at other sizes)
# (prover: bench.py proves once before its warm-up - config.first_proof_ms - so positions shift; the proof that ends the trace is
# the last timed one either way)
last = steps[-1] if workload == "prover" else steps[11]


def counters(d, cname):
    rows = [r for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))) if r["Counter_Name"] == cname and is_hot(r["Kernel_Name"])]
    st = split_steps(rows)
    return [float(r["Counter_Value"]) for r in (st[-1] if workload == "prover" else st[2])]


fetch = counters(d_fetch, "FETCH_SIZE")
write = counters(d_write, "WRITE_SIZE")
assert len(fetch) == len(last) == len(write), (len(fetch), len(last), len(write))


def bench_name(kind, kf, ks, log_in):
    return _bench_name(kind, kf, ks, log_in).replace("GoldilocksMont", FIELD)


def _bench_name(kind, kf, ks, log_in):
    if kind == "pass":
        return "sc::pass_kernel<GoldilocksMont,%d,%d> on 2^%d-entry tables" % (kf, ks, log_in)
    if kind == "grid_pass":
        return "sc::wgrid_pass_kernel<GoldilocksMont,ks> (kf=%d, ks=%d) on 2^%d-entry tables" % (kf, ks, log_in)
    if kind == "wfold_pass":
        return "sc::wfold_pass_kernel<GoldilocksMont,%d,%d> on 2^%d-entry tables" % (kf, ks, log_in)
    if kind == "gram_pass":
        return "sc::gram_pass_kernel<%d> (rounds 1..%d from one read) on 2^%d-entry tables" % (ks, ks, log_in)
    if kind == "gram_finish":
        return "sc::gram_finish_kernel<GoldilocksMont,%d> (partials -> %d cells) behind the 2^%d-entry pass" % (ks, 3 ** ks, log_in)
    if kind == "evaluate":
        return "sc::evaluate_kernel<GoldilocksMont> on a 2^%d-entry table" % log_in
    if kind == "fold":
        return "sc::fold_kernel<GoldilocksMont,%d> on a 2^%d-entry table" % (kf, log_in)
    if kind == "fix_low":
        return "sc::fix_low_kernel<GoldilocksMont> (%d variables) on a 2^%d-entry table" % (log_in // 2, log_in)
    return "%s 2^%d" % (kind, log_in)


lines = []
cmd = "python3 bench.py --steps 10 --warmup 2 --cpu-num-vars 0" + (" --workload mle --num-vars %d" % n if workload == "mle" else "") + (
    " --field generic" if FIELD != "GoldilocksMont" else "") + "  (SC_BENCH_RAMP_MS=0 exported: the profiled runs skip bench.py's clock-ramp setup phase)"
lines.append("# %s: rocprofv3 summary, %s workload, n=%d, 1 x MI355X\n" % (tag, workload, n))
lines.append("Command (on the GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -- %s`;" % cmd)
lines.append("counters from two more runs of the same command with `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (+ `--kernel-trace`), `--steps 2 --warmup 1`.\n")
lines.append("## Kernel totals (`%s_%s_kernel_stats.csv`)\n" % (tag, workload))
lines.append("| kernel | calls | avg us | total ms | % |")
lines.append("|---|---|---|---|---|")
for r in stats:
    lines.append("| `%s` | %s | %.1f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
lines.append("\n## One step, launch by launch (last step of the trace)\n")
lines.append("HBM bytes = FETCH_SIZE x 2 x 1024 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM) + WRITE_SIZE x 1024.\n")
lines.append("| # | kernel | input entries/table | blocks x threads | kernel us | bytes the launch must move | PMC HBM bytes | TB/s of bytes moved | frac of 8 TB/s |")
lines.append("|---|---|---|---|---|---|---|---|---|")
size = n
tot_t = tot_b = tot_p = 0.0
per_kernel = {}
schedule = (bench or {}).get("config", {}).get("schedule") or []
for i, r in enumerate(last):
    kind, kf, ks = describe(r["Kernel_Name"])
    if kf < 0 or ks < 0:                      # run-time (kf, ks): the i-th launch of a proof in bench.py's launch log
        assert i < len(schedule) and schedule[i][0] == "grid_pass", (i, schedule)
        kf, ks = schedule[i][1], schedule[i][2]
    t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    wg = int(r.get("Workgroup_Size_X", 256) or 256)
    grid = "%d x %d" % (int(r["Grid_Size_X"]) // wg, wg)
    if workload == "prover":
        log_in = size
        need = 16 * 2 ** size + (16 * 2 ** (size - kf) if kf else 0)
        size -= kf
    else:
        log_in = n
        k_fix = {("fold", 1): 1, ("fold", 3): 3, ("fix_low", 0): n // 2}.get((kind, kf), None)
        need = 8 * 2 ** n + (8 * 2 ** (n - k_fix) if k_fix else 0)
    pmc = fetch[i] * 2 * 1024 + write[i] * 1024
    name = bench_name(kind, kf, ks, log_in)
    if kind in ("gram_pass", "gram_finish"):   # the partials between the two: from the launch log of the un-profiled run
        rec = next((k for k in (bench or {}).get("roofline", {}).get("kernels", []) if k["kernel"] == name), None)
        need = rec["bytes_per_launch"] if rec else (16 * 2 ** size if kind == "gram_pass" else 0)
    per_kernel.setdefault(name, []).append(pmc)
    lines.append("| %d | `%s` | 2^%d | %s | %.1f | %.4g | %.4g | %.2f | %.3f |" % (i, name, log_in, grid, t, need, pmc, need / t / 1e6, need / t / 1e6 / 8))
    tot_t += t
    tot_b += need
    tot_p += pmc
lines.append("| total | | | | %.1f | %.5g | %.5g | %.2f | %.3f |" % (tot_t, tot_b, tot_p, tot_b / tot_t / 1e6, tot_b / tot_t / 1e6 / 8))
if workload == "prover":
    alg = 64 * 2 ** n - 96
    lines.append("\nSURVEY.md section 8d's one-round-per-pass byte model for this instance: 64*2^n - 96 = %.5g B; the schedule above moves %.5g B"
                 " (PMC: %.5g).  Summed kernel time %.1f us -> %.0f GB/s of bytes actually moved = %.1f %% of the 8 TB/s HBM peak." % (
                     alg, tot_b, tot_p, tot_t, tot_b / tot_t / 1e3, tot_b / tot_t / 1e3 / 80))
if bench:
    rf = bench["roofline"]
    lines.append("\nbench.py line of the un-profiled run of the same build: value = %.4g %s, ms_per_step = %.4f (median %.4f); roofline: `%s` "
                 "%.0f GB/s = %.3f of peak (avg launch %.1f us); step: %.5g B moved in %.4f ms of kernel time = %.3f of peak." % (
                     bench["value"], bench["unit"], bench["ms_per_step"], bench.get("ms_per_step_median", 0), rf["kernel"], rf["achieved"], rf["frac"],
                     rf["avg_launch_us"], rf["step"]["bytes_moved"], rf["step"]["kernel_ms"], rf["step"]["frac_of_kernel_time"]))
open(os.path.join(P, "%s_%s_summary.md" % (tag, workload)), "w").write("\n".join(lines) + "\n")

tj_path = os.path.join(P, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
if workload == "prover":
    first = describe(last[0]["Kernel_Name"])[2]
    key = "n%d_gpus1_vpp2_first%d" % (n, first) + ("" if FIELD == "GoldilocksMont" else "_generic")
else:
    key = "mle_n%d" % n
tj[key] = {
    "hbm_bytes_per_step": tot_p, "fetch_bytes_corrected": sum(fetch) * 2 * 1024, "write_bytes": sum(write) * 1024,
    "kernels": {name: {"hbm_bytes_per_launch": sum(v) / len(v), "launches_per_step": len(v)} for name, v in per_kernel.items()},
    "source": "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md" % tag,
    # what the bytes were measured on (tools/provenance.py): bench.py hands them out only while the kernel sources are these
    "provenance": provenance.record(tag, os.environ.get("SC_PMC_COMMAND") or
                                    "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-num-vars 0"
                                    + ("" if workload != "prover" or n == 28 else " --num-vars %d" % n) + ("" if FIELD == "GoldilocksMont" else " --field generic")),
}
json.dump(tj, open(tj_path, "w"), indent=1)
print(open(os.path.join(P, "%s_%s_summary.md" % (tag, workload))).read())
