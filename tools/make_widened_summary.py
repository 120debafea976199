"""rocprofv3 outputs of one `bench.py --workload gkr|gnew|triangle` run -> profiles/<tag>_<workload>_kernel_stats.csv,
profiles/<tag>_<workload>_summary.md and a profiles/traffic.json record (PMC HBM bytes per launch of every kernel of
the run's largest launches, keyed by the kernel names bench.py prints).

usage: make_widened_summary.py <tag> <workload> <stats_dir> <fetch_dir> <write_dir> <bench_json>"""
import csv
import glob
import json
import provenance
import os
import re
import sys

tag, workload, d_stats, d_fetch, d_write, bench_json = sys.argv[1:7]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "profiles")


def one(d, pat):
    return max(glob.glob(os.path.join(d, "**", pat), recursive=True), key=os.path.getmtime)


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")


bench = None
for line in open(bench_json):
    if line.startswith("{"):
        bench = json.loads(line)
stats = list(csv.DictReader(open(one(d_stats, "*_kernel_stats.csv"))))
with open(os.path.join(P, "%s_%s_kernel_stats.csv" % (tag, workload)), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def counters(d, cname):
    """kernel -> counter values of its LARGEST launches (>= 90 % of the kernel's maximum: the timed steps at the
    benchmarked size; the parity gates launch the same kernels on smaller tables)"""
    per = {}
    for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))):
        if r["Counter_Name"] == cname:
            per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return {k: [x for x in v if x >= 0.9 * max(v)] for k, v in per.items()}


fetch, write = counters(d_fetch, "FETCH_SIZE"), counters(d_write, "WRITE_SIZE")
lines = ["# %s: rocprofv3 summary, `bench.py --workload %s`, 1 x MI355X\n" % (tag, workload),
         "Command (on the GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload %s --steps 10 --warmup 2`;" % workload,
         "counters from two more runs with `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (+ `--kernel-trace`), `--steps 2 --warmup 1`.\n",
         "## Kernel totals (`%s_%s_kernel_stats.csv`; includes the parity gates' launches on smaller tables)\n" % (tag, workload),
         "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
for r in stats[:14]:
    lines.append("| `%s` | %s | %.1f | %.3f | %s |" % (short(r["Name"])[:110], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
lines += ["\n## PMC HBM bytes per launch (largest launches of each kernel)\n",
          "HBM bytes = FETCH_SIZE x 2 x 1024 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM) + WRITE_SIZE x 1024.\n",
          "| kernel | launches averaged | FETCH (corrected) B | WRITE B | HBM B per launch |", "|---|---|---|---|---|"]
pmc = {}
for k in sorted(fetch, key=lambda k: -sum(fetch[k]) / len(fetch[k])):
    fb = sum(fetch[k]) / len(fetch[k]) * 2 * 1024
    wv = write.get(k, [0.0])
    wb = sum(wv) / len(wv) * 1024
    pmc[k] = fb + wb
    if fb + wb > 1e6:
        lines.append("| `%s` | %d | %.5g | %.5g | %.5g |" % (k[:110], len(fetch[k]), fb, wb, fb + wb))
rf = bench["roofline"]
lines.append("\n## bench.py line of the un-profiled run of the same build\n")
lines.append("`%s`: value = %.4g %s, ms_per_step = %.4f (median %.4f); roofline (%s): `%s` %.4g %s = %.3f of peak (avg launch %.1f us)." % (
    bench["metric"], bench["value"], bench["unit"], bench["ms_per_step"], bench["ms_per_step_median"], rf["bound"], rf["kernel"],
    rf["achieved"], rf["unit"], rf["frac"], rf["avg_launch_us"]))
lines.append("\n| launch group (bench.py launch log) | launches/step | avg us | bytes per launch | GB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|")
for k in rf["kernels"]:
    lines.append("| `%s` | %.0f | %.1f | %.5g | %.0f | %.3f |" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["bytes_per_launch"], k["GBps"] or 0, (k["GBps"] or 0) / 8000))
if "matsq" in rf:
    lines.append("\nmatsq: %s" % json.dumps(rf["matsq"]))
lines.append("\nparity gate of that run: %s" % bench["config"]["parity_gate"])
lines.append("\ncpu_baseline: %s" % json.dumps(bench["cpu_baseline"]))
open(os.path.join(P, "%s_%s_summary.md" % (tag, workload)), "w").write("\n".join(lines) + "\n")

# traffic.json: the dominant kernel's PMC bytes under the name bench.py prints for it
tj_path = os.path.join(P, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
key = {"gkr": "gkr_k%d", "gnew": "gnew_n%d", "triangle": "triangle_k%d"}[workload] % bench["config"].get("k", bench["config"].get("n", 0))
frag = {"gkr": "gkr_phase1_kernel", "gnew": "coldot_kernel", "triangle": "wgrid_pass_kernel"}[workload]
cand = [v for k, v in pmc.items() if frag in k]
tj[key] = {"kernels": {rf["kernel"]: {"hbm_bytes_per_launch": max(cand) if cand else None}},
           "source": "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md" % tag,
           "provenance": provenance.record(tag, os.environ.get("SC_PMC_COMMAND") or
                                           "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --workload %s ..." % workload)}
json.dump(tj, open(tj_path, "w"), indent=1)
print(open(os.path.join(P, "%s_%s_summary.md" % (tag, workload))).read())
