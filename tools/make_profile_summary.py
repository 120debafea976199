"""Turn rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.
usage: make_profile_summary.py <tag> <stats_dir> <fetch_dir> <write_dir> <bench_json> [n] [vpp]"""
import csv, glob, json, os, re, sys
tag, d_stats, d_fetch, d_write, bench_json = sys.argv[1:6]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 28
vpp = int(sys.argv[7]) if len(sys.argv) > 7 else 2
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

def one(d, pat):
    return max(glob.glob(os.path.join(d, "*", pat)), key=os.path.getmtime)

def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "")

# ---- kernel stats (rocprofv3 --kernel-trace --stats)
stats = list(csv.DictReader(open(one(d_stats, "*_kernel_stats.csv"))))
with open(os.path.join(P, "%s_kernel_stats.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

# ---- per-launch trace of the last prover run
trace = list(csv.DictReader(open(one(d_stats, "*_kernel_trace.csv"))))
def is_pass(name):
    return "pass_kernel<" in name or "small_pass3_kernel<" in name
def passinfo(r):
    m = re.search(r"small_pass3_kernel<sc::(\w+), (\d)>", r["Kernel_Name"])
    if m:
        kf, ks = int(m.group(2)), 3      # the three-round tail pass
    else:
        m = re.search(r"pass_kernel<sc::(\w+), (\d), (\d)(?:, \d)?>", r["Kernel_Name"])
        kf, ks = int(m.group(2)), int(m.group(3))
    return (kf, ks, int(r["Grid_Size_X"]) // 256, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
seq = [passinfo(r) for r in trace if is_pass(r["Kernel_Name"])]
runs, cur = [], []
for x in seq:
    if x[0] == 0 and cur:
        runs.append(cur); cur = []
    cur.append(x)
runs.append(cur)

def counters(d, name):
    rows = list(csv.DictReader(open(one(d, "*_counter_collection.csv"))))
    vals = [float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == name and is_pass(r["Kernel_Name"])]
    return vals[-len(runs[-1]):]
fetch = counters(d_fetch, "FETCH_SIZE")
write = counters(d_write, "WRITE_SIZE")

bench = None
for line in open(bench_json):
    if line.startswith("{"):
        bench = json.loads(line)

lines = []
lines.append("# %s: rocprofv3 summary, sumcheck prover n=%d, vars_per_pass=%d, 1 x MI355X\n" % (tag, n, vpp))
lines.append("Command (on the GPU box): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 --cpu-num-vars 0`;")
lines.append("counters from two more runs of the same command with `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (+ `--kernel-trace`), `--steps 2 --warmup 1`.\n")
lines.append("## Kernel totals (`%s_kernel_stats.csv`)\n" % tag)
lines.append("| kernel | calls | avg us | total ms | % |")
lines.append("|---|---|---|---|---|")
for r in stats:
    lines.append("| `%s` | %s | %.1f | %.3f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
lines.append("\n## One prover run, launch by launch (last step of the trace)\n")
lines.append("HBM bytes = FETCH_SIZE x 2 x 1024 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM) + WRITE_SIZE x 1024.\n")
lines.append("| pass | kf | ks | input entries/table | grid | kernel us | schedule bytes (read+write) | PMC HBM bytes | schedule TB/s |")
lines.append("|---|---|---|---|---|---|---|---|---|")
size = n
tot_t = tot_b = tot_p = 0
for i, (kf, ks, grid, t) in enumerate(runs[-1]):
    rd = 16 * 2**size; wr = 16 * 2**(size - kf) if kf else 0
    pmc = fetch[i] * 2 * 1024 + write[i] * 1024
    lines.append("| %d | %d | %d | 2^%d | %d | %.1f | %.4g | %.4g | %.2f |" % (i, kf, ks, size, grid, t, rd + wr, pmc, (rd + wr) / t / 1e6))
    tot_t += t; tot_b += rd + wr; tot_p += pmc
    size -= kf
alg = 64 * 2**n - 96
lines.append("| total | | | | | %.1f | %.5g | %.5g | %.2f |" % (tot_t, tot_b, tot_p, tot_b / tot_t / 1e6))
lines.append("\nAlgorithmic bytes of the instance (SURVEY.md section 8d): 64*2^n - 96 = %.5g.  Summed pass-kernel time of this run %.1f us"
             " -> %.0f GB/s algorithmic (%.1f %% of the 8 TB/s HBM peak), %.0f GB/s of bytes actually moved (%.1f %%)." % (
                 alg, tot_t, alg / tot_t / 1e3, alg / tot_t / 1e3 / 80, tot_b / tot_t / 1e3, tot_b / tot_t / 1e3 / 80))
if bench:
    lines.append("\nbench.py line of the un-profiled run of the same build: value = %.4g %s, ms_per_step = %.3f, roofline.achieved = %.0f GB/s (kernel_ms_per_step %.3f)." % (
        bench["value"], bench["unit"], bench["ms_per_step"], bench["roofline"]["achieved"], bench["roofline"]["kernel_ms_per_step"]))
open(os.path.join(P, "%s_summary.md" % tag), "w").write("\n".join(lines) + "\n")

tj_path = os.path.join(P, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
first = runs[-1][0][1] if vpp == 2 else 1   # rounds served by the first pass of the traced run
tj["n%d_gpus1_vpp%d_first%d" % (n, vpp, first)] = {
    "hbm_bytes_per_step": tot_p, "fetch_bytes_corrected": sum(fetch) * 2 * 1024, "write_bytes": sum(write) * 1024,
    "source": "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE doubled per MI355X_MICROARCH.md" % tag,
}
json.dump(tj, open(tj_path, "w"), indent=1)
print(open(os.path.join(P, "%s_summary.md" % tag)).read())
