"""summarise a rocprofv3 kernel-trace CSV: per-launch durations of the last prover run"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
def fmt(r):
    m = re.search(r'pass_kernel<sc::(\w+), (\d), (\d)(?:, \d)?>', r['Kernel_Name'])
    return (m.group(2) + m.group(3), int(r['Grid_Size_X']) // 256, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
seq = [fmt(r) for r in rows if 'pass_kernel' in r['Kernel_Name']]
# split into prover runs: a run starts with a kf=0 pass
runs, cur = [], []
for x in seq:
    if x[0][0] == '0' and cur:
        runs.append(cur); cur = []
    cur.append(x)
runs.append(cur)
for label, run in (("first-schedule", runs[len(runs)//2 - 1]), ("last-schedule", runs[-1])):
    print(label, "passes=%d total=%.1f us" % (len(run), sum(t for _, _, t in run)))
    size = n
    for kfks, grid, t in run:
        kf, ks = int(kfks[0]), int(kfks[1])
        rd = 16 * 2**size; wr = 16 * 2**(size - kf) if kf else 0
        print("  kf=%d ks=%d in=2^%d grid=%d  %.1f us  -> %.2f TB/s actual" % (kf, ks, size, grid, t, (rd + wr) / t / 1e6))
        size -= kf
fr = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'final_reduce' in r['Kernel_Name']]
print("final_reduce: n=%d avg %.2f us" % (len(fr), sum(fr)/max(1,len(fr))))
