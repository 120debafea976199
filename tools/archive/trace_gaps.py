"""rocprofv3 kernel-trace CSV -> idle time between consecutive pass kernels of the last proof"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'pass' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-11:] if len(rows) >= 11 else rows
prev = None
tot_k = tot_g = 0
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][-38:]
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-40s grid=%6d  dur=%8.1f us  gap_before=%6.1f us" % (name, int(r['Grid_Size_X']) // 256, (e - s) / 1e3, gap))
    tot_k += (e - s) / 1e3; tot_g += gap
    prev = e
print("kernels %.1f us, gaps %.1f us" % (tot_k, tot_g))
