"""rocprofv3 kernel-trace CSV -> the last N dispatches in start order: duration and the idle gap in front of each
usage: trace_timeline.py <kernel_trace.csv> [N=40]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
prev = None
tk = tg = 0.0
for r in rows[-n:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sc::', '')[:44]
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%-46s grid=%5d dur=%7.1f us gap_before=%7.1f us" % (name, int(r['Grid_Size_X']) // max(int(r.get('Workgroup_Size_X', 256) or 256), 1), (e - s) / 1e3, gap))
    tk += (e - s) / 1e3
    tg += gap
    prev = e
print("kernels %.1f us, gaps %.1f us" % (tk, tg))
