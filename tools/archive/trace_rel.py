"""rocprofv3 kernel-trace CSV -> the launches of the LAST proof with start / end relative to the proof's first kernel (us):
shows kernels that overlap (pre-launched passes are resident while their predecessor runs)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'pass' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last proof = from the last gram/first pass on
starts = [i for i, r in enumerate(rows) if 'gram_pass' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # 1 = the last proof (tools/probe.py times its kernels: no pre-launch), 2 = the one before
idx = starts[-which]
last = rows[idx:(starts[-which + 1] if which > 1 else len(rows))]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][-52:]
    print("%-54s grid=%5d  start %8.1f  end %8.1f  dur %7.1f  queue %s" % (name, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?')))
