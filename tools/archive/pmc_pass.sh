#!/bin/bash
# SQ counters of the pass kernels of one n=28 proof (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf $O/pmc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc$i -- python3 $R/tools/archive/probe.py 28 2 > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY' > $O/pmc$i.txt 2>&1
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if 'pass_kernel' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X',''), r['Counter_Name'])
    agg.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k[0], k[1], k[2], "n=%d last=%.4g" % (len(v), v[-1]))
PY
  rm -rf $O/pmc$i
done
cat $O/pmc1.txt $O/pmc2.txt | grep -E "0, 3, 1>|3, 2, 3>" | grep -v "n=0"
