#!/bin/bash
# SQ counters of pass_kernel<4,2> on 2^28-entry tables in its three forms (run on the GPU box): is the three-waves-per-SIMD form
# waiting less?  -> gpurun_out/r05_fold42_sq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
: > $O/r05_fold42_sq.txt
for form in "fold_dma=1 fold_oneset=0" "fold_dma=0 fold_oneset=0" "fold_dma=0 fold_oneset=1"; do
  for set in "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_LDS"; do
    rm -rf $O/pmcf
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmcf -- python3 $R/tools/archive/probe.py 28 2 $form > $O/pmcf.log 2>&1
    f=$(find $O/pmcf -name '*counter_collection.csv' | head -1)
    python3 - "$f" "$form" >> $O/r05_fold42_sq.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if 'pass_kernel' not in r['Kernel_Name'] or ', 4, 2, ' not in r['Kernel_Name']: continue
    k = (r['Kernel_Name'].split('(')[0][-44:], r['Counter_Name'])
    agg.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in agg.items():
    print("%-26s %-46s %-22s launches=%d last=%.5g" % (sys.argv[2], k[0], k[1], len(v), v[-1]))
PY
    rm -rf $O/pmcf
  done
done
cat $O/r05_fold42_sq.txt
