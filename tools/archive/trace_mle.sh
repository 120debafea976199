#!/bin/bash
# kernel durations of the single-table paths at n = 28 (rocprofv3 kernel trace of probe_mle.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr_m -- python3 $R/tools/archive/probe_mle.py 28 > $R/gpurun_out/tr_m.log 2>&1
f=$(find $R/gpurun_out/tr_m -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-70s calls=%5s avg=%9.1f us min=%9.1f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf $R/gpurun_out/tr_m
