#!/bin/bash
# per-launch pass-kernel durations of an n=28 proof under both first-pass schedules
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 2 3; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_f$f -- python3 $R/tools/archive/probe.py 28 2 first_pass_vars=$f > $R/gpurun_out/tr_f$f.log 2>&1
  csv=$(find $R/gpurun_out/tr_f$f -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/archive/trace_summary.py $csv 28 > $R/gpurun_out/tr_f$f.txt 2>&1
  rm -rf $R/gpurun_out/tr_f$f
done
cat $R/gpurun_out/tr_f2.txt $R/gpurun_out/tr_f3.txt
