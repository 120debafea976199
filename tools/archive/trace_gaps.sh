#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_g -- python3 $R/tools/archive/probe.py 28 2 > $R/gpurun_out/tr_g.log 2>&1
csv=$(find $R/gpurun_out/tr_g -name '*kernel_trace.csv' | head -1)
python3 $R/tools/archive/trace_gaps.py $csv
rm -rf $R/gpurun_out/tr_g
