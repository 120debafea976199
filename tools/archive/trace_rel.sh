#!/bin/bash
# usage on the GPU box: bash tools/archive/trace_rel.sh 25 [opt=val ...]   -> gpurun_out/trace_rel_n$1.txt
N=${1:-25}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trr_n$N -- python3 $R/tools/archive/probe.py $N 2 "$@" > $R/gpurun_out/trr_n$N.log 2>&1
csv=$(find $R/gpurun_out/trr_n$N -name '*kernel_trace.csv' | head -1)
python3 $R/tools/archive/trace_rel.py $csv > $R/gpurun_out/trace_rel_n$N.txt
rm -rf $R/gpurun_out/trr_n$N
cat $R/gpurun_out/trace_rel_n$N.txt
