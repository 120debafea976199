#!/bin/bash
# rocprofv3 kernel trace of proofs at n = $1 (default 25): every launch of the last proof with the idle time in front of it.
# usage on the GPU box: bash tools/archive/trace_n.sh 25 [opt=val ...]     -> gpurun_out/trace_n$1.txt
N=${1:-25}; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_n$N -- python3 $R/tools/archive/probe.py $N 2 "$@" > $R/gpurun_out/tr_n$N.log 2>&1
csv=$(find $R/gpurun_out/tr_n$N -name '*kernel_trace.csv' | head -1)
python3 $R/tools/archive/trace_gaps.py $csv > $R/gpurun_out/trace_n$N.txt
tail -3 $R/gpurun_out/tr_n$N.log >> $R/gpurun_out/trace_n$N.txt
rm -rf $R/gpurun_out/tr_n$N
cat $R/gpurun_out/trace_n$N.txt
