#!/bin/bash
# timeline (kernel durations and idle gaps) of a GKR W layer proof (k = 13) and a triangle proof (k = 10):
# the last dispatches of tools/probe_log.py under the rocprofv3 kernel trace -> gpurun_out/trace_widened.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr_w
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_w -- python3 $R/tools/archive/probe_log.py ${1:-13} ${2:-10} > /tmp/tr_w.log 2>&1
csv=$(find /tmp/tr_w -name '*kernel_trace.csv' | head -1)
python3 $R/tools/archive/trace_timeline.py $csv ${3:-120} > $R/gpurun_out/trace_widened.txt
tail -45 $R/gpurun_out/trace_widened.txt
