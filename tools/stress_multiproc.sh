#!/bin/bash
# the 8-process peer-transport tests N times in a row, single attempt each:
#   tools/stress_multiproc.sh [N]      -> gpurun_out/stress_multiproc.txt
N=${1:-20}
out=gpurun_out/stress_multiproc.txt
mkdir -p gpurun_out
: > $out
fail=0
for i in $(seq 1 $N); do
  s=$(date +%s.%N)
  if python -m pytest tests/test_gpu_00_multiprocess.py -x -q -m gpu -k "test_peer_transport_processes_one_device and 8 or test_bench_eight_ranks" > gpurun_out/stress_one.log 2>&1; then r=ok; else r=FAIL; fail=$((fail+1)); cp gpurun_out/stress_one.log gpurun_out/stress_fail_$i.log; fi
  e=$(date +%s.%N)
  echo "run $i: $r $(python3 -c "print(round($e - $s, 1))") s" >> $out
done
echo "failures: $fail of $N" >> $out
tail -3 $out
