#!/bin/bash
# A/B of library options on one box: tools/ab_options.sh "first_ring=0" "first_ring=1" [rounds] [bench args...]
# alternates `python bench.py` runs with SC_BENCH_OPTIONS=<A> / <B> and prints ms per step and the three longest kernels
A=$1; B=$2; R=${3:-3}; shift 3
for i in $(seq $R); do
  for opt in "$A" "$B"; do
    SC_BENCH_OPTIONS=$opt python bench.py --cpu-num-vars 0 --steps 200 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
ks = d['roofline'].get('kernels') or d['roofline']['step'].get('kernels')
print('%-24s mean %.4f median %.4f kernel %.4f | ' % ('$opt', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['step']['kernel_ms']) +
      ' '.join('%.1f' % k['avg_us'] for k in ks[:4]))"
  done
done
