#!/bin/bash
# Run ON THE GPU BOX: the generic-modulus field (p = 2^64-59) at n = 28, round 3's arithmetic (every product reduced; a second
# build of the library with the old field.hpp, tools/build/libsumcheck_hip_r03field.so) against the lazy 160-bit sums, and the
# SQ counters of the new kernels.  gpurun -- 'bash tools/generic_ab.sh'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
summ() { python3 - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("ms_per_step %.4f  mul-adds/s %.4g  dominant frac %.3f  step frac_of_kernel_time %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["step"]["frac_of_kernel_time"]))
for k in d["roofline"]["kernels"]:
    print("   %-72s %8.1f us %7.0f GB/s  %.3f of peak" % (k["kernel"], k["avg_us"], k["GBps"] or 0, (k["GBps"] or 0) / 8000.0))
PY
}
if [ -f $R/tools/build/libsumcheck_hip_r03field.so ]; then
  echo "== round 3 arithmetic (one Montgomery reduction per product)"
  SUMCHECK_HIP_LIB=$R/tools/build/libsumcheck_hip_r03field.so python3 $R/bench.py --field generic --steps 10 --warmup 3 --cpu-num-vars 0 > $O/generic_old.json 2> $O/generic_old.err
  summ $O/generic_old.json
fi
echo "== round 4 (lazy 160-bit sums, one reduction per thread)"
python3 $R/bench.py --field generic --steps 10 --warmup 3 --cpu-num-vars 0 > $O/generic_new.json 2> $O/generic_new.err
summ $O/generic_new.json
echo "== Goldilocks on the same box"
python3 $R/bench.py --steps 10 --warmup 3 --cpu-num-vars 0 > $O/generic_gold.json 2> $O/generic_gold.err
summ $O/generic_gold.json
echo "== SQ counters, generic field, one proof"
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  i=$((i+1))
  rm -rf $O/gpmc$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/gpmc$i -- python3 $R/bench.py --field generic --steps 1 --warmup 1 --cpu-num-vars 0 > $O/gpmc$i.log 2>&1
  f=$(find $O/gpmc$i -name '*counter_collection.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    if 'pass_kernel' not in r['Kernel_Name']: continue
    name = r['Kernel_Name'].split('(')[0]
    name = name[name.find('sc::'):]
    k = (name, r.get('Grid_Size', r.get('Grid_Size_X', '')), r['Counter_Name'])
    agg.setdefault(k, []).append(float(r['Counter_Value']))
for k, v in agg.items():
    print("%-58s grid %-8s %-22s n=%d last=%.5g" % (k[0], k[1], k[2], len(v), v[-1]))
PY
  rm -rf $O/gpmc$i
done
