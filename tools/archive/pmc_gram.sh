#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/archive/pmc_gram.sh'): SQ counters of the matrix-core first pass and the fold behind it, one n = 28 proof
# per counter set (rocprofv3 --pmc in its own runs, --kernel-trace only).  -> gpurun_out/r04_gram_sq_counters.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export SC_BENCH_RAMP_MS=0
: > $O/r04_gram_sq_counters.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf $O/gpmc
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/gpmc -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-num-vars 0 > $O/gpmc.log 2>&1
  f=$(find $O/gpmc -name '*counter_collection.csv' | head -1)
  python3 - "$f" >> $O/r04_gram_sq_counters.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r['Kernel_Name']
    if 'gram_pass_kernel' not in n and 'pass_kernel<sc::GoldilocksMont, 4, 2' not in n and 'gram_finish' not in n: continue
    name = n.split('(')[0]
    name = name[name.find('sc::'):]
    agg.setdefault((name, r['Counter_Name']), []).append(float(r['Counter_Value']))
for k, v in agg.items():
    print("%-52s %-26s launches=%d last=%.6g" % (k[0], k[1], len(v), v[-1]))
PY
  rm -rf $O/gpmc
done
cat $O/r04_gram_sq_counters.txt
