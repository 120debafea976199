#!/bin/bash
# on the GPU box: TLB counters of the fold pass in fast and slow contexts of one process (tools/archive/probe_modes_pmc.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SETS=("TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_READ_REQ_LATENCY_sum TCC_WRITE_REQ_LATENCY_sum" "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum")
for set in "${SETS[@]}"; do
  tag=$(echo $set | tr ' ' '+')
  rm -rf $O/prof_modes
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/prof_modes -- python3 $R/tools/archive/probe_modes_pmc.py 8 > $O/modes_$tag.log 2>&1
  echo "== $set"; grep "context" $O/modes_$tag.log
  python3 - "$O/prof_modes" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
# per dispatch: kernel name, counter values
by = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "pass_kernel" not in k or "Li3ELi2E" not in k and "3, 2" not in k:
        continue
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
per_ctx = [ids[i:i + 4] for i in range(0, len(ids), 4)]          # four proofs per context
for ci, grp in enumerate(per_ctx):
    c = by[grp[-1]]
    print("   context %d (dispatch %d)  " % (ci, grp[-1]) + "  ".join("%s=%.5g" % (k, v) for k, v in sorted(c.items())))
PY
done
rm -rf $O/prof_modes
