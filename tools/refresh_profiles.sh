#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh r03'): the un-profiled bench lines, the rocprofv3
# kernel trace + stats of the same commands, and the two PMC passes each; then the summaries under profiles/
# (copied back through gpurun_out/profiles_<tag>/).  rocprofv3 always gets the program itself after `--`.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
# the profiled runs skip bench.py's clock-ramp setup phase (the summaries index the proofs of a run by position); the
# un-profiled lines keep it (their own SC_BENCH_RAMP_MS=80)
export SC_BENCH_RAMP_MS=0
export SC_BENCH_SELF_PMC=0   # (the profiled runs below ARE the PMC passes; bench.py must not start its own)
mkdir -p $O/profiles_$TAG
run_one() {   # name, bench args, summariser, extra summariser args
  local WL=$1 EXTRA=$2 SUM=$3 SARG=$4
  SC_BENCH_RAMP_MS=80 python3 $R/bench.py $EXTRA > $O/bench_${TAG}_$WL.json 2> $O/bench_${TAG}_$WL.err
  rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 $EXTRA > $O/prof_stats_$WL.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $EXTRA > $O/prof_fetch_$WL.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 $EXTRA > $O/prof_write_$WL.log 2>&1
  python3 $R/tools/$SUM $TAG $WL $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_$WL.json $SARG > $O/summary_${TAG}_$WL.log 2>&1
  cp $R/profiles/${TAG}_${WL}_kernel_stats.csv $R/profiles/${TAG}_${WL}_summary.md $O/profiles_$TAG/ 2>/dev/null
  cp $O/bench_${TAG}_$WL.json $O/profiles_$TAG/${TAG}_bench_$WL.json
  tail -3 $O/summary_${TAG}_$WL.log
}
# the headline workload: the CPU baseline runs in the un-profiled line only
SC_BENCH_RAMP_MS=80 python3 $R/bench.py > $O/bench_${TAG}_prover.json 2> $O/bench_${TAG}_prover.err
WL=prover
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 > $O/prof_stats_$WL.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 > $O/prof_fetch_$WL.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 > $O/prof_write_$WL.log 2>&1
python3 $R/tools/make_profile_summary.py $TAG $WL $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_$WL.json 28 > $O/summary_${TAG}_$WL.log 2>&1
cp $R/profiles/${TAG}_${WL}_kernel_stats.csv $R/profiles/${TAG}_${WL}_summary.md $O/profiles_$TAG/ 2>/dev/null
cp $O/bench_${TAG}_$WL.json $O/profiles_$TAG/${TAG}_bench_$WL.json
tail -3 $O/summary_${TAG}_$WL.log
# config 3 (n = 26 on one GPU) and the shard of an 8-GPU run (n = 25): the same prover workload at those sizes
for NV in 26 25; do
  SC_BENCH_RAMP_MS=80 python3 $R/bench.py --num-vars $NV --cpu-num-vars $NV > $O/bench_${TAG}_prover$NV.json 2> $O/bench_${TAG}_prover$NV.err
  rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 --num-vars $NV > $O/prof_stats_prover$NV.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --num-vars $NV > $O/prof_fetch_prover$NV.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --num-vars $NV > $O/prof_write_prover$NV.log 2>&1
  python3 $R/tools/make_profile_summary.py ${TAG}n$NV prover $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_prover$NV.json $NV > $O/summary_${TAG}_prover$NV.log 2>&1
  cp $R/profiles/${TAG}n${NV}_prover_kernel_stats.csv $R/profiles/${TAG}n${NV}_prover_summary.md $O/profiles_$TAG/ 2>/dev/null
  cp $O/bench_${TAG}_prover$NV.json $O/profiles_$TAG/${TAG}n${NV}_bench_prover.json
  tail -3 $O/summary_${TAG}_prover$NV.log
done
# the generic-modulus field (p = 2^64-59) at the headline size
SC_BENCH_RAMP_MS=80 python3 $R/bench.py --field generic > $O/bench_${TAG}_prover_generic.json 2> $O/bench_${TAG}_prover_generic.err
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 --field generic > $O/prof_stats_generic.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --field generic > $O/prof_fetch_generic.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --field generic > $O/prof_write_generic.log 2>&1
python3 $R/tools/make_profile_summary.py ${TAG}generic prover $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_prover_generic.json 28 MontGeneric > $O/summary_${TAG}_prover_generic.log 2>&1
cp $R/profiles/${TAG}generic_prover_kernel_stats.csv $R/profiles/${TAG}generic_prover_summary.md $O/profiles_$TAG/ 2>/dev/null
cp $O/bench_${TAG}_prover_generic.json $O/profiles_$TAG/${TAG}generic_bench_prover.json
tail -3 $O/summary_${TAG}_prover_generic.log
# SQ counters of the five-round passes: one proof at n = 25 (the shard of an 8-GPU run) and n = 28
for NV in 25 28; do
  i=0
  : > $O/profiles_$TAG/${TAG}_wgrid_sq_counters_n$NV.txt
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    rm -rf $O/wpmc
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/wpmc -- python3 $R/bench.py --num-vars $NV --steps 1 --warmup 1 --cpu-num-vars 0 > $O/wpmc$i.log 2>&1
    f=$(find $O/wpmc -name '*counter_collection.csv' | head -1)
    python3 - "$f" >> $O/profiles_$TAG/${TAG}_wgrid_sq_counters_n$NV.txt <<'PY'
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
    print("%-58s grid %-8s %-22s launches=%d last=%.5g" % (k[0], k[1], k[2], len(v), v[-1]))
PY
    rm -rf $O/wpmc
  done
done
# the one-process multi-device handle (all entries device 0 on this pool) and the clock ramp
python3 $R/tools/probe_multi.py 28 25 20 12 > $O/profiles_$TAG/${TAG}_multi_handle_probe.txt 2>&1
# config 2 at its stated size
SC_BENCH_RAMP_MS=80 python3 $R/bench.py --workload mle --num-vars 24 > $O/bench_${TAG}_mle24.json 2> $O/bench_${TAG}_mle24.err
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --workload mle --num-vars 24 > $O/prof_stats_mle.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --workload mle --num-vars 24 > $O/prof_fetch_mle.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --workload mle --num-vars 24 > $O/prof_write_mle.log 2>&1
python3 $R/tools/make_profile_summary.py ${TAG}n24 mle $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_mle24.json 24 > $O/summary_${TAG}_mle24.log 2>&1
cp $R/profiles/${TAG}n24_mle_kernel_stats.csv $R/profiles/${TAG}n24_mle_summary.md $O/profiles_$TAG/ 2>/dev/null
cp $O/bench_${TAG}_mle24.json $O/profiles_$TAG/${TAG}n24_bench_mle.json
tail -3 $O/summary_${TAG}_mle24.log
# the callers either side of the path
run_one gkr "--workload gkr" make_widened_summary.py
run_one gnew "--workload gnew" make_widened_summary.py
run_one triangle "--workload triangle" make_widened_summary.py
cp $R/profiles/traffic.json $O/profiles_$TAG/
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
cut -c1-400 $O/bench_${TAG}_prover.json
