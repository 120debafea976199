#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh r03'): the un-profiled bench lines, the rocprofv3
# kernel trace + stats of the same commands, and the two PMC passes each; then the summaries under profiles/
# (copied back through gpurun_out/profiles_<tag>/).  rocprofv3 always gets the program itself after `--`.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/profiles_$TAG
run_one() {   # name, bench args, summariser, extra summariser args
  local WL=$1 EXTRA=$2 SUM=$3 SARG=$4
  python3 $R/bench.py $EXTRA > $O/bench_${TAG}_$WL.json 2> $O/bench_${TAG}_$WL.err
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
python3 $R/bench.py > $O/bench_${TAG}_prover.json 2> $O/bench_${TAG}_prover.err
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
  python3 $R/bench.py --num-vars $NV --cpu-num-vars $NV > $O/bench_${TAG}_prover$NV.json 2> $O/bench_${TAG}_prover$NV.err
  rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 --num-vars $NV > $O/prof_stats_prover$NV.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --num-vars $NV > $O/prof_fetch_prover$NV.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --num-vars $NV > $O/prof_write_prover$NV.log 2>&1
  python3 $R/tools/make_profile_summary.py ${TAG}n$NV prover $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_prover$NV.json $NV > $O/summary_${TAG}_prover$NV.log 2>&1
  cp $R/profiles/${TAG}n${NV}_prover_kernel_stats.csv $R/profiles/${TAG}n${NV}_prover_summary.md $O/profiles_$TAG/ 2>/dev/null
  cp $O/bench_${TAG}_prover$NV.json $O/profiles_$TAG/${TAG}n${NV}_bench_prover.json
  tail -3 $O/summary_${TAG}_prover$NV.log
done
# config 2 at its stated size
python3 $R/bench.py --workload mle --num-vars 24 > $O/bench_${TAG}_mle24.json 2> $O/bench_${TAG}_mle24.err
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
