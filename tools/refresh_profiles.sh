#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh r02'): the un-profiled bench lines, the rocprofv3
# kernel trace + stats of the same commands, and the two PMC passes each; then the summaries under profiles/
# (copied back through gpurun_out/profiles_<tag>/).  rocprofv3 always gets the program itself after `--`.
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/profiles_$TAG
for WL in prover mle; do
  if [ $WL = mle ]; then EXTRA="--workload mle --num-vars 28"; NV=28; else EXTRA=""; NV=28; fi
  python3 $R/bench.py $EXTRA > $O/bench_${TAG}_$WL.json 2> $O/bench_${TAG}_$WL.err
  rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 $EXTRA > $O/prof_stats_$WL.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 $EXTRA > $O/prof_fetch_$WL.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 $EXTRA > $O/prof_write_$WL.log 2>&1
  python3 $R/tools/make_profile_summary.py $TAG $WL $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_$WL.json $NV > $O/summary_${TAG}_$WL.log 2>&1
  cp $R/profiles/${TAG}_${WL}_kernel_stats.csv $R/profiles/${TAG}_${WL}_summary.md $O/profiles_$TAG/ 2>/dev/null
  cp $O/bench_${TAG}_$WL.json $O/profiles_$TAG/${TAG}_bench_$WL.json
done
cp $R/profiles/traffic.json $O/profiles_$TAG/
# widened rows: one GKR W layer (k = 13) and one triangle proof (1024 vertices) under the kernel trace
rm -rf $O/prof_wide
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wide -- python3 $R/tools/probe_log.py 13 10 > $O/profiles_$TAG/${TAG}_widened_launch_log.txt 2>&1
f=$(find $O/prof_wide -name '*_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/profiles_$TAG/${TAG}_widened_kernel_stats.csv
# the raw traces are large: keep only the summaries
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_wide
tail -3 $O/summary_${TAG}_prover.log; tail -3 $O/summary_${TAG}_mle.log
cut -c1-400 $O/bench_${TAG}_prover.json
