#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh r01'): the un-profiled bench line,
# the rocprofv3 kernel trace + stats of the same command, and the two PMC passes; then the
# summaries under profiles/ (copied back through gpurun_out/profiles_<tag>/).
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_$TAG.json 2> $O/bench_$TAG.err
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 > $O/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 > $O/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 > $O/prof_write.log 2>&1
python3 $R/tools/make_profile_summary.py $TAG $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_$TAG.json > $O/summary_$TAG.log 2>&1
mkdir -p $O/profiles_$TAG
cp $R/profiles/${TAG}_kernel_stats.csv $R/profiles/${TAG}_summary.md $R/profiles/traffic.json $O/profiles_$TAG/
cp $O/bench_$TAG.json $O/profiles_$TAG/${TAG}_bench.json
# the raw traces are large: keep only the summaries
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
tail -5 $O/bench_$TAG.json | cut -c1-600
