TAG=r02n24
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/profiles_$TAG
EXTRA="--workload mle --num-vars 24"
python3 $R/bench.py $EXTRA > $O/bench_${TAG}_mle.json 2> $O/bench_${TAG}_mle.err
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 $EXTRA > $O/prof_stats_mle.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 $EXTRA > $O/prof_fetch_mle.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 $EXTRA > $O/prof_write_mle.log 2>&1
python3 $R/tools/make_profile_summary.py $TAG mle $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_mle.json 24 > $O/summary_${TAG}_mle.log 2>&1
cp $R/profiles/${TAG}_mle_kernel_stats.csv $R/profiles/${TAG}_mle_summary.md $O/profiles_$TAG/ 2>/dev/null
cp $O/bench_${TAG}_mle.json $O/profiles_$TAG/${TAG}_bench_mle.json
cp $R/profiles/traffic.json $O/profiles_$TAG/
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
tail -5 $O/summary_${TAG}_mle.log
