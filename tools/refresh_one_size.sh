TAG=${1:-r05}; NV=${2:-25}   # tools/refresh_one_size.sh <tag> <n>: one prover size of refresh_profiles.sh alone (un-profiled line, stats, two PMC passes, summary)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; export SC_BENCH_RAMP_MS=0
mkdir -p $O/profiles_$TAG
SC_BENCH_RAMP_MS=80 python3 $R/bench.py --num-vars $NV --cpu-num-vars $NV > $O/bench_${TAG}_prover$NV.json 2> $O/bench_${TAG}_prover$NV.err
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-num-vars 0 --num-vars $NV > $O/prof_stats_prover$NV.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --num-vars $NV > $O/prof_fetch_prover$NV.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-num-vars 0 --num-vars $NV > $O/prof_write_prover$NV.log 2>&1
find $O/prof_stats $O/prof_fetch $O/prof_write -name '*.csv' | head -20
python3 $R/tools/make_profile_summary.py ${TAG}n$NV prover $O/prof_stats $O/prof_fetch $O/prof_write $O/bench_${TAG}_prover$NV.json $NV > $O/summary_${TAG}_prover$NV.log 2>&1
cp $R/profiles/${TAG}n${NV}_prover_kernel_stats.csv $R/profiles/${TAG}n${NV}_prover_summary.md $O/profiles_$TAG/ 2>/dev/null
cp $O/bench_${TAG}_prover$NV.json $O/profiles_$TAG/${TAG}n${NV}_bench_prover.json
tail -30 $O/summary_${TAG}_prover$NV.log
