# on the GPU box: the differential fuzz and the back-to-back stress runs of the final code (gpurun -- bash tools/run_fuzz_stress.sh [seconds])
cd $GRAFT_REPO_ROOT
S=${1:-300}
timeout $((S + 120)) python tools/fuzz_diff.py $S 11 18 > gpurun_out/fuzz_diff.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/fuzz_diff.txt
timeout $((S + 120)) python tools/fuzz_diff.py $S 12 21 >> gpurun_out/fuzz_diff.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/fuzz_diff.txt
timeout 900 python tools/stress_gram.py > gpurun_out/stress_gram.txt 2>&1; echo "stress_gram rc=$?"; tail -2 gpurun_out/stress_gram.txt
timeout 900 python tools/stress_grid.py > gpurun_out/stress_grid.txt 2>&1; echo "stress_grid rc=$?"; tail -2 gpurun_out/stress_grid.txt
timeout 900 python tools/soak_gram_multi.py > gpurun_out/soak_gram_multi.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/soak_gram_multi.txt
timeout 900 python tools/stress_handover.py > gpurun_out/stress_handover.txt 2>&1; echo "stress_handover rc=$?"; tail -2 gpurun_out/stress_handover.txt
