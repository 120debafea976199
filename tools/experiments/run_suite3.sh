cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_abi_misc.py tests/test_gpu_grid_pass.py tests/test_gpu_headline.py tests/test_gpu_multi.py tests/test_gpu_wfold.py -q -m gpu > gpurun_out/r05b_gpu_suite3.txt 2>&1; echo "suite rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05b_gpu_suite3.txt | tail -20
bash tools/ab_options.sh "wfold_log=0,host_tail_log=10" "" 3 > gpurun_out/r05b_ab_n28.txt 2>&1
cat gpurun_out/r05b_ab_n28.txt
bash tools/ab_options.sh "wfold_log=0,host_tail_log=10" "" 3 --num-vars 25 > gpurun_out/r05b_ab_n25.txt 2>&1
cat gpurun_out/r05b_ab_n25.txt
