cd $GRAFT_REPO_ROOT
timeout 3400 python -m pytest tests/ -q -m gpu > gpurun_out/r05b_gpu_suite.txt 2>&1; echo "suite rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05b_gpu_suite.txt | tail -40
