# on the GPU box (gpurun -- bash tools/run_gpu_suite.sh): the whole -m gpu suite, failures listed
cd $GRAFT_REPO_ROOT
timeout 3400 python -m pytest tests/ -q -m gpu > gpurun_out/gpu_suite.txt 2>&1; echo "suite rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gpu_suite.txt | tail -40
