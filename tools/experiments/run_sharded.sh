cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_00_multiprocess.py -q -m gpu > gpurun_out/r05b_sharded.txt 2>&1; echo "rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r05b_sharded.txt | tail -20
