# on the GPU box: gpurun -- bash tools/run_gpu_files.sh tests/test_gpu_sharded.py ...   (the named test files of the -m gpu suite)
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest "$@" -q -m gpu > gpurun_out/gpu_files.txt 2>&1; echo "rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gpu_files.txt | tail -20
