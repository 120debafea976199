cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r05b_gpu_suite.txt 2>&1; echo "suite rc=$?"
tail -15 gpurun_out/r05b_gpu_suite.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err; echo "bench rc=$?"
cat gpurun_out/r05b_bench.json
