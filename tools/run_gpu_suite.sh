# on the GPU box (gpurun -- bash tools/run_gpu_suite.sh): the whole -m gpu suite, failures listed
cd $GRAFT_REPO_ROOT
timeout 3400 python -m pytest tests/ -q -m gpu > gpurun_out/gpu_suite.txt 2>&1; echo "suite rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gpu_suite.txt | tail -40
# the multi-device file's own code, rehearsed on this box (every device = GPU 0, the librccl stand-in): tests/test_gpu_multi_device.py
SC_MULTI_DEVICE_REHEARSAL=1 timeout 1800 python -m pytest tests/test_gpu_multi_device.py -q -m gpu > gpurun_out/gpu_suite_rehearsal.txt 2>&1; echo "rehearsal rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/gpu_suite_rehearsal.txt | tail -10
