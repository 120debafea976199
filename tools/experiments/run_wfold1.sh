cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wfold.py -x -q > gpurun_out/r05b_wfold_tests.txt 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r05b_wfold_tests.txt
timeout 600 python tools/probe_sched.py wfold_log 0 25 25 24 23 22 21 > gpurun_out/r05b_wfold_ab.txt 2>&1
timeout 300 python tools/probe_sched.py wfold_log 0 28 28 26 >> gpurun_out/r05b_wfold_ab.txt 2>&1
timeout 300 python tools/probe_sched.py host_tail_log 10 11 25 28 20 >> gpurun_out/r05b_wfold_ab.txt 2>&1
cat gpurun_out/r05b_wfold_ab.txt
