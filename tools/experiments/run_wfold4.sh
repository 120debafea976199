cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_wfold.py tests/test_gpu_headline.py -x -q > gpurun_out/r05b_wfold_tests4.txt 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r05b_wfold_tests4.txt
timeout 900 python tools/probe_sched.py wfold_min_log 21 26 25 27 28 > gpurun_out/r05b_wfold5_ab.txt 2>&1
cat gpurun_out/r05b_wfold5_ab.txt
