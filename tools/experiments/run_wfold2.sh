cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wfold.py -x -q > gpurun_out/r05b_wfold_tests2.txt 2>&1; echo "tests rc=$?" 
tail -5 gpurun_out/r05b_wfold_tests2.txt
timeout 600 python tools/probe_sched.py wfold_mix 0 1 25 24 22 > gpurun_out/r05b_wfold_mix_ab.txt 2>&1
cat gpurun_out/r05b_wfold_mix_ab.txt
