cd $GRAFT_REPO_ROOT
timeout 900 python tools/probe_sched.py wfold_log 25 28 26 27 28 > gpurun_out/r05b_wfold_big_ab.txt 2>&1
cat gpurun_out/r05b_wfold_big_ab.txt
