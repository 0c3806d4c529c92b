# full-batch parity (65 536 distinct aircraft / envs vs the oracle) + the 4096 scale tests in one process (one drift.json)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity_full_batch.py tests/test_gpu_parity_scale.py -m gpu -x -q -s > gpurun_out/c52_parity.log 2>&1
echo rc=$?
grep -A4 "drift@full" gpurun_out/c52_parity.log; tail -5 gpurun_out/c52_parity.log
