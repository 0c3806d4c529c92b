R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_agents.py tests/test_gpu_dropin.py tests/test_gpu_parity_scale.py -q -m gpu -s > gpurun_out/r02_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r02_parity.log
bash scratch/prof_r02_kernels.sh cascade r02b_cascade > gpurun_out/prof_cascade.log 2>&1
bash scratch/prof_r02_kernels.sh physics r02b_physics > gpurun_out/prof_physics.log 2>&1
grep -E "drift|passed|failed|rc=|Error|assert " gpurun_out/r02_parity.log | tail -n 50
tail -n 3 gpurun_out/prof_cascade.log gpurun_out/prof_physics.log
