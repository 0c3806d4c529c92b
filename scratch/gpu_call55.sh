# packed epilogue arithmetic in policy_fe64 / lstm_mfma64: policy tests, then rollout + train bench
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_learned_agent.py -m gpu -x -q > gpurun_out/c55_tests.log 2>&1
echo tests rc=$?
timeout -k 10 200 python bench.py --workload rollout --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/c55_rollout.json 2> gpurun_out/c55_rollout.err
echo rollout rc=$?
timeout -k 10 200 python bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/c55_train.json 2> gpurun_out/c55_train.err
echo train rc=$?
tail -3 gpurun_out/c55_tests.log; cut -c1-250 gpurun_out/c55_rollout.json; echo; cut -c1-250 gpurun_out/c55_train.json
