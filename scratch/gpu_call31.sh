cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c31
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_learned_agent.py -x -q -m gpu > gpurun_out/c31/tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/c31/tests.log
timeout -k 10 300 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 200 --warmup 20 > gpurun_out/c31/rollout.json 2> gpurun_out/c31/rollout.err; cut -c100-300 gpurun_out/c31/rollout.json
FDYN_NO_TRUNK=1 timeout -k 10 300 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 200 --warmup 20 > gpurun_out/c31/rollout_notrunk.json 2> gpurun_out/c31/rollout_notrunk.err; cut -c100-300 gpurun_out/c31/rollout_notrunk.json
