cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c30
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_learned_agent.py tests/test_ppo_distributed.py -x -q -m gpu > gpurun_out/c30/tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/c30/tests.log
timeout -k 10 300 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 200 --warmup 20 > gpurun_out/c30/rollout.json 2> gpurun_out/c30/rollout.err; cut -c100-300 gpurun_out/c30/rollout.json
timeout -k 10 300 python bench.py --workload train --ppo-minibatches 2 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c30/train.json 2> gpurun_out/c30/train.err; cut -c100-300 gpurun_out/c30/train.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c30/env.json 2> gpurun_out/c30/env.err; cut -c100-300 gpurun_out/c30/env.json
