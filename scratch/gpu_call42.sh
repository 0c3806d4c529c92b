cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c42
timeout -k 10 200 python scratch/dbg_graph_grads.py 2>&1 | grep replay | cut -c1-300
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_ppo_distributed.py tests/test_gpu_learned_agent.py -x -q -m gpu > gpurun_out/c42/tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/c42/tests.log
for i in 1 2; do timeout -k 10 300 python bench.py --workload train --ppo-minibatches 2 --steps 8 --warmup 2 --no-cpu-baseline --no-extras | cut -c100-220; done
