cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_training.py -x -q -k "policy_features or fused_rollout or policy_gpu" 2>&1 | tail -15 && \
timeout -k 10 120 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 | cut -c1-220 && \
FDYN_NO_FE64=1 timeout -k 10 120 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 | cut -c1-220
