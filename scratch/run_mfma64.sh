cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_training.py -x -q -k "mfma or fused_rollout or policy_gpu" 2>&1 | tail -4 && \
timeout -k 10 120 python scratch/bench_lstm.py 2>&1 | grep -v amdgpu | head -1 && \
FDYN_MFMA64=0 timeout -k 10 120 python scratch/bench_lstm.py 2>&1 | grep -v amdgpu | head -1 && \
timeout -k 10 120 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 | cut -c1-200
