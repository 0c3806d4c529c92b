cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_training.py -x -q -k "policy_features" 2>&1 | tail -3 && \
for f in scratch/ubench/fe64_*.so; do timeout -k 10 60 python scratch/bench_fe64.py $f 2>&1 | grep -v amdgpu || exit 1; done
