cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_training.py -x -q -k "mfma" 2>&1 | tail -2 && \
for f in scratch/ubench/m64_*.so; do timeout -k 10 60 python scratch/bench_m64.py $f 2>&1 | grep -v amdgpu || exit 1; done
