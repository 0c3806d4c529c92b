cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -15 && \
timeout -k 10 300 python bench.py --workload train --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/bench_train_splitk.json 2> gpurun_out/bench_train_splitk.err && cut -c1-300 gpurun_out/bench_train_splitk.json
