cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -15 && \
timeout -k 10 300 python bench.py --workload train --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/bench_train_graph.json 2> gpurun_out/bench_train_graph.err && cut -c1-300 gpurun_out/bench_train_graph.json && \
timeout -k 10 300 python bench.py --workload train --batch 16384 --ppo-steps 64 --ppo-epochs 4 --ppo-minibatches 8 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_train_cfg4.json 2> gpurun_out/bench_train_cfg4.err && cut -c1-300 gpurun_out/bench_train_cfg4.json
tail -3 gpurun_out/bench_train_graph.err
