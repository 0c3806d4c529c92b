# HEAD sanity after the container was re-created: the whole -m gpu suite, the default bench line, a full kernel listing of the train workload
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/c17
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/c17/gpu_tests.log 2>&1; echo "pytest rc $?" && tail -3 gpurun_out/c17/gpu_tests.log &&
timeout -k 10 300 python bench.py > gpurun_out/c17/bench_default.json 2> gpurun_out/c17/bench_default.err && cut -c1-300 gpurun_out/c17/bench_default.json &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c17/train_trace -- python3 $R/bench.py --workload train --ppo-minibatches 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/c17/train_bench.json 2> $R/gpurun_out/c17/train_err.log
find $R/gpurun_out/c17/train_trace -type f ! -name "*_kernel_stats.csv" -delete
cut -c1-400 $R/gpurun_out/c17/train_bench.json
