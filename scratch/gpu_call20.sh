R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/c38
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c38/train_trace -- python3 $R/bench.py --workload train --ppo-minibatches 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/c38/train_bench.json 2> $R/gpurun_out/c38/train_err.log
find $R/gpurun_out/c38/train_trace -type f ! -name "*_kernel_stats.csv" -delete
echo done
