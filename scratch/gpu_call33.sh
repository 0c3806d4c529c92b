R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/c35
timeout -k 10 900 python -m pytest tests/test_gpu_training.py -x -q -m gpu -k "trunk or rollout_step or policy_gpu" > gpurun_out/c35/tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/c35/tests.log
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/c35
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload rollout --steps 60 --warmup 6 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/err.log
find $OUT -type f ! -name "*_kernel_stats.csv" ! -name "*.json" ! -name "*.log" -delete
cd $R
timeout -k 10 300 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 200 --warmup 20 | cut -c100-220
