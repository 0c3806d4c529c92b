R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/c32
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload rollout --steps 60 --warmup 6 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/err.log
find $OUT -type f ! -name "*_kernel_stats.csv" ! -name "*.json" ! -name "*.log" -delete
f=$(find $OUT/trace -name "*_kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | cut -c1-110 | head -12
