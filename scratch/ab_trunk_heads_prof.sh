# per-kernel durations of the rollout step, heads as MFMA layer (tree) vs per-lane dot products (old .so)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export FDYN_LIB=$R/scratch/libfdyn_trunk_4wave.so; else unset FDYN_LIB; fi
  O=$R/gpurun_out/prof_ab_trunk_$v; rm -rf $O; mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 > $O/bench.json 2> $O/err.log || exit 1
  echo "== $v"; f=$(find $O -name "*_kernel_stats.csv" | head -1); head -8 $f | cut -d, -f1-4 | cut -c1-160
  find $O -type f ! -name "*_kernel_stats.csv" ! -name "*.json" ! -name "*.log" -delete
done
