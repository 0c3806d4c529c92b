cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -q -s > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log; grep drift gpurun_out/gpu_tests.log
for p in mixed f32 f64; do
  timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_env_${p}.json
done
for w in physics cascade env_pid; do
  timeout -k 10 200 python bench.py --workload $w --precision mixed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${w}_mixed.json
done
python scratch/show.py gpurun_out/bench_env_mixed.json gpurun_out/bench_env_f32.json gpurun_out/bench_env_f64.json gpurun_out/bench_physics_mixed.json gpurun_out/bench_cascade_mixed.json gpurun_out/bench_env_pid_mixed.json
