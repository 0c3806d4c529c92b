set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
for p in mixed f32 f64; do
  timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline --graph 0 2>/dev/null | tail -1 > gpurun_out/bench_env_${p}_eager.json
  timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline --graph 1 2>/dev/null | tail -1 > gpurun_out/bench_env_${p}_graph.json
done
for w in physics cascade env_pid; do
  timeout -k 10 200 python bench.py --workload $w --precision mixed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${w}_mixed.json
  timeout -k 10 200 python bench.py --workload $w --precision f32 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_${w}_f32.json
done
timeout -k 10 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json
cat gpurun_out/bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception as e: print('bad line', l[:100]); continue
    print(d['config']['precision'], d['config']['launch'], d['config']['workload'][:40], '| value %.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel_ms %.4f'%d['roofline']['kernel_ms'], 'hbm frac %.4f'%d['roofline']['frac'], 'valu frac %.3f'%d['compute']['frac'], d.get('cpu_baseline',{}).get('value'))
"
