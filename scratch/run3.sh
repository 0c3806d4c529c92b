cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_training.py -m gpu -q -x > gpurun_out/gpu_train_tests.log 2>&1; tail -5 gpurun_out/gpu_train_tests.log
for d in bf16; do
 timeout -k 10 300 python bench.py --workload rollout --policy-dtype $d --steps 100 --warmup 10 --no-cpu-baseline 2>gpurun_out/rollout_$d.err | tail -1 > gpurun_out/bench_rollout_$d.json
 timeout -k 10 300 python bench.py --workload train --policy-dtype $d --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/train_$d.err | tail -1 > gpurun_out/bench_train_$d.json
done
tail -n 3 gpurun_out/rollout_bf16.err gpurun_out/train_bf16.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_rollout_bf16.json')+glob.glob('gpurun_out/bench_train_bf16.json')):
    try:
        d=json.loads(open(f).read())
        print(f, 'value %.3e'%d['value'], 'ms/step %.3f'%d['ms_per_step'], d.get('policy'))
    except Exception as e: print(f, 'bad', e)
PY
