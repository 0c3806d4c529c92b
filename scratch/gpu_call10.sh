R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_all.log
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras > gpurun_out/bench_env_new.json 2> gpurun_out/bench_env_new.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/bench_env_20.json 2>> gpurun_out/bench_env_new.err
tail -n 6 gpurun_out/r02_gpu_all.log
grep -E "drift@scale\] (cfg2|env) mixed" -A3 gpurun_out/r02_gpu_all.log | head -n 20
python -c "
import json
for f in ('gpurun_out/bench_env_new.json','gpurun_out/bench_env_20.json'):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['repeats'], d['region_wall_s'], d['roofline']['valu']['frac'])
"
