R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras > gpurun_out/bench_env_new.json 2> gpurun_out/bench_env_new.err
python -c "
import json
d=json.load(open('gpurun_out/bench_env_new.json')); print(d['value'], d['ms_per_step'], d['roofline']['valu']['frac'])"
bash scratch/prof_r02_kernels.sh env r02_env > gpurun_out/prof_env.log 2>&1
