# A/B on ONE box: packed vs scalar env-step arithmetic (default env bench, 1 Mi envs, cascade, rollout), interleaved, twice
R=$GRAFT_REPO_ROOT
cd $R
NEW=$R/hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/csrc/libfdyn_hip.so
OLD=$R/scratch/libfdyn_scalar_env.so
one() { FDYN_LIB=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.split('/')[-1], '$2', 'ms_per_step', round(d['ms_per_step'],5), 'value', '%.4g' % d['value'])"; }
for rep in 1 2; do
  for L in $OLD $NEW; do
    one $L "--workload env"
    one $L "--workload env --batch 1048576 --steps 100 --warmup 10"
    one $L "--workload cascade --steps 200 --warmup 20"
    one $L "--workload rollout --steps 300 --warmup 30"
  done
done > gpurun_out/c58_ab_env.log 2>&1
cat gpurun_out/c58_ab_env.log
