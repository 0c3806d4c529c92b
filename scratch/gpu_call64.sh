# A/B on one box: unconditional per-env requests (counted waits survive the prologue join) against the previous build
R=$GRAFT_REPO_ROOT
cd $R
NEW=$R/hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/csrc/libfdyn_hip.so
OLD=$R/scratch/libfdyn_ab_old.so
one() { FDYN_LIB=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1'.split('/')[-1], '$2', 'ms_per_step', round(d['ms_per_step'],5), 'value', '%.4g' % d['value'])"; }
for rep in 1 2; do
  for L in $OLD $NEW; do
    one $L "--workload env"
    one $L "--workload env --batch 1048576 --steps 100 --warmup 10"
    one $L "--workload env_pid"
  done
done > gpurun_out/c64_ab.log 2>&1
cat gpurun_out/c64_ab.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -2
