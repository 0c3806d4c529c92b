# A/B on ONE box: packed vs scalar gate epilogues (policy_fe64, lstm_mfma64): stand-alone kernel timings and the rollout bench, twice each, interleaved
R=$GRAFT_REPO_ROOT
cd $R
NEW=$R/hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/csrc/libfdyn_hip.so
OLD=$R/scratch/libfdyn_old_epilogue.so
for rep in 1 2; do
  for L in $OLD $NEW; do
    timeout -k 10 100 python scratch/bench_m64.py $L
    timeout -k 10 100 python scratch/bench_fe64.py $L
  done
done > gpurun_out/c57_ab_kernels.log 2>&1
for rep in 1 2; do
  for L in $OLD $NEW; do
    FDYN_LIB=$L timeout -k 10 200 python bench.py --workload rollout --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-1], d['ms_per_step'])"
  done
done > gpurun_out/c57_ab_rollout.log 2>&1
cat gpurun_out/c57_ab_kernels.log gpurun_out/c57_ab_rollout.log
