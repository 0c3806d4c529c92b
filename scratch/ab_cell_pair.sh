R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_learned_agent.py -m gpu -x -q > gpurun_out/ab_pair_tests.log 2>&1; rc=$?; tail -3 gpurun_out/ab_pair_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export FDYN_NO_CELL_PAIR=1; else unset FDYN_NO_CELL_PAIR; fi
    echo -n "$v: "
    timeout -k 10 200 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])" || exit 1
  done
done
