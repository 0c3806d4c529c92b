R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_learned_agent.py -m gpu -x -q > gpurun_out/ab_trunk_tests.log 2>&1; rc=$?; tail -3 gpurun_out/ab_trunk_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
bash scratch/ab_trunk_heads_prof.sh
