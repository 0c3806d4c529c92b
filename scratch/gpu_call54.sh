# packed env step at HEAD (663 VALU / sub-step): whole GPU suite, default bench, rocprofv3 stats + PMC passes (env, rollout)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/c54_gpu_tests.log 2>&1
echo suite rc=$?
timeout -k 10 300 python bench.py > gpurun_out/c54_bench.json 2> gpurun_out/c54_bench.err
echo bench rc=$?
timeout -k 10 500 bash scratch/prof_final.sh > gpurun_out/c54_prof.log 2>&1
echo prof rc=$?
tail -3 gpurun_out/c54_gpu_tests.log; cut -c1-300 gpurun_out/c54_bench.json; tail -5 gpurun_out/c54_prof.log
