# end-of-session check of the committed tree: whole GPU suite, smoke, default bench
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/c65_gpu_tests.log 2>&1; echo suite rc=$?
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c65_smoke.log 2>&1; echo smoke rc=$?
timeout -k 10 300 python bench.py > gpurun_out/c65_bench.json 2> gpurun_out/c65_bench.err; echo bench rc=$?
tail -2 gpurun_out/c65_gpu_tests.log; tail -1 gpurun_out/c65_smoke.log; cut -c1-260 gpurun_out/c65_bench.json
