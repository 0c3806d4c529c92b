# round 2, re-entry: full -m gpu suite (incl. the one-rank RCCL rehearsal), smoke, default bench at HEAD
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/c50_gpu_tests.log 2>&1 && \
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c50_smoke.log 2>&1 && \
timeout -k 10 300 python bench.py > gpurun_out/c50_bench.json 2> gpurun_out/c50_bench.err
echo rc=$?
tail -5 gpurun_out/c50_gpu_tests.log; tail -2 gpurun_out/c50_smoke.log; cut -c1-600 gpurun_out/c50_bench.json
