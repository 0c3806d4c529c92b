R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 && \
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1 && \
timeout -k 10 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err && \
bash scratch/prof_final.sh > gpurun_out/final_prof.log 2>&1
echo rc=$?
tail -3 gpurun_out/final_gpu_tests.log; cat gpurun_out/final_smoke.log | tail -2; cut -c1-400 gpurun_out/final_bench.json
