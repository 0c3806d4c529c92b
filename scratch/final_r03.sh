# round 3: full GPU test suite, smoke, default bench, then rocprofv3 summaries for the env, rollout and train workloads
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1; echo "tests rc=$?"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"
bash scratch/prof_final.sh > gpurun_out/final_prof.log 2>&1; echo "prof rc=$?"
cd /tmp && export TMPDIR=/tmp
OUT3=$R/gpurun_out/prof_train
rm -rf $OUT3; mkdir -p $OUT3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT3/trace -- python3 $R/bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT3/trace_bench.json 2> $OUT3/trace_err.log; echo "train prof rc=$?"
find $OUT3 -type f ! -name "*_kernel_stats.csv" ! -name "*.json" ! -name "*.log" -delete
cd $R
tail -3 gpurun_out/final_gpu_tests.log; tail -2 gpurun_out/final_smoke.log; cut -c1-300 gpurun_out/final_bench.json; du -sh gpurun_out
