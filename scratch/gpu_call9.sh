R=$GRAFT_REPO_ROOT
cd $R
bash scratch/prof_r02_kernels.sh env r02_env > gpurun_out/prof_env.log 2>&1
bash scratch/prof_r02_kernels.sh cascade r02_cascade > gpurun_out/prof_cascade.log 2>&1
bash scratch/prof_r02_kernels.sh physics r02_physics > gpurun_out/prof_physics.log 2>&1
cd /tmp && export TMPDIR=/tmp
OUT2=$R/gpurun_out/prof_r02_rollout
rm -rf $OUT2; mkdir -p $OUT2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT2/trace -- python3 $R/bench.py --workload rollout --steps 60 --warmup 6 --no-cpu-baseline --no-extras > $OUT2/trace_bench.json 2> $OUT2/trace_err.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT2/p_mfma -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
OUT3=$R/gpurun_out/prof_r02_train
rm -rf $OUT3; mkdir -p $OUT3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT3/trace -- python3 $R/bench.py --workload train --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT3/trace_bench.json 2> $OUT3/trace_err.log
find $OUT2 $OUT3 -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
cd $R
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/bench_default.json
