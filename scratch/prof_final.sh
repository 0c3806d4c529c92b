R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --graph 0 --steps 300 --warmup 30 --no-cpu-baseline --no-extras > $OUT/trace_bench.json 2> $OUT/trace_err.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p_fetch -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p_write -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p_sq -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/p_grbm -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline --no-extras > /dev/null 2>&1
OUT2=$R/gpurun_out/prof_rollout
rm -rf $OUT2; mkdir -p $OUT2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT2/trace -- python3 $R/bench.py --workload rollout --steps 60 --warmup 6 --no-cpu-baseline --no-extras > $OUT2/trace_bench.json 2> $OUT2/trace_err.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT2/p_mfma -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT2/p_fetch -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT2/p_write -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
ls $OUT $OUT2
cat $OUT/trace_bench.json | cut -c1-200
# keep only the summaries (the raw traces can exceed the 64 MiB merge limit)
find $R/gpurun_out/prof $R/gpurun_out/prof_rollout -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
du -sh $R/gpurun_out
