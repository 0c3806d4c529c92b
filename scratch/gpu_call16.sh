R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT2=$R/gpurun_out/prof_r02e_rollout
rm -rf $OUT2; mkdir -p $OUT2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT2/trace -- python3 $R/bench.py --workload rollout --steps 60 --warmup 6 --no-cpu-baseline --no-extras > $OUT2/trace_bench.json 2> $OUT2/trace_err.log &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT2/p_mfma -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
find $OUT2 -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
echo done
