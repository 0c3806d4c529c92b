# rocprofv3 passes at HEAD for the cascade, physics and rollout workloads (kernel-trace stats + separate PMC passes)
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 bash scratch/prof_r02_kernels.sh cascade r02i_cascade > gpurun_out/c56_cascade.log 2>&1; echo cascade rc=$?
timeout -k 10 400 bash scratch/prof_r02_kernels.sh physics r02i_physics > gpurun_out/c56_physics.log 2>&1; echo physics rc=$?
cd /tmp && export TMPDIR=/tmp
OUT2=$R/gpurun_out/prof_rollout
rm -rf $OUT2; mkdir -p $OUT2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT2/trace -- python3 $R/bench.py --workload rollout --steps 60 --warmup 6 --no-cpu-baseline --no-extras > $OUT2/trace_bench.json 2> $OUT2/trace_err.log; echo rollout trace rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT2/p_mfma -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo rollout pmc rc=$?
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT2/p_insts -- python3 $R/bench.py --workload rollout --steps 30 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1; echo rollout pmc2 rc=$?
find $OUT2 -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
du -sh $R/gpurun_out
