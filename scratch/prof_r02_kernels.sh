# rocprofv3 passes for one bench workload: kernel trace + stats, then separate PMC passes (never combined with traces
# other than --kernel-trace).  usage: prof_r02_kernels.sh <workload> <tag> [extra bench args]
R=$GRAFT_REPO_ROOT
WL=$1; TAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
B="python3 $R/bench.py --workload $WL --graph 0 --no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 300 --warmup 30 > $OUT/trace_bench.json 2> $OUT/trace_err.log &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p_fetch -- $B --steps 100 --warmup 10 > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p_write -- $B --steps 100 --warmup 10 > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/p_sq -- $B --steps 100 --warmup 10 > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p_grbm -- $B --steps 100 --warmup 10 > /dev/null 2>&1
find $OUT -type f ! -name "*_kernel_stats.csv" ! -name "*_counter_collection.csv" ! -name "*.json" ! -name "*.log" -delete
cut -c1-400 $OUT/trace_bench.json
