R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --graph 0 --steps 300 --warmup 30 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace_err.log
echo trace rc=$?
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch_err.log
echo fetch rc=$?
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write_err.log
echo write rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_sq_err.log
echo sq rc=$?
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- python3 $R/bench.py --graph 0 --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2> $OUT/pmc_grbm_err.log
echo grbm rc=$?
find $OUT -name "*.csv" | head -30
du -sh $OUT
