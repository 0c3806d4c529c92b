cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c24
FDYN_LIB=$GRAFT_REPO_ROOT/scratch/libfdyn_stamps.so timeout -k 10 200 python scratch/phase_stamps.py 65536 1200 > gpurun_out/c24/stamps.log 2>&1
grep "wave totals\|p100\|p90 \|p50 \|mean counts" gpurun_out/c24/stamps.log | head -15
for s in 20 100 1000; do timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps $s --warmup 10 | cut -c100-260; done
