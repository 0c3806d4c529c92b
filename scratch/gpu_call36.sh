cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c36
FDYN_LIB=$GRAFT_REPO_ROOT/scratch/libfdyn_stamps.so timeout -k 10 200 python scratch/phase_stamps.py 65536 1200 normal > gpurun_out/c36/stamps.log 2>&1
grep "wave totals\|p100\|p90 \|p50 \|mean counts" gpurun_out/c36/stamps.log | head -10; grep -A7 "^launch" gpurun_out/c36/stamps.log | tail -8
