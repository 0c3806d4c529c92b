cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c22
FDYN_LIB=$GRAFT_REPO_ROOT/scratch/libfdyn_stamps.so timeout -k 10 200 python scratch/phase_stamps.py 65536 > gpurun_out/c22/stamps.log 2>&1
grep "wave totals\|p100\|p90 \|mean counts" gpurun_out/c22/stamps.log
