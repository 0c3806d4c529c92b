R=$GRAFT_REPO_ROOT
cd $R
FDYN_LIB=$R/scratch/libfdyn_stamps.so timeout -k 10 200 python scratch/phase_stamps.py 65536 200 bench > gpurun_out/c61_stamps.log 2>&1
echo rc=$?
tail -40 gpurun_out/c61_stamps.log
