cd $GRAFT_REPO_ROOT
echo "--- product"; timeout -k 10 120 python scratch/bench_lstm.py 2>&1 | grep "mfma fused"
for v in SKELETON SKELETONNOWEIGHTS; do echo "--- $v"; FDYN_LIB=$GRAFT_REPO_ROOT/scratch/ubench/libfdyn_$v.so timeout -k 10 120 python scratch/bench_lstm.py 2>&1 | grep "mfma fused"; done
