R=$GRAFT_REPO_ROOT
cd $R
NEW=$R/hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/csrc/libfdyn_hip.so
OLD=$R/scratch/libfdyn_heads_old.so
for rep in 1 2; do for L in $OLD $NEW; do timeout -k 10 100 python scratch/bench_heads.py $L 2>/dev/null; done; done > gpurun_out/c60_heads.log 2>&1
cat gpurun_out/c60_heads.log
