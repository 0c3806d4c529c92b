cd $GRAFT_REPO_ROOT
for f in scratch/ubench/m64_*.so; do timeout -k 10 60 python scratch/bench_m64.py $f 2>&1 | grep -v amdgpu || exit 1; done
