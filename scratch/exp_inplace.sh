cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 200 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 | cut -c1-160 && \
FDYN_INPLACE_STATE=1 timeout -k 10 200 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 100 --warmup 10 | cut -c1-160
done
