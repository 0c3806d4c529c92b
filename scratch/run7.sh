cd $GRAFT_REPO_ROOT
echo "== rollout b8192"; timeout -k 10 200 python bench.py --workload rollout --batch 8192 --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== train fp32 b8192"; timeout -k 10 200 python bench.py --workload train --policy-dtype fp32 --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== train bf16 b8192"; timeout -k 10 200 python bench.py --workload train --batch 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
