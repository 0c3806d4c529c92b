R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/c21
for m in all seq fe 0 all 0; do
FDYN_MFMA_TRAIN=$m timeout -k 10 300 python bench.py --workload train --ppo-minibatches 2 --steps 8 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c21/train_$m.json 2> gpurun_out/c21/train_$m.err || exit 1
echo "$m $(python -c "import json;print(json.load(open('gpurun_out/c21/train_$m.json'))['ms_per_step'])")"
done
