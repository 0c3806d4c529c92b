cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
# two ranks on one GPU over gloo: rehearsal of the N > 1 launch path (the driver's N > 1 runs use nccl, one rank per GPU)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --device-index 0 --workload train --batch 16384 --steps 3 --warmup 1 > gpurun_out/bench_n2_train.json 2> gpurun_out/n2t.err && wc -l gpurun_out/bench_n2_train.json && cut -c1-200 gpurun_out/bench_n2_train.json && \
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --device-index 0 --steps 300 --warmup 30 > gpurun_out/bench_n2_rehearsal.json 2> gpurun_out/n2.err && wc -l gpurun_out/bench_n2_rehearsal.json && cut -c1-200 gpurun_out/bench_n2_rehearsal.json && \
timeout -k 10 300 python bench.py --steps 300 --warmup 30 --no-extras > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err && wc -l gpurun_out/bench_default.json && python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
echo rc=$?
grep -c Gloo gpurun_out/n2t.err
