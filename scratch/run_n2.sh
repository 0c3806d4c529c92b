cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --device-index 0 --workload train --batch 16384 --steps 3 --warmup 1 > gpurun_out/bench_n2_train.json 2> gpurun_out/n2t.err && cut -c1-260 gpurun_out/bench_n2_train.json && \
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --device-index 0 --steps 300 --warmup 30 > gpurun_out/bench_n2_rehearsal.json 2> gpurun_out/n2.err && cut -c1-260 gpurun_out/bench_n2_rehearsal.json && \
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err && python -c "
import json; d=json.load(open('gpurun_out/bench_default.json'))
for k,v in d['extras'].items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))"
tail -3 gpurun_out/n2t.err
