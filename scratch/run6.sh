cd $GRAFT_REPO_ROOT
# N=2 rehearsal on ONE GPU: two ranks share cuda:0, gloo rendezvous (the driver's real N>1 runs use nccl, one GPU per rank)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 200 --warmup 20 --backend gloo --device-index 0 2> gpurun_out/n2.err | tail -1 > gpurun_out/bench_n2_rehearsal.json
tail -n 2 gpurun_out/n2.err
python scratch/show.py gpurun_out/bench_n2_rehearsal.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --workload train --backend gloo --device-index 0 --batch 8192 2> gpurun_out/n2t.err | tail -1 > gpurun_out/bench_n2_train.json
tail -n 2 gpurun_out/n2t.err
python scratch/show.py gpurun_out/bench_n2_train.json
timeout -k 10 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json
