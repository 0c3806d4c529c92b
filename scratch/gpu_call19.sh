# MFMA BPTT-forward cell: its test, the training tests, the train workload with and without it
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/c19
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_ppo_distributed.py -x -q -m gpu > gpurun_out/c19/tests.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/c19/tests.log
timeout -k 10 300 python bench.py --workload train --ppo-minibatches 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c19/train.json 2> gpurun_out/c19/train.err; cut -c1-260 gpurun_out/c19/train.json; tail -3 gpurun_out/c19/train.err
FDYN_NO_MFMA_TRAIN=1 timeout -k 10 300 python bench.py --workload train --ppo-minibatches 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c19/train_nomfma.json 2> gpurun_out/c19/train_nomfma.err; cut -c1-260 gpurun_out/c19/train_nomfma.json
