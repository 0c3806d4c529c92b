# three-gate zero-state layer + bias partial sums + ReLU epilogue: training tests, then the train workload (2 and 10 epochs)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/c18
timeout -k 10 900 python -m pytest tests/test_gpu_training.py tests/test_ppo_distributed.py -x -q -m gpu > gpurun_out/c18/tests.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/c18/tests.log
timeout -k 10 300 python bench.py --workload train --ppo-minibatches 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/c18/train.json 2> gpurun_out/c18/train.err; cut -c1-260 gpurun_out/c18/train.json; tail -3 gpurun_out/c18/train.err
