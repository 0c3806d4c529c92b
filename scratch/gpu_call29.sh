cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c29
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/c29/tests.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/c29/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c29/bench.json 2> gpurun_out/c29/bench.err; cut -c100-330 gpurun_out/c29/bench.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --batch 1048576 --steps 200 --warmup 20 > gpurun_out/c29/bench_1m.json 2> gpurun_out/c29/bench_1m.err; cut -c100-330 gpurun_out/c29/bench_1m.json
