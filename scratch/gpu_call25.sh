cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c28
FDYN_LIB=$GRAFT_REPO_ROOT/scratch/libfdyn_stamps.so timeout -k 10 200 python scratch/phase_stamps.py 65536 1200 > gpurun_out/c28/stamps.log 2>&1
grep "wave totals\|p100\|p90 \|p50 \|mean counts" gpurun_out/c28/stamps.log | head -10
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_dropin.py tests/test_gpu_agents.py -x -q -m gpu > gpurun_out/c28/tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/c28/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c28/bench.json 2> gpurun_out/c28/bench.err; cut -c1-330 gpurun_out/c28/bench.json
