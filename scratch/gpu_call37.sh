cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c37
FDYN_LIB=$GRAFT_REPO_ROOT/scratch/libfdyn_stamps.so timeout -k 10 200 python scratch/phase_stamps.py 65536 1200 > gpurun_out/c37/stamps.log 2>&1
grep "wave totals" gpurun_out/c37/stamps.log | head -3; grep -A7 "^launch" gpurun_out/c37/stamps.log | tail -8
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_scale.py tests/test_gpu_dropin.py tests/test_gpu_agents.py tests/test_gpu_sensor.py tests/test_gpu_learned_agent.py -x -q -m gpu > gpurun_out/c37/tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/c37/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/c37/bench.json 2> gpurun_out/c37/bench.err; cut -c100-330 gpurun_out/c37/bench.json
timeout -k 10 300 python bench.py --workload rollout --no-cpu-baseline --no-extras --steps 200 --warmup 20 | cut -c100-220
