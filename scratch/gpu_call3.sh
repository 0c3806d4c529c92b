R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_agents.py tests/test_gpu_dropin.py -x -q -m gpu -s > gpurun_out/r02_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r02_parity.log
timeout -k 10 600 python scratch/drift_explore.py > gpurun_out/drift_explore2.log 2>&1; echo "rc=$?" >> gpurun_out/drift_explore2.log
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras > gpurun_out/bench_env_new.json 2> gpurun_out/bench_env_new.err
bash scratch/prof_r02_kernels.sh env r02b_env > gpurun_out/prof_env.log 2>&1
grep -E "drift|passed|failed|rc=" gpurun_out/r02_parity.log | tail -n 30
tail -n 8 gpurun_out/drift_explore2.log
cut -c1-600 gpurun_out/bench_env_new.json
