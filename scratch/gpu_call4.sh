R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_all.log
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras > gpurun_out/bench_env_new.json 2> gpurun_out/bench_env_new.err
bash scratch/prof_r02_kernels.sh env r02b_env > gpurun_out/prof_env.log 2>&1
tail -n 15 gpurun_out/r02_gpu_all.log
cut -c1-300 gpurun_out/bench_env_new.json
