R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_all.log
bash scratch/prof_r02_kernels.sh cascade r02b_cascade > gpurun_out/prof_cascade.log 2>&1
bash scratch/prof_r02_kernels.sh physics r02b_physics > gpurun_out/prof_physics.log 2>&1
bash scratch/prof_r02_kernels.sh env r02b_env > gpurun_out/prof_env.log 2>&1
tail -n 6 gpurun_out/r02_gpu_all.log
for f in cascade physics env; do cut -c1-250 gpurun_out/prof_r02b_$f/trace_bench.json; echo; done
