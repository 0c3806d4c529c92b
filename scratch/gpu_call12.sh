R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02_gpu_all.log 2>&1; echo "rc=$?" >> gpurun_out/r02_gpu_all.log
tail -n 5 gpurun_out/r02_gpu_all.log
grep -E "drift@scale\] env mixed" -A3 gpurun_out/r02_gpu_all.log | head -n 6
bash scratch/gpu_call11.sh
