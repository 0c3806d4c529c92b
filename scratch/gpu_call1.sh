set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
./scratch/ubench/issue > gpurun_out/ubench_issue.txt 2>&1 &&
timeout -k 10 900 python -m pytest tests/test_gpu_parity_scale.py -x -q -m gpu -s > gpurun_out/r02_scale_tests.log 2>&1; echo "scale tests rc=$?" >> gpurun_out/r02_scale_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_bench_multi.py -x -q -m gpu -s > gpurun_out/r02_bench_multi.log 2>&1; echo "bench multi rc=$?" >> gpurun_out/r02_bench_multi.log
bash scratch/prof_r02_kernels.sh cascade r02a_cascade > gpurun_out/prof_cascade.log 2>&1
bash scratch/prof_r02_kernels.sh physics r02a_physics > gpurun_out/prof_physics.log 2>&1
tail -5 gpurun_out/r02_scale_tests.log gpurun_out/r02_bench_multi.log
cat gpurun_out/ubench_issue.txt
