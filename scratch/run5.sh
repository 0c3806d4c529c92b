cd $GRAFT_REPO_ROOT
echo "== default build"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log; grep drift gpurun_out/gpu_tests.log | grep -v f64
for p in mixed f32; do timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json; python scratch/show.py gpurun_out/b.json; done
echo "== hardware v_sin/v_cos build"
export FDYN_LIB=$GRAFT_REPO_ROOT/scratch/libfdyn_hwtrig.so
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s > gpurun_out/gpu_tests_hw.log 2>&1; tail -2 gpurun_out/gpu_tests_hw.log; grep drift gpurun_out/gpu_tests_hw.log | grep -v f64
for p in mixed f32; do timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json; python scratch/show.py gpurun_out/b.json; done
