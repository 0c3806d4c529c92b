cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
for lpw in 64 32 16; do
  for p in mixed f32; do
    FDYN_LPW=$lpw timeout -k 10 200 python bench.py --precision $p --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json
    echo -n "lpw=$lpw "; python scratch/show.py gpurun_out/b.json
  done
done
for w in physics cascade; do for lpw in 64 32 16; do
    FDYN_LPW=$lpw timeout -k 10 200 python bench.py --workload $w --precision mixed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json
    echo -n "lpw=$lpw "; python scratch/show.py gpurun_out/b.json
done; done
FDYN_LPW=32 timeout -k 10 200 python bench.py --precision f64 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json; echo -n "lpw=32 "; python scratch/show.py gpurun_out/b.json
FDYN_LPW=16 timeout -k 10 200 python bench.py --precision f64 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/b.json; echo -n "lpw=16 "; python scratch/show.py gpurun_out/b.json
