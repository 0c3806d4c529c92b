# packed env-step arithmetic: parity (full batch + scale + OCC2 equality) in one process, then the whole GPU suite and the bench
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity_full_batch.py tests/test_gpu_parity_scale.py -m gpu -x -q -s > gpurun_out/c53_parity.log 2>&1
echo parity rc=$?
cp gpurun_out/drift.json gpurun_out/c53_drift.json
timeout -k 10 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity_full_batch.py --deselect tests/test_gpu_parity_scale.py > gpurun_out/c53_gpu_tests.log 2>&1
echo suite rc=$?
timeout -k 10 300 python bench.py > gpurun_out/c53_bench.json 2> gpurun_out/c53_bench.err
echo bench rc=$?
grep "drift@" gpurun_out/c53_parity.log | cut -c1-220; tail -3 gpurun_out/c53_parity.log; tail -3 gpurun_out/c53_gpu_tests.log; cut -c1-300 gpurun_out/c53_bench.json
