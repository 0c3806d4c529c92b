R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python scratch/drift_explore.py > gpurun_out/drift_explore.log 2>&1; echo "rc=$?" >> gpurun_out/drift_explore.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity_scale.py -q -m gpu -s -k "env or capped" > gpurun_out/r02_scale_tests2.log 2>&1; echo "rc=$?" >> gpurun_out/r02_scale_tests2.log
tail -n 12 gpurun_out/drift_explore.log
tail -n 30 gpurun_out/r02_scale_tests2.log
