# round 2, re-entry: issue-rate microbenchmark with 4-byte encodings; rocprofv3 kernel stats of the train workload at HEAD
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 120 scratch/ubench/issue > gpurun_out/c51_issue.log 2>&1
echo issue rc=$?
timeout -k 10 500 bash scratch/prof_train.sh > gpurun_out/c51_prof_train.log 2>&1
echo prof rc=$?
cat gpurun_out/c51_issue.log; head -40 gpurun_out/c51_prof_train.log
