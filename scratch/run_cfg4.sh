cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg4_easy_16384.yaml
mkdir -p gpurun_out
( time timeout -k 10 400 python train_rate.py --config $CFG --bf16 ) > gpurun_out/train_cfg4_A.log 2>&1 && \
for d in easy medium hard; do timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg4_ckpt/final_model.pt --n-episodes 4096 --difficulty $d --compare-pid; done > gpurun_out/eval_cfg4_A.log 2>&1
echo rc=$?
grep -E "iter (10|100|190) |final|real" gpurun_out/train_cfg4_A.log | cut -c1-200
grep -E "RMSE|Total Reward|Success" gpurun_out/eval_cfg4_A.log
