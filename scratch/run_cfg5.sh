cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg5_curriculum_16384.yaml
( time timeout -k 10 900 python train_rate.py --config $CFG --bf16 --bc-pretrain 3 ) > gpurun_out/train_cfg5.log 2>&1 && \
for d in easy medium hard; do timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_ckpt/final_model.pt --n-episodes 4096 --difficulty $d --compare-pid; done > gpurun_out/eval_cfg5.log 2>&1 && \
timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_ckpt/final_model.pt --n-episodes 4096 --difficulty hard --command-type random >> gpurun_out/eval_cfg5.log 2>&1
echo rc=$?
grep -E "phase|BC|iter (10|140|190|50|90) |final|real" gpurun_out/train_cfg5.log | cut -c1-200
grep -E "^#|Evaluating|RMSE|Reward|Success|Settling" gpurun_out/eval_cfg5.log | head -60
