cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg4_easy_16384.yaml
for b in approx; do
  echo "=== bootstrap_timeouts=$b"
  ( time timeout -k 10 400 python train_rate.py --config $CFG --bf16 --set ppo.bootstrap_timeouts=$b --set paths.model_save_dir=gpurun_out/ckpt_boot ) 2>&1 | grep -E "iter (50|100|150|190) |final|real" | cut -c1-170
  for d in easy medium hard; do timeout -k 10 200 python eval_rate.py --model gpurun_out/ckpt_boot/final_model.pt --n-episodes 4096 --difficulty $d 2>&1 | grep -E "RMSE|Reward|Success"; done
done
