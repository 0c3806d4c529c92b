cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg4_easy_16384.yaml
for mb in 2 4; do
  echo "=== n_minibatches=$mb"
  ( time timeout -k 10 400 python train_rate.py --config $CFG --bf16 --set ppo.n_minibatches=$mb --set ppo.ent_coef=0.001 --set paths.model_save_dir=gpurun_out/ckpt_mb$mb ) 2>&1 | grep -E "iter (50|100|150|190) |final|real" | cut -c1-170
  timeout -k 10 200 python eval_rate.py --model gpurun_out/ckpt_mb$mb/final_model.pt --n-episodes 4096 --difficulty medium 2>&1 | grep -E "RMSE|Reward|Success"
done
