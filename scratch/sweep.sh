cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg4_easy_16384.yaml
run() { echo "=== $*"; timeout -k 10 300 python train_rate.py --config $CFG --bf16 --set ppo.n_minibatches=8 --set training.total_timesteps=40000000 --set training.log_interval=6 "$@" 2>&1 | grep -E "ppo|final" | cut -c1-185; }
run --set ppo.learning_rate=1e-3
run --set ppo.learning_rate=3e-4 --set ppo.reward_scale=0.02
run --set ppo.learning_rate=1e-3 --set ppo.reward_scale=0.02
