cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg4_easy_16384.yaml
( time timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 train_rate.py --config $CFG --bf16 --backend gloo --device-index 0 --set training.n_envs=8192 --set training.total_timesteps=60000000 --set paths.model_save_dir=gpurun_out/ckpt_n2 ) > gpurun_out/train_n2.log 2>&1
echo rc=$?
grep -E "iter (10|30|50) |final|real|Error|error" gpurun_out/train_n2.log | cut -c1-180
