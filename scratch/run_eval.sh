cd $GRAFT_REPO_ROOT
true && \
for d in easy medium hard; do
  echo "##### difficulty $d (4096 episodes, step commands, mixed precision)"
  timeout -k 10 300 python eval_rate.py --model scratch/cfg4_final_model.ckpt --compare-pid --n-episodes 4096 --difficulty $d || exit 1
done > gpurun_out/eval_rate_cfg4.log 2>&1
echo "##### PID told the true dt (0.02), medium" >> gpurun_out/eval_rate_cfg4.log
timeout -k 10 300 python eval_rate.py --pid-only --pid-true-dt --n-episodes 4096 --difficulty medium >> gpurun_out/eval_rate_cfg4.log 2>&1
tail -5 gpurun_out/eval_rate_cfg4.log
