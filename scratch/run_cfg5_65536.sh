cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg5_curriculum_65536.yaml
( time timeout -k 10 1000 python train_rate.py --config $CFG --bf16 --bc-pretrain 3 --callbacks ) > gpurun_out/train_cfg5_65536.log 2>&1 && \
for d in easy medium hard; do timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_65536_ckpt/final_model.pt --n-episodes 4096 --difficulty $d --compare-pid; done > gpurun_out/eval_cfg5_65536.log 2>&1 && \
timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_65536_ckpt/final_model.pt --n-episodes 4096 --difficulty hard --command-type random >> gpurun_out/eval_cfg5_65536.log 2>&1
echo rc=$?
grep -E "phase|BC|iter (10|40|80|120|160|200|240|280|320|360|400|440|470) |final|real" gpurun_out/train_cfg5_65536.log | cut -c1-220
grep -E "^#|Evaluating|RMSE|Reward|Success|Settling" gpurun_out/eval_cfg5_65536.log | head -60
python - <<'PY'
import numpy as np
d = np.load("gpurun_out/cfg5_65536_best/evaluations.npz")
for t, r, l in zip(d["timesteps"], d["results"].mean(1), d["ep_lengths"].mean(1)):
    print(f"eval @ {t:>12d} steps: mean reward {r:9.2f} mean length {l:6.1f}")
PY
