# The trainer plumbing end to end on the curriculum config: BC + 3 phases with callbacks (evaluations every 2000 vec-steps,
# checkpoints every 4000), recover_training.py picks the checkpoint nearest the best evaluation, eval_rate.py scores it
# from the archive written in SB3's layout, and the scalars are read back from the TensorBoard event file.
cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg5_curriculum_16384.yaml
rm -rf gpurun_out/tb gpurun_out/best gpurun_out/cfg5_ckpt
( time timeout -k 10 900 python train_rate.py --config $CFG --bf16 --bc-pretrain 3 --callbacks \
    --set training.eval_freq=2000 --set training.save_freq=4000 --set evaluation.n_eval_episodes=256 ) > gpurun_out/train_cfg5_cb.log 2>&1 && \
timeout -k 10 100 python recover_training.py --checkpoint-dir gpurun_out/cfg5_ckpt --eval-dir gpurun_out/best --output gpurun_out/recovered.pt > gpurun_out/recover_cfg5.log 2>&1 && \
timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_ckpt/final_model.zip --n-episodes 4096 --difficulty hard --compare-pid > gpurun_out/eval_cfg5_zip.log 2>&1 && \
timeout -k 10 100 python - > gpurun_out/cfg5_eval_curve.md <<'PY'
import sys
import numpy as np
sys.path.insert(0, ".")
from hcrl_amd import tfevents
ev = np.load("gpurun_out/best/evaluations.npz")
tb = tfevents.load_scalars("gpurun_out/tb")
er = dict((s, v) for s, v, _ in tb.get("rollout/ep_rew_mean", []))
steps = sorted(er)
print("| timesteps | eval mean reward (256 episodes, deterministic) | eval mean length | nearest rollout/ep_rew_mean |\n|---|---|---|---|")
for t, r, l in zip(ev["timesteps"], ev["results"].mean(1), ev["ep_lengths"].mean(1)):
    near = min(steps, key=lambda s: abs(s - t)) if steps else None
    print(f"| {int(t):,} | {r:+.1f} | {l:.0f} | {er[near]:+.1f} |" if near is not None else f"| {int(t):,} | {r:+.1f} | {l:.0f} | - |")
PY
echo rc=$?
ls gpurun_out/cfg5_ckpt > gpurun_out/cfg5_ckpt_listing.txt; rm -f gpurun_out/cfg5_ckpt/rate_controller_*_steps.pt gpurun_out/recovered.pt   # keep the merge under 64 MiB
grep -E "real|final" gpurun_out/train_cfg5_cb.log | cut -c1-160; grep -E "Best evaluation|Copying|loaded|Policy" gpurun_out/recover_cfg5.log; grep -E "RMSE:|Total Reward" gpurun_out/eval_cfg5_zip.log | head -4; cat gpurun_out/cfg5_eval_curve.md; cat gpurun_out/cfg5_ckpt_listing.txt
