# cfg-4 training with the callbacks on (evaluations.npz, checkpoints, progress.jsonl + TensorBoard event file), then the
# learning curve read back from the event file the way visualize/learning_curves.py:35-84 tabulates it
cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg4_easy_16384.yaml
rm -rf gpurun_out/tb gpurun_out/best gpurun_out/cfg4_ckpt
( time timeout -k 10 500 python train_rate.py --config $CFG --bf16 --callbacks ) > gpurun_out/train_cfg4_tb.log 2>&1 && \
timeout -k 10 100 python - > gpurun_out/train_cfg4_tb_scalars.md <<'PY'
import sys
sys.path.insert(0, ".")
from hcrl_amd import tfevents
tb = tfevents.load_scalars("gpurun_out/tb")
print("| tag | points | first (step, value) | last (step, value) |\n|---|---|---|---|")
for tag in sorted(tb):
    r = tb[tag]
    print(f"| `{tag}` | {len(r)} | {r[0][0]}, {r[0][1]:.4g} | {r[-1][0]}, {r[-1][1]:.4g} |")
print()
print("| step | rollout/ep_rew_mean | rollout/ep_len_mean | train/approx_kl | train/std |\n|---|---|---|---|---|")
rows = {t: dict((s, v) for s, v, _ in tb[t]) for t in ("rollout/ep_rew_mean", "rollout/ep_len_mean", "train/approx_kl", "train/std") if t in tb}
steps = sorted(rows["train/approx_kl"])
for s in steps[:: max(1, len(steps) // 16)] + [steps[-1]]:
    print(f"| {s} | " + " | ".join(f"{rows[t].get(s, float('nan')):.4g}" for t in ("rollout/ep_rew_mean", "rollout/ep_len_mean", "train/approx_kl", "train/std")) + " |")
PY
echo rc=$?
grep -E "real|final" gpurun_out/train_cfg4_tb.log | cut -c1-200; ls -la gpurun_out/tb; head -30 gpurun_out/train_cfg4_tb_scalars.md
