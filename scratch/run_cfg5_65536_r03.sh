# round 3: the BASELINE-config-5 shape on ONE rank (65 536 envs, curriculum, BC pre-training) with this round's policy kernels
# (features extractor with pre-scaled gate rows, both cells in one launch, trunks + heads on eight waves), then eval_rate.py.
cd $GRAFT_REPO_ROOT
CFG=hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/configs/training/cfg5_curriculum_65536.yaml
( time timeout -k 10 1000 python train_rate.py --config $CFG --bf16 --bc-pretrain 3 --callbacks ) > gpurun_out/r03_train_cfg5_65536.log 2>&1 && \
for d in easy medium hard; do timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_65536_ckpt/final_model.pt --n-episodes 4096 --difficulty $d --compare-pid; done > gpurun_out/r03_eval_cfg5_65536.log 2>&1 && \
timeout -k 10 200 python eval_rate.py --model gpurun_out/cfg5_65536_ckpt/final_model.pt --n-episodes 4096 --difficulty hard --command-type random >> gpurun_out/r03_eval_cfg5_65536.log 2>&1
echo rc=$?
python - <<'PY' > gpurun_out/r03_evaluations_cfg5_65536.txt 2>&1
import numpy as np
d = np.load("gpurun_out/cfg5_65536_best/evaluations.npz")
for t, r, l in zip(d["timesteps"], d["results"].mean(1), d["ep_lengths"].mean(1)):
    print(f"eval @ {t:>12d} steps: mean reward {r:9.2f} mean length {l:6.1f}")
PY
# the checkpoints would push gpurun_out/ past the merge limit: keep the logs only
rm -rf gpurun_out/cfg5_65536_ckpt gpurun_out/cfg5_65536_best gpurun_out/cfg5_65536_tb 2>/dev/null
grep -E "phase|BC|iter (10|40|80|120|160|200|240|280|320|360|400|440|470) |final|real" gpurun_out/r03_train_cfg5_65536.log | cut -c1-200 | tail -30
grep -E "^#|Evaluating|RMSE|Reward|Success|Settling" gpurun_out/r03_eval_cfg5_65536.log | head -50
cat gpurun_out/r03_evaluations_cfg5_65536.txt
du -sh gpurun_out
