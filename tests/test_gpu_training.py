"""GPU: the policy / PPO / BC glue on the real device-resident env."""
import os

import numpy as np
import pytest
import torch

from hcrl_amd.policy import RateLSTMPolicy
from hcrl_amd.ppo import PPOConfig, RecurrentPPO
from hcrl_amd.rate_env import GpuRateVecEnv
from hcrl_amd import training_utils as tu

pytestmark = pytest.mark.gpu


def test_pid_demonstrations_have_distinct_rows_and_bc_fits_them():
    obs, acts = tu.collect_pid_demonstrations(n_episodes=256, difficulty="medium", seed=42)
    assert obs.shape[1] == 18 and acts.shape[1] == 4 and obs.shape[0] == acts.shape[0] > 256 * 20
    assert np.unique(obs[:2000], axis=0).shape[0] > 1900          # the reference's pickle has ONE unique row
    assert np.all(np.abs(acts[:, :3]) <= 1.0) and np.allclose(acts[:, 3], 0.6)
    env = GpuRateVecEnv(64, "easy", seed=0)
    m = RecurrentPPO(env, RateLSTMPolicy(), PPOConfig(n_steps=4))
    losses = tu.behavior_cloning_pretrain(m, obs[:20000], acts[:20000], epochs=3, batch_size=1024)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_ppo_iterations_run_and_stay_finite(dtype):
    env = GpuRateVecEnv(2048, "easy", 10.0, 0.02, "step", seed=5, precision="mixed", sampling="device")
    m = RecurrentPPO(env, RateLSTMPolicy(compute_dtype=dtype), PPOConfig(n_steps=16, n_epochs=2, n_minibatches=4), seed=1)
    m.learn(2048 * 16 * 3, log_interval=0)
    assert m.num_timesteps == 2048 * 16 * 3
    assert all(np.isfinite(v) for v in m.last_stats.values())
    assert all(bool(torch.isfinite(p).all()) for p in m.policy.parameters())
    ev = tu.run_final_evaluation(m, difficulty="easy", n_episodes=32)
    assert 0 < ev["mean_length"] <= 500


@pytest.mark.parametrize("mode", [True, "approx"])
def test_timeout_bootstrap_path(mode):
    """True: V(terminal observation) as SB3 does (host sync per step); "approx": gamma * V(s_t), device-side, inside the
    rollout graph.  10-step episodes: every rollout sees truncations, and their rewards must have gained the bootstrap."""
    env = GpuRateVecEnv(256, "easy", 0.2, 0.02, "step", seed=2)
    m = RecurrentPPO(env, RateLSTMPolicy(use_lstm=False), PPOConfig(n_steps=12, n_epochs=1, n_minibatches=2,
                                                                     bootstrap_timeouts=mode))
    m.learn(256 * 12, log_interval=0)
    assert np.isfinite(m.last_stats["value_loss"])
    plain = RecurrentPPO(GpuRateVecEnv(256, "easy", 0.2, 0.02, "step", seed=2), RateLSTMPolicy(use_lstm=False),
                         PPOConfig(n_steps=12, n_epochs=1, n_minibatches=2), seed=0, use_graph=False)
    plain.policy.load_state_dict(m.policy.state_dict())
    boot = RecurrentPPO(GpuRateVecEnv(256, "easy", 0.2, 0.02, "step", seed=2), RateLSTMPolicy(use_lstm=False),
                        PPOConfig(n_steps=12, n_epochs=1, n_minibatches=2, bootstrap_timeouts=mode), seed=0, use_graph=False)
    boot.policy.load_state_dict(m.policy.state_dict())
    for r in (plain, boot):
        r.policy._noise_seed = 7
        torch.manual_seed(5)
        r.collect_rollout()
    # the bootstrap term lives in its own buffer: logged rewards stay the env's own, GAE sees rew + boot
    assert torch.equal(boot.buf_act, plain.buf_act) and torch.equal(boot.buf_rew, plain.buf_rew)
    diff = boot.buf_boot
    assert plain.buf_boot is None and float(diff.abs().max()) > 0.0
    timeouts = (boot.buf_start[1:] > 0)                               # an episode start at t+1 = an episode end at t
    assert float(diff[:-1][~timeouts].abs().max()) == 0.0             # only steps that ended an episode were touched
    assert not torch.equal(boot.adv, plain.adv)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("zero_state", [False, True])
def test_fused_lstm_cell_matches_plain_torch_fp32(dtype, tol, zero_state):
    """HIP fused cell (fwd + bwd) vs a plain PyTorch fp32 reference of the same op."""
    from hcrl_amd.fused import lstm_cell, _lstm_cell_torch
    torch.manual_seed(0)
    B, H = 777, 256                                                   # ragged batch
    g32 = (torch.randn(B, 4 * H, device="cuda") * 2).requires_grad_()
    c32 = None if zero_state else torch.randn(B, H, device="cuda").requires_grad_()
    h_ref, c_ref = _lstm_cell_torch(g32, c32)
    w_h, w_c = torch.randn_like(h_ref), torch.randn_like(c_ref)
    (h_ref * w_h).sum().add((c_ref * w_c).sum()).backward()
    g = g32.detach().to(dtype).requires_grad_()
    c = None if zero_state else c32.detach().clone().requires_grad_()
    if dtype == torch.bfloat16:                                       # reference on the SAME rounded inputs
        g32b = g.detach().float().requires_grad_()
        c32b = None if zero_state else c32.detach().clone().requires_grad_()
        h_ref, c_ref = _lstm_cell_torch(g32b, c32b)
        (h_ref * w_h).sum().add((c_ref * w_c).sum()).backward()
        g32, c32 = g32b, c32b
    h, cn = lstm_cell(g, c)
    assert h.dtype == dtype and cn.dtype == torch.float32
    (h.float() * w_h).sum().add((cn * w_c).sum()).backward()
    assert (h.float() - h_ref).abs().max() < tol and (cn - c_ref).abs().max() < tol
    assert (g.grad.float() - g32.grad).abs().max() < tol * 4
    if not zero_state:
        assert (c.grad - c32.grad).abs().max() < tol * 4
    else:
        # the h-only form of a zero-state cell (what the feature extractor's layers use): c is never stored, the backward
        # rebuilds it from the saved gates -- same h, same gate gradient as the plain reference differentiated through h alone
        g2 = g.detach().clone().requires_grad_()
        h2, none_c = lstm_cell(g2, None, need_c=False)
        assert none_c is None and torch.equal(h2, h)
        (h2.float() * w_h).sum().backward()
        gr = g32.detach().clone().requires_grad_()
        (_lstm_cell_torch(gr, None)[0] * w_h).sum().backward()
        assert (g2.grad.float() - gr.grad).abs().max() < tol * 4


def test_fused_gae_matches_torch_loop():
    from hcrl_amd.ppo import compute_gae
    torch.manual_seed(3)
    T, N = 37, 1000
    rew, val = torch.randn(T, N), torch.randn(T, N)
    st = (torch.rand(T, N) < 0.1).float()
    lv, ld = torch.randn(N), (torch.rand(N) < 0.3).float()
    a_ref, r_ref = compute_gae(rew, val, st, lv, ld, 0.99, 0.95)      # CPU tensors -> plain torch loop
    a, r = compute_gae(rew.cuda(), val.cuda(), st.cuda(), lv.cuda(), ld.cuda(), 0.99, 0.95)
    assert (a.cpu() - a_ref).abs().max() < 1e-4 and (r.cpu() - r_ref).abs().max() < 1e-4


def test_policy_gpu_matches_cpu_forward():
    torch.manual_seed(0)
    p = RateLSTMPolicy()
    obs = torch.randn(64, 18)
    st = p.initial_state(64)
    torch.manual_seed(1); a_c, v_c, lp_c, s_c = p.step(obs, st, torch.zeros(64), deterministic=True)
    pg = RateLSTMPolicy(); pg.load_state_dict(p.state_dict()); pg.cuda()
    a_g, v_g, lp_g, s_g = pg.step(obs.cuda(), pg.initial_state(64, "cuda"), torch.zeros(64, device="cuda"), deterministic=True)
    assert (a_g.cpu() - a_c).abs().max() < 1e-4 and (v_g.cpu() - v_c).abs().max() < 1e-4
    assert (s_g.pi_c.cpu() - s_c.pi_c).abs().max() < 1e-4


def _mfma_cell(x, h, c, keep, W, bias, H, with_h32=True):
    from hcrl_amd import _lib
    lib = _lib.load()
    B = x.shape[0]
    kh = 0 if h is None else h.shape[1]
    h_out = torch.empty((B, H), dtype=torch.bfloat16, device="cuda")
    c_out = torch.empty((B, H), dtype=torch.float32, device="cuda")
    h32 = torch.empty((B, H), dtype=torch.float32, device="cuda") if with_h32 else None
    _lib.check(lib.fdyn_lstm_cell_mfma(x.data_ptr(), x.shape[1], _lib.ptr(h), kh, _lib.ptr(c), _lib.ptr(keep), W.data_ptr(),
                                       bias.data_ptr(), h_out.data_ptr(), c_out.data_ptr(), _lib.ptr(h32), B, H,
                                       _lib.current_stream()), "lstm_cell_mfma")
    return h_out, c_out, (h32 if with_h32 else h_out.float())


@pytest.mark.parametrize("with_h32", [True, False])
@pytest.mark.parametrize("B", [128, 777, 4096, 65536])
@pytest.mark.parametrize("kx,kh", [(128, 256), (128, 0), (256, 0), (128, 128)])
def test_lstm_mfma_cell_matches_plain_torch_fp32(B, kx, kh, with_h32):
    """Hand-written MFMA LSTM cell vs plain PyTorch fp32 on the same bf16-rounded inputs (asymmetric random data, so a
    transposed or permuted fragment map cannot pass).  `with_h32=False` on full 32-row tiles is the rollout's configuration:
    the software-pipelined path (epilogue folded under the next unit's MFMAs); an fp32 copy of h' or a ragged tile takes the
    generic path; 777 rows mix both in one launch; 65 536 rows is the benchmark size (every workgroup full, no split)."""
    torch.manual_seed(B + kx + kh)
    H = 256
    x = (torch.randn(B, kx, device="cuda") * 0.7).bfloat16()
    W = (torch.randn(4 * H, kx + kh, device="cuda") * 0.08).bfloat16()
    bias = torch.randn(4 * H, device="cuda") * 0.3
    if kh:
        h = (torch.randn(B, kh, device="cuda") * 0.5).bfloat16()
        c = torch.randn(B, H, device="cuda")
        keep = (torch.rand(B, device="cuda") > 0.25).float()
        xin = torch.cat([x.float(), h.float() * keep[:, None]], 1)
        c_eff = c * keep[:, None]
    else:
        h = c = keep = None
        xin, c_eff = x.float(), None
    gates = xin @ W.float().t() + bias
    i, f, g, o = gates.chunk(4, 1)
    c_ref = torch.sigmoid(i) * torch.tanh(g) + (torch.sigmoid(f) * c_eff if kh else 0)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    h_out, c_out, h32 = _mfma_cell(x, h, c, keep, W, bias, H, with_h32)
    assert (c_out - c_ref).abs().max() < 2e-3, float((c_out - c_ref).abs().max())
    assert (h32 - h_ref).abs().max() < (2e-3 if with_h32 else 1e-2)
    assert (h_out.float() - h_ref).abs().max() < 1e-2


@pytest.mark.parametrize("B", [777, 4096, 32768])
@pytest.mark.parametrize("kx,kh", [(128, 256), (128, 0), (256, 0)])
def test_lstm_mfma_train_cell_matches_plain_torch_fp32(B, kx, kh):
    """The BPTT-forward instantiation of the MFMA cell (fdyn_lstm_cell_mfma_train) vs plain PyTorch fp32 on the same
    bf16-rounded inputs: h', c', the activated gates it leaves for the backward pass (four-gate layout for the recurrent
    cell, three-gate (i, g, o) for a zero-state layer) and h' * keep_next packed into rows of a wider stride."""
    from hcrl_amd import _lib
    torch.manual_seed(B + kx + kh + 1)
    H, lib = 256, _lib.load()
    x = (torch.randn(B, kx, device="cuda") * 0.7).bfloat16()
    W = (torch.randn(4 * H, kx + kh, device="cuda") * 0.08).bfloat16()
    bias = torch.randn(4 * H, device="cuda") * 0.3
    if kh:
        h = (torch.randn(B, kh, device="cuda") * 0.5).bfloat16()
        c = torch.randn(B, H, device="cuda")
        keep = (torch.rand(B, device="cuda") > 0.25).float()
        xin = torch.cat([x.float(), h.float() * keep[:, None]], 1)
        c_eff = c * keep[:, None]
    else:
        h = c = keep = None
        xin, c_eff = x.float(), None
    keep_next = (torch.rand(B, device="cuda") > 0.25).float()
    gates = xin @ W.float().t() + bias
    i, f, g, o = gates.chunk(4, 1)
    si, sf, tg, so = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c_ref = si * tg + (sf * c_eff if kh else 0)
    h_ref = so * torch.tanh(c_ref)
    act_ref = torch.cat([si, sf, tg, so] if kh else [si, tg, so], 1)
    ng, K2 = (4 if kh else 3), kx + H + 8
    h_out = torch.empty(B, H, dtype=torch.bfloat16, device="cuda")
    c_out = torch.empty(B, H, device="cuda") if kh else None
    act = torch.full((B, ng * H), 7.0, dtype=torch.bfloat16, device="cuda")
    nxt = torch.full((B, K2), 9.0, dtype=torch.bfloat16, device="cuda")          # h' * keep_next goes to columns kx .. kx + H
    _lib.check(lib.fdyn_lstm_cell_mfma_train(x.data_ptr(), kx, _lib.ptr(h), kh, _lib.ptr(c), _lib.ptr(keep), W.data_ptr(),
                                             bias.data_ptr(), h_out.data_ptr(), _lib.ptr(c_out), act.data_ptr(),
                                             nxt.data_ptr() + kx * 2, K2, keep_next.data_ptr(), B, H, _lib.current_stream()),
               "lstm_cell_mfma_train")
    torch.cuda.synchronize()
    if kh:
        assert (c_out - c_ref).abs().max() < 2e-3
    assert (h_out.float() - h_ref).abs().max() < 1e-2
    assert (act.float() - act_ref).abs().max() < 1e-2, float((act.float() - act_ref).abs().max())
    assert (nxt[:, kx:kx + H].float() - h_out.float() * keep_next[:, None]).abs().max() == 0.0
    assert bool((nxt[:, :kx] == 9.0).all()) and bool((nxt[:, kx + H:] == 9.0).all())      # nothing outside its columns


@pytest.mark.parametrize("B", [77, 4096, 65536])
def test_policy_trunks_kernel_matches_plain_torch_fp32(B):
    """fdyn_policy_trunks (both trunks, two layers each, one MFMA launch; the first layer's output tile is the second layer's
    operand) against plain fp32 PyTorch on the same bf16-rounded inputs, with the intermediate rounded to bf16 as the kernel
    keeps it.  Asymmetric random data: a wrong fragment map or k permutation cannot pass."""
    from hcrl_amd import _lib
    from hcrl_amd.policy import _KPERM16
    torch.manual_seed(B)
    lib, bf = _lib.load(), torch.bfloat16
    h = [(torch.randn(B, 256, device="cuda") * 0.6).to(bf) for _ in range(2)]
    W1 = (torch.randn(2, 128, 256, device="cuda") * 0.08).to(bf)
    b1 = torch.randn(2, 128, device="cuda") * 0.2
    W2 = (torch.randn(2, 64, 128, device="cuda") * 0.1).to(bf)
    b2 = torch.randn(2, 64, device="cuda") * 0.2
    perm = torch.tensor([16 * (k // 16) + _KPERM16[k % 16] for k in range(128)], device="cuda")
    W2p = W2[:, :, perm].contiguous()
    lat = [torch.full((B, 64), 5.0, dtype=bf, device="cuda") for _ in range(2)]
    _lib.check(lib.fdyn_policy_trunks(h[0].data_ptr(), h[1].data_ptr(), W1.data_ptr(), b1.data_ptr(), W2p.data_ptr(), b2.data_ptr(),
                                      lat[0].data_ptr(), lat[1].data_ptr(), B, _lib.current_stream()), "policy_trunks")
    torch.cuda.synchronize()
    for g in range(2):
        mid = torch.relu(h[g].float() @ W1[g].float().t() + b1[g]).to(bf).float()
        ref = torch.relu(mid @ W2[g].float().t() + b2[g])
        err = float((lat[g].float() - ref).abs().max())
        assert err < 2e-2 * max(1.0, float(ref.abs().max())), (g, err)


def test_episode_flags_kernel_matches_tensor_ops():
    """fused.episode_flags (one launch: episode_start, keep = 1 - episode_start, noise counter += 1) against the tensor ops it
    replaces, ragged size, repeated calls."""
    from hcrl_amd.fused import episode_flags
    torch.manual_seed(5)
    n = 70001
    start = torch.full((n,), 7.0, device="cuda"); keep = torch.full((n,), 7.0, device="cuda")
    counter = torch.zeros(1, dtype=torch.int32, device="cuda")
    for k in range(3):
        term = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8)
        trunc = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8)
        episode_flags(term, trunc, start, keep, counter)
        ref = (term | trunc).float()
        assert torch.equal(start, ref) and torch.equal(keep, 1.0 - ref) and int(counter.item()) == k + 1
    episode_flags(term, trunc, start, None, None)                      # optional outputs
    assert torch.equal(start, ref) and int(counter.item()) == 3


def test_fused_rollout_step_matches_unfused_bf16_path():
    torch.manual_seed(0)
    p = RateLSTMPolicy(compute_dtype=torch.bfloat16).cuda()
    B = 300
    obs = torch.randn(B, 18, device="cuda")
    st = p.initial_state(B, "cuda")
    st = type(st)(*(t + 0.3 * torch.randn_like(t) for t in st))
    start = (torch.rand(B, device="cuda") < 0.3).float()
    with torch.no_grad():
        a0, v0, lp0, s0 = p.step(obs, st, start, deterministic=True)           # un-fused (no inference cache yet)
        p.prepare_inference()
        a1, v1, lp1, s1 = p.step(obs, st, start, deterministic=True)           # fused MFMA cells
    assert (a0 - a1).abs().max() < 3e-2 and (v0 - v1).abs().max() < 5e-2
    assert (s0.pi_c - s1.pi_c).abs().max() < 3e-2 and (s0.vf_h.float() - s1.vf_h.float()).abs().max() < 3e-2


def test_lstm_mfma_cell_full_chip_variant():
    """More than 2 workgroups per CU (no hidden-slice split), ragged tail included: full tiles take the software-pipelined
    path and the last, ragged workgroup the generic one inside the same launch."""
    test_lstm_mfma_cell_matches_plain_torch_fp32(256 * 256 + 77, 128, 256, False)
    test_lstm_mfma_cell_matches_plain_torch_fp32(256 * 256 + 77, 128, 256, True)
    test_lstm_mfma_cell_matches_plain_torch_fp32(256 * 256, 256, 0, False)


def test_train_rate_cli_end_to_end(tmp_path):
    """The reference-shaped CLI (config YAML + curriculum) runs end to end on a tiny budget and writes a checkpoint."""
    import yaml
    from hcrl_amd import train_rate
    cfg = yaml.safe_load(open(train_rate.DEFAULT_CONFIG))
    cfg["training"]["n_envs"] = 512
    cfg["ppo"]["n_steps"] = 8
    cfg["ppo"]["n_epochs"] = 1
    for ph in cfg["curriculum"]["phases"]:
        ph["timesteps"] = 512 * 8
    cfg["paths"]["model_save_dir"] = str(tmp_path / "ckpt")
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    train_rate.main(["--config", str(p), "--bf16", "--bc-pretrain", "1"])
    ck = torch.load(tmp_path / "ckpt" / "final_model.pt", weights_only=True)
    assert ck["num_timesteps"] == 3 * 512 * 8 and "policy" in ck
    # the same weights in the reference's archive layout (train_rate.py:353-355 -> final_model.zip), readable by the Level-4 agent
    from hcrl_amd import sb3_zip
    from hcrl_amd.flight_types import ControllerConfig
    from hcrl_amd.learned_rate_agent import LearnedRateAgent
    zpath = tmp_path / "ckpt" / "final_model.zip"
    assert sb3_zip.is_sb3_zip(zpath)
    sd, meta = sb3_zip.read_sb3_zip(zpath)
    assert meta["data"]["num_timesteps"] == 3 * 512 * 8 and meta["data"]["n_steps"] == 8
    for k, v in ck["policy"].items():
        assert torch.equal(sd[k], v.cpu().float()), k
    agent = LearnedRateAgent(str(zpath), ControllerConfig(), fallback_to_pid=False)
    assert agent.is_recurrent


def test_callbacks_checkpoints_best_model_and_resume(tmp_path):
    """create_callbacks (training_utils.py:72-156): periodic checkpoints, evaluations.npz + best model, then --resume
    continues the timestep count from a checkpoint."""
    import yaml
    from hcrl_amd import train_rate
    from hcrl_amd.training_utils import find_best_checkpoint
    cfg = yaml.safe_load(open(train_rate.DEFAULT_CONFIG))
    cfg["training"].update(n_envs=256, total_timesteps=256 * 8 * 6, eval_freq=16, save_freq=24)      # in vec-env steps
    cfg["curriculum"]["enabled"] = False
    cfg["ppo"].update(n_steps=8, n_epochs=1)
    cfg["evaluation"]["n_eval_episodes"] = 8
    cfg["paths"] = {"model_save_dir": str(tmp_path / "ckpt"), "tensorboard_log": str(tmp_path / "tb"),
                    "best_model_path": str(tmp_path / "best")}
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    train_rate.main(["--config", str(p), "--callbacks"])
    saved = sorted(f for f in os.listdir(tmp_path / "ckpt") if f.startswith("rate_controller_"))
    assert saved == ["rate_controller_12288_steps.pt", "rate_controller_6144_steps.pt"], saved       # every 24 vec-steps
    ev = np.load(tmp_path / "best" / "evaluations.npz")
    assert list(ev["timesteps"]) == [4096, 8192, 12288] and ev["results"].shape == (3, 8) and ev["ep_lengths"].shape == (3, 8)
    assert (tmp_path / "best" / "best_model.pt").exists()
    import json
    rows = [json.loads(l) for l in open(tmp_path / "tb" / "progress.jsonl")]
    assert [r["timesteps"] for r in rows] == [2048 * k for k in range(1, 7)] and "approx_kl" in rows[0]
    # the same scalars as a TensorBoard event file under SB3's tags (what visualize/learning_curves.py:35-121 loads)
    from hcrl_amd import tfevents
    tb = tfevents.load_scalars(str(tmp_path / "tb"))
    for tag in ("train/value_loss", "train/policy_gradient_loss", "train/approx_kl", "train/clip_fraction", "train/entropy_loss",
                "train/std", "train/learning_rate", "time/fps"):
        assert [r[0] for r in tb[tag]] == [2048 * k for k in range(1, 7)], tag
    assert [r[1] for r in tb["train/approx_kl"]] == pytest.approx([r["approx_kl"] for r in rows], rel=1e-6, abs=1e-9)
    assert [r[0] for r in tb["eval/mean_reward"]] == [4096, 8192, 12288]
    assert [r[1] for r in tb["eval/mean_reward"]] == pytest.approx(ev["results"].mean(1), rel=1e-6)
    assert all("episodes" in r for r in rows)                     # Monitor-style episode bookkeeping ran every iteration
    best = find_best_checkpoint(str(tmp_path / "best"))
    assert best[0] in (4096, 8192, 12288) and abs(best[1] - ev["results"].mean(1).max()) < 1e-6
    train_rate.main(["--config", str(p), "--resume", str(tmp_path / "ckpt" / "rate_controller_6144_steps.pt")])
    ck = torch.load(tmp_path / "ckpt" / "final_model.pt", weights_only=True)
    assert ck["num_timesteps"] == 6144 + 256 * 8 * 6


def test_curriculum_rebuilds_the_callbacks_on_each_phase_task(tmp_path, monkeypatch):
    """learned_controllers/train_rate.py:150-170: eval env and callbacks are recreated inside every curriculum phase, so the
    periodic evaluation flies the phase's difficulty and command type and the best-reward baseline restarts; here one progress
    logger and one evaluation history run through all phases."""
    import yaml
    from hcrl_amd import train_rate, training_utils as tu2
    cfg = yaml.safe_load(open(train_rate.DEFAULT_CONFIG))
    cfg["training"].update(n_envs=256, eval_freq=8, save_freq=1000)
    cfg["ppo"].update(n_steps=8, n_epochs=1)
    cfg["evaluation"]["n_eval_episodes"] = 4
    cfg["curriculum"] = {"enabled": True, "phases": [
        {"name": "a", "difficulty": "easy", "command_type": "step", "timesteps": 256 * 8 * 2},
        {"name": "b", "difficulty": "hard", "command_type": "random", "timesteps": 256 * 8 * 2}]}
    cfg["paths"] = {"model_save_dir": str(tmp_path / "ckpt"), "tensorboard_log": str(tmp_path / "tb"),
                    "best_model_path": str(tmp_path / "best")}
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(cfg))
    seen, real = [], tu2.run_final_evaluation

    def spy(model, **kw):
        seen.append((kw.get("difficulty"), kw.get("command_type"), int(model.num_timesteps)))
        return real(model, **kw)

    monkeypatch.setattr(tu2, "run_final_evaluation", spy)
    monkeypatch.setattr(train_rate, "run_final_evaluation", spy)
    train_rate.main(["--config", str(p), "--callbacks"])
    assert seen == [("easy", "step", 2048), ("easy", "step", 4096), ("hard", "random", 6144), ("hard", "random", 8192),
                    ("hard", "random", 8192)]                    # two periodic evaluations per phase + the final one
    ev = np.load(tmp_path / "best" / "evaluations.npz")
    assert list(ev["timesteps"]) == [2048, 4096, 6144, 8192]      # one history across the phases
    import json
    rows = [json.loads(l) for l in open(tmp_path / "tb" / "progress.jsonl")]
    assert [r["timesteps"] for r in rows] == [2048 * k for k in range(1, 5)]     # ONE progress logger across the phases


def test_gaussian_head_statistics_and_logprob():
    """Fused sampling head: z = (a - mean)/std must be ~N(0,1), log_prob must equal the closed form, and two successive
    calls (device-side step counter) must draw different noise."""
    import math
    from hcrl_amd import _lib
    lib = _lib.load()
    B = 200000
    mean = torch.randn(B, 4, device="cuda").bfloat16()
    log_std = torch.tensor([0.0, -0.5, 0.3, -1.0], device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    outs = []
    for _ in range(2):
        step.add_(1)
        a = torch.empty(B, 4, device="cuda"); lp = torch.empty(B, device="cuda")
        _lib.check(lib.fdyn_gaussian_head(mean.data_ptr(), 1, log_std.data_ptr(), 1234, step.data_ptr(), 0, a.data_ptr(),
                                          lp.data_ptr(), B, _lib.current_stream()))
        outs.append((a, lp))
    a, lp = outs[0]
    z = (a - mean.float()) / log_std.exp()
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert abs(float((z[:, 0] * z[:, 1]).mean())) < 0.01                       # Box-Muller pairs uncorrelated
    ref = (-0.5 * z ** 2 - log_std - 0.5 * math.log(2 * math.pi)).sum(1)
    assert (lp - ref).abs().max() < 1e-3
    assert not torch.equal(outs[0][0], outs[1][0])
    d = torch.empty(B, 4, device="cuda")
    _lib.check(lib.fdyn_gaussian_head(mean.data_ptr(), 1, log_std.data_ptr(), 1234, step.data_ptr(), 1, d.data_ptr(),
                                      lp.data_ptr(), B, _lib.current_stream()))
    assert torch.equal(d, mean.float())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("M", [8200, 300000])
def test_zero_state_layer_three_gate_matches_plain_torch_fp32(dtype, tol, M):
    """The features extractor's LSTM layer as one node in the three-gate layout (fused.zero_state_lstm_layer: no forget gate,
    bias gradient from per-block partial sums, split-K dW) against nn.LSTM's arithmetic in plain fp32 torch on the same
    (rounded) inputs: h, dX, dW_ih (f rows exactly zero), db_ih = db_hh."""
    from hcrl_amd import fused
    torch.manual_seed(11)
    K, H = 128, 256
    x = (torch.randn(M, K, device="cuda") * 0.7).to(dtype)
    w = (torch.randn(4 * H, K, device="cuda") * 0.1).requires_grad_()
    bi, bh = (torch.randn(4 * H, device="cuda") * 0.1).requires_grad_(), (torch.randn(4 * H, device="cuda") * 0.1).requires_grad_()
    wh = torch.randn(M, H, device="cuda") / M ** 0.5
    xr = x.detach().float().requires_grad_()
    wr = w.detach().to(dtype).float().requires_grad_()              # the layer multiplies in `dtype`
    bsum = (bi + bh).detach().to(dtype).float()
    br = bsum.clone().requires_grad_()
    g = xr @ wr.t() + br
    if dtype == torch.bfloat16:
        g = g + (g.detach().to(dtype).float() - g.detach())        # the GEMM result is stored in bf16: same values, gradient of g
    i, f, gg, o = g.chunk(4, -1)
    h_ref = torch.sigmoid(o) * torch.tanh(torch.sigmoid(i) * torch.tanh(gg))
    (h_ref * wh).sum().backward()
    xq = x.detach().clone().requires_grad_()
    h = fused.zero_state_lstm_layer(xq, w, bi, bh)
    assert h.dtype == dtype and h.shape == (M, H)
    (h.float() * wh).sum().backward()
    assert float((h.float() - h_ref).abs().max()) < tol
    rel = lambda a, b: float((a.float() - b).norm() / (b.norm() + 1e-20))      # noqa: E731
    assert rel(xq.grad, xr.grad) < tol * 2
    assert rel(w.grad, wr.grad) < tol * 2 and float(w.grad[H:2 * H].abs().max()) == 0.0
    assert rel(bi.grad, br.grad) < tol * 2 and torch.equal(bi.grad, bh.grad) and float(bi.grad[H:2 * H].abs().max()) == 0.0
    assert float(wr.grad[H:2 * H].abs().max()) == 0.0              # the reference agrees: the forget gate has no gradient


@pytest.mark.parametrize("dtype,tol", [(None, 2e-4), (torch.bfloat16, 6e-2)])
def test_deferred_splitk_weight_gradients_match_plain_autograd(dtype, tol):
    """The BPTT path with deferred / split-K weight gradients (fused.DeferredWgrad, fused.linear) against plain autograd
    through F.linear on the same parameters and batch: every parameter gradient, relative to its norm."""
    from hcrl_amd import fused
    torch.manual_seed(3)
    pol = RateLSTMPolicy(compute_dtype=dtype).cuda()
    T, B = 6, 8192
    obs = torch.randn(T, B, 18, device="cuda")
    act = torch.randn(T, B, 4, device="cuda").clamp(-1, 1)
    starts = (torch.rand(T, B, device="cuda") < 0.05).float()
    st = pol.initial_state(B, "cuda")
    st = type(st)(*[torch.randn_like(s) * 0.3 for s in st])
    adv = torch.randn(T, B, device="cuda")

    def grads(deferred, min_rows, sequence=False):
        pol.deferred_wgrad, pol.sequence_bptt = deferred, sequence
        pol.zero_grad(set_to_none=True)
        orig = fused.linear.__defaults__
        fused.linear.__defaults__ = (min_rows,)
        try:
            v, lp, ent = pol.evaluate_sequence(obs, act, starts, st)
            ((lp * adv).mean() + 0.5 * (v ** 2).mean() - 0.01 * ent).backward()
        finally:
            fused.linear.__defaults__ = orig
        return {n: p.grad.detach().float().clone() for n, p in pol.named_parameters() if p.grad is not None}   # zero-state W_hh: unused

    plain, fast, seq = grads(False, 1 << 60), grads(True, 8192), grads(True, 8192, sequence=True)    # seq: both cells in one node
    pol.group_cells_max_batch = 0
    seq1 = grads(True, 8192, sequence=True)                   # one node per cell (the large-slice path)
    pol.group_cells_max_batch = 8192
    # every BPTT forward through the fused MFMA cell (fdyn_lstm_cell_mfma_train; bf16 only -- the default uses it for the
    # features extractor's layers alone), and none at all
    mode = fused.MFMA_TRAIN_DEFAULT
    try:
        fused.MFMA_TRAIN_DEFAULT = "all"
        seq_mfma = grads(True, 8192, sequence=True)
        fused.MFMA_TRAIN_DEFAULT = "0"
        seq_plain_fwd = grads(True, 8192, sequence=True)
    finally:
        fused.MFMA_TRAIN_DEFAULT = mode
    assert set(plain) == set(fast) == set(seq) == set(seq1) == set(seq_mfma) == set(seq_plain_fwd)
    for n in plain:
        for other in (fast, seq, seq1, seq_mfma, seq_plain_fwd):      # seq: the whole recurrence as one autograd node
            a, b = plain[n], other[n]
            err = float((a - b).norm() / (a.norm() + 1e-12))
            assert err < tol, (n, err, other is seq, other is seq1)
    # flat-buffer gradients (FlatGrad: p.grad are views that autograd and the deferred flush must ADD into)
    from hcrl_amd.ppo import FlatGrad
    flat = FlatGrad(pol)
    flat.zero()
    pol.deferred_wgrad, pol.sequence_bptt = True, False
    v, lp, ent = pol.evaluate_sequence(obs, act, starts, st)
    ((lp * adv).mean() + 0.5 * (v ** 2).mean() - 0.01 * ent).backward()
    for n, p in pol.named_parameters():
        if n not in fast:
            continue
        assert p.grad.data_ptr() >= flat.buf.data_ptr() and p.grad.data_ptr() < flat.buf.data_ptr() + flat.buf.numel() * 4, n
        err = float((p.grad.float() - fast[n]).norm() / (fast[n].norm() + 1e-12))
        assert err < (1e-5 if dtype is None else 2e-2), (n, err)


def test_graphed_update_equals_eager_update():
    """The PPO update with forward+backward replayed from a hipGraph over static slice buffers against the eager loop on
    the SAME rollout, parameters, optimizer state and slice permutations (fp32 policy): same parameters after two epochs."""
    import copy
    env = GpuRateVecEnv(4096, "easy", 10.0, 0.02, "step", seed=11, precision="mixed", sampling="device")
    m = RecurrentPPO(env, RateLSTMPolicy(), PPOConfig(n_steps=8, n_epochs=2, n_minibatches=2), seed=3, use_graph=False)
    m.collect_rollout()
    p0 = copy.deepcopy(m.policy.state_dict())
    o0 = copy.deepcopy(m.opt.state_dict())

    def run(graph):
        m.policy.load_state_dict(p0)
        m.opt.load_state_dict(copy.deepcopy(o0))
        m.use_update_graph = graph
        torch.manual_seed(99)                                # the slice permutations
        st = m.update()
        assert m.use_update_graph == graph                   # capture must not have fallen back
        return [p.detach().clone() for p in m.policy.parameters()], st

    pe, se = run(False)
    pg, sg = run(True)
    moved = max(float((a - b).abs().max()) for a, b in zip(pe, p0.values()) if a.shape == b.shape)
    assert moved > 1e-4                                      # the update did something
    for a, b in zip(pe, pg):
        assert torch.allclose(a, b, atol=5e-5, rtol=1e-3), float((a - b).abs().max())
    for k in se:
        assert abs(se[k] - sg[k]) <= 1e-3 * max(1.0, abs(se[k])), (k, se[k], sg[k])


def test_fused_ppo_loss_and_colsum_match_torch():
    """fdyn_ppo_loss (loss, statistics, gradients w.r.t. action means / values / log_std) against the same arithmetic in
    plain torch ops on the CPU in fp64-free fp32; fdyn_colsum against torch.sum."""
    from hcrl_amd import fused
    torch.manual_seed(0)
    T, B = 16, 4096
    mean = torch.randn(T, B, 4, device="cuda") * 0.3
    act = (mean + 0.5 * torch.randn_like(mean)).clamp(-1, 1)
    log_std = torch.tensor([-0.7, -0.2, 0.1, -1.0], device="cuda")
    values, ret = torch.randn(T, B, device="cuda"), torch.randn(T, B, device="cuda")
    adv = torch.randn(T, B, device="cuda") * 3 + 0.5
    old_v = values + 0.3 * torch.randn_like(values)
    with torch.no_grad():
        var = (2 * log_std).exp()
        old_logp = (-((act - mean) ** 2) / (2 * var) - log_std - 0.9189385332).sum(-1) + 0.2 * torch.randn(T, B, device="cuda")
    for clip_vf in (None, 0.2):
        outs = []
        for dev in ("cuda", "cpu"):
            m, v, ls = (t.to(dev).clone().requires_grad_(True) for t in (mean, values, log_std))
            loss, st = fused.ppo_loss(m, v, ls, act.to(dev), old_logp.to(dev), adv.to(dev), ret.to(dev), old_v.to(dev), True, 0.2,
                                      clip_vf, 0.5, 0.01)
            loss.backward()
            outs.append([t.detach().cpu() for t in (loss, st, m.grad, v.grad, ls.grad)])
        for a, b, name in zip(outs[0], outs[1], ("loss", "stats", "dmean", "dvalues", "dlog_std")):
            assert torch.allclose(a, b, rtol=2e-4, atol=2e-6 if name.startswith("d") else 2e-5), (name, clip_vf, float((a - b).abs().max()))
    for dt in (torch.float32, torch.bfloat16):
        for n in (384, 1024, 18, 4, 1):                      # 16-byte-vector path and the one-lane-per-column path
            x = torch.randn(70001, n, device="cuda").to(dt)
            got, want = fused.colsum(x), x.double().sum(0)
            assert float((got.double() - want).abs().max()) < 2e-2, (dt, n)
        for (t_, g_, b_, n) in ((7, 2, 2048, 1024), (5, 3, 768, 128), (4, 2, 100, 64)):      # [T][G][B] row order
            x = torch.randn(t_ * g_ * b_, n, device="cuda").to(dt)
            got = fused.colsum(x, b_, g_)
            want = x.double().view(t_, g_, b_, n).sum((0, 2))
            assert got.shape == (g_, n) and float((got.double() - want).abs().max()) < 2e-2, (dt, t_, g_, b_, n)


@pytest.mark.parametrize("N,T,n_mb", [(16384, 64, 8), (512, 8, 4)])
def test_update_graph_replays_match_eager_at_training_size(N, T, n_mb):
    """Six replays of the captured forward+backward on fresh slices (16 384 envs x 64 steps, bf16 policy) against the eager
    pass on the same static buffers: every parameter gradient and the loss statistics.  (Guards the graph path against
    stale / uninitialised reductions under replay.)  The SMALL configuration (512 envs x 8 steps in 4 slices = 1024 rows per
    layer) is the one that used to fall below fused.linear's row threshold and take the framework's bias-gradient reduction:
    garbage from the second replay on, NaN weights in the third PPO iteration."""
    env = GpuRateVecEnv(N, "easy", 10.0, 0.02, "step", seed=42, precision="mixed", sampling="device")
    m = RecurrentPPO(env, RateLSTMPolicy(compute_dtype=torch.bfloat16), PPOConfig(n_steps=T, n_epochs=1, n_minibatches=n_mb,
                                                                                  reward_scale=0.02), seed=42, use_graph=False)
    m.collect_rollout()
    mb = N // n_mb
    ug = m._build_update_graph(mb)
    for trial in range(6):
        idx = torch.randperm(N, device="cuda")[:mb]
        for k, src in (("obs", m.buf_obs), ("act", m.buf_act), ("starts", m.buf_start), ("adv", m.adv), ("ret", m.ret),
                       ("old_logp", m.buf_logp), ("old_v", m.buf_val)):
            torch.index_select(src, 1, idx, out=ug[k])
        for dst, src in zip(ug["states"], m.rollout_states):
            torch.index_select(src, 0, idx, out=dst)
        ug["graph"].replay()
        g_flat, g_st = m.flat.buf.clone(), ug["stats"].clone()
        m.flat.zero()
        loss, st = m._minibatch_loss(ug["obs"], ug["act"], ug["starts"], ug["adv"], ug["ret"], ug["old_logp"], ug["old_v"], ug["states"])
        loss.backward()
        e_flat = m.flat.buf.clone()
        assert torch.allclose(g_st, st, rtol=1e-3, atol=1e-5), (trial, g_st.tolist(), st.tolist())
        off = 0
        for (n, p) in zip([n for n, _ in m.policy.named_parameters()], m.flat.params):
            a, b = g_flat[off:off + p.numel()], e_flat[off:off + p.numel()]
            off += p.numel()
            err = float((a - b).norm() / (b.norm() + 1e-9))
            assert err < 2e-2 and bool(torch.isfinite(a).all()), (trial, n, err)


@pytest.mark.parametrize("residual", [False, True])
def test_train_with_imitation_cli(tmp_path, residual):
    """train_with_imitation.py:51-199 end to end on a tiny budget: demonstrations -> BC -> PPO (optionally on the residual env)
    -> checkpoints + evaluation."""
    from hcrl_amd import train_with_imitation as twi
    argv = ["--n-demos", "64", "--bc-epochs", "2", "--rl-steps", str(512 * 16 * 3), "--n-envs", "512", "--n-steps", "16",
            "--model-dir", str(tmp_path / "m"), "--demo-path", str(tmp_path / "demos.npz")] + (["--residual"] if residual else [])
    ev = twi.main(argv)
    assert np.isfinite(ev["mean_reward"]) and 0 < ev["mean_length"] <= 500
    assert (tmp_path / "m" / "final_model.pt").exists() and (tmp_path / "m" / "bc_pretrained.pt").exists()
    assert (tmp_path / "demos.npz").exists()
    ev2 = twi.main(argv + ["--rl-steps", str(512 * 16)])          # second run re-uses the stored demonstrations
    assert np.isfinite(ev2["mean_reward"])


def test_train_rate_accepts_the_overnight_schema(tmp_path, capsys):
    """A train_overnight.py-style YAML (network / parallel / approach / demonstrations ...) through train_rate: imitation from
    the file's own settings, MLP policy, two curriculum phases."""
    import yaml
    from hcrl_amd import train_rate
    cfg = yaml.safe_load("""
approach: {use_imitation: true, use_residual: false, use_curriculum: true}
demonstrations: {n_episodes: 64, difficulty: easy, save_path: DEMO_PATH}
behavior_cloning: {epochs: 2, batch_size: 256, learning_rate: 0.001}
curriculum:
  phases:
    - {name: easy, difficulty: easy, timesteps: 8192, command_type: step}
    - {name: hard_mixed, difficulty: hard, timesteps: 8192, command_type: random}
environment: {episode_length: 10.0, dt: 0.02}
ppo: {learning_rate: 0.0003, n_steps: 16, batch_size: 256, n_epochs: 1, gamma: 0.99, gae_lambda: 0.95, clip_range: 0.2,
      ent_coef: 0.01, vf_coef: 0.5, max_grad_norm: 0.5}
network: {type: mlp, mlp: {net_arch: [64, 64]}, lstm: {hidden_size: 256, n_layers: 2}}
parallel: {n_envs: 512, vec_env_type: subproc}
evaluation: {eval_freq: 100000, n_eval_episodes: 4, deterministic: true}
checkpointing: {save_freq: 500000, keep_last_n: 5}
logging: {log_interval: 10, verbose: 1}
seed: 7
""")
    cfg["paths"] = {"model_dir": str(tmp_path / "m"), "tensorboard_log": str(tmp_path / "tb"), "best_model": str(tmp_path / "best")}
    cfg["demonstrations"]["save_path"] = str(tmp_path / "demos" / "pid_demos.pkl")
    p = tmp_path / "overnight.yaml"
    p.write_text(yaml.safe_dump(cfg))
    train_rate.main(["--config", str(p)])
    ck = torch.load(tmp_path / "m" / "final_model.pt", weights_only=True)
    assert ck["num_timesteps"] == 2 * 8192 and not any(k.startswith("lstm_actor") for k in ck["policy"])
    # train_overnight.py:116-183: the demonstrations are kept (data-only .npz) and the cloned policy is saved before PPO
    demos = np.load(tmp_path / "demos" / "pid_demos.pkl.npz")
    assert demos["observations"].shape[1] == 18 and demos["actions"].shape == (len(demos["observations"]), 4)
    bc = torch.load(tmp_path / "m" / "bc_pretrained.pt", weights_only=True)
    assert bc["num_timesteps"] == 0
    # the reference's entry point: same flags, callbacks always on; --skip-demos reuses the file, --skip-bc skips cloning
    from hcrl_amd import train_overnight
    stamp = os.path.getmtime(tmp_path / "demos" / "pid_demos.pkl.npz")
    os.remove(tmp_path / "m" / "bc_pretrained.pt")
    train_overnight.main(["--config", str(p), "--skip-demos"])
    assert os.path.getmtime(tmp_path / "demos" / "pid_demos.pkl.npz") == stamp and (tmp_path / "m" / "bc_pretrained.pt").exists()
    out = capsys.readouterr().out
    assert "OVERNIGHT TRAINING" in out and "Loading existing demos" in out and "TRAINING COMPLETE!" in out
    assert (tmp_path / "tb" / "progress.jsonl").exists() and any(f.startswith("events.out.tfevents.") for f in os.listdir(tmp_path / "tb"))
    os.remove(tmp_path / "m" / "bc_pretrained.pt")
    train_overnight.main(["--config", str(p), "--skip-bc"])
    assert not (tmp_path / "m" / "bc_pretrained.pt").exists() and "BC losses" not in capsys.readouterr().out


def test_episode_statistics_of_training_rollouts():
    """rollout/ep_rew_mean, ep_len_mean: with 0.4 s episodes every env finishes episodes inside each rollout; lengths must be
    the truncation length and the mean return must equal the return computed from the buffers episode by episode."""
    from hcrl_amd.policy import RateLSTMPolicy
    from hcrl_amd.ppo import PPOConfig, RecurrentPPO
    from hcrl_amd.rate_env import GpuRateVecEnv
    env = GpuRateVecEnv(64, "easy", 0.4, 0.02, "step", seed=3, precision="mixed", sampling="device")      # 20-step episodes
    ppo = RecurrentPPO(env, RateLSTMPolicy(compute_dtype=torch.bfloat16), PPOConfig(n_steps=32, n_epochs=1, n_minibatches=1), seed=0)
    ppo.track_episode_stats = True
    seen = []
    for _ in range(3):
        ppo.collect_rollout()
        rew = (ppo.buf_rew / ppo.cfg.reward_scale).double().cpu().numpy()
        starts = ppo.buf_start.cpu().numpy()
        final = ppo.episode_start.cpu().numpy()
        st = ppo.update()
        seen.append((st["ep_rew_mean"], st["ep_len_mean"], st["episodes"], rew, starts, final))
    run_ret, run_len = np.zeros(64), np.zeros(64)
    for mean_ret, mean_len, count, rew, starts, final in seen:
        rets, lens = [], []
        for t in range(rew.shape[0]):
            done = starts[t + 1] if t + 1 < rew.shape[0] else final
            run_ret += rew[t]; run_len += 1
            for n in np.nonzero(done > 0)[0]:
                rets.append(run_ret[n]); lens.append(run_len[n]); run_ret[n] = 0.0; run_len[n] = 0
        assert count == len(rets) and count >= 64
        assert mean_ret == pytest.approx(np.mean(rets), rel=1e-9) and mean_len == pytest.approx(np.mean(lens))
        assert max(lens) <= 20                                       # truncation at episode_length / dt steps


@pytest.mark.parametrize("B", [32768, 65536])
def test_policy_features_kernel_matches_plain_torch_fp32(B):
    """csrc/policy_fe64.hip (observation -> features in one kernel, activations in registers between the four layers) against
    plain PyTorch fp32 on the same bf16-rounded weights with the SAME rounding points (every layer's output rounded to bf16),
    and against the layer-wise device path it replaces.  Random weights with non-zero biases and an asymmetric observation
    batch: a wrong k permutation, fragment order or gate offset cannot pass."""
    from hcrl_amd import _lib
    from hcrl_amd.policy import FE_GATE_SCALE
    lib = _lib.load()
    torch.manual_seed(11)
    p = RateLSTMPolicy(compute_dtype=torch.bfloat16).cuda()
    fe = p.features_extractor
    with torch.no_grad():
        for prm in fe.parameters():
            prm.copy_(torch.randn_like(prm) * (0.5 if prm.dim() == 1 else 1.6 / prm.shape[-1] ** 0.5))
    p.prepare_inference()
    inf = p._inf
    obs = torch.randn(B, 18, device="cuda") * torch.linspace(0.2, 2.0, 18, device="cuda")
    feats = torch.empty((B, 128), dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.fdyn_policy_features(obs.data_ptr(), inf["fe_img"].data_ptr(), inf["fe_bias"].data_ptr(), feats.data_ptr(), B,
                                        _lib.current_stream()), "policy_features")
    torch.cuda.synchronize()
    bf = lambda t: t.to(torch.bfloat16).float()      # noqa: E731
    x = bf(torch.relu(bf(obs) @ bf(fe.embedding[0].weight).t() + fe.embedding[0].bias))
    for k in range(2):
        # the weight image holds the LSTM rows scaled by their gate's exponent factor and THEN rounded to bf16 (FE_GATE_SCALE)
        gs = torch.tensor(FE_GATE_SCALE, device="cuda").repeat_interleave(256)[:, None]
        w = bf(getattr(fe.lstm, f"weight_ih_l{k}") * gs) / gs; b = getattr(fe.lstm, f"bias_ih_l{k}") + getattr(fe.lstm, f"bias_hh_l{k}")
        i, _, g, o = (x @ w.t() + b).chunk(4, 1)
        x = bf(torch.sigmoid(o) * torch.tanh(torch.sigmoid(i) * torch.tanh(g)))
    ref = torch.relu(x @ bf(fe.output_proj[0].weight).t() + fe.output_proj[0].bias)
    err = (feats.float() - ref).abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err
    assert (feats.float() - ref).abs().mean().item() < 2e-3
    # the layer-wise path (hipBLASLt GEMMs + one MFMA cell kernel per layer)
    os.environ["FDYN_NO_FE64"] = "1"
    try:
        st = p.initial_state(B, "cuda")
        keep = torch.ones(B, device="cuda")
        with torch.no_grad():
            x2 = p._mlp_bf16(obs.to(torch.bfloat16), inf["emb"])
            for w, b in zip(inf["fe_w"], inf["fe_b"]):
                h = torch.empty((B, 256), dtype=torch.bfloat16, device="cuda")
                _lib.check(lib.fdyn_lstm_cell_mfma(x2.data_ptr(), x2.shape[1], None, 0, None, None, w.data_ptr(), b.data_ptr(),
                                                   h.data_ptr(), None, None, B, 256, _lib.current_stream()), "cell")
                x2 = h
            layerwise = p._mlp_bf16(x2, inf["proj"])
    finally:
        del os.environ["FDYN_NO_FE64"]
    assert (feats.float() - layerwise.float()).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,inplace", [(256, False), (1024, True), (65536, True)])
def test_recurrent_cells_one_launch_matches_plain_torch_fp32(B, inplace):
    """csrc/policy_rc64.hip -- actor and critic nn.LSTM(128, 256) cells in one launch, lane = batch row, state in the kernel's own
    layouts, updated in place -- against the same cells in plain fp32 PyTorch on the bf16-rounded operands."""
    from hcrl_amd import _lib
    from hcrl_amd.policy import pack_rc_weights, rc_pack_c, rc_pack_h, rc_pack_x, rc_unpack_c, rc_unpack_h
    lib = _lib.load()
    dev, bf = torch.device("cuda"), torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(B)
    rn = lambda *s, k=1.0: torch.randn(*s, device=dev, generator=g) * k          # noqa: E731
    x = rn(B, 128, k=0.7).to(bf)
    keep = (torch.rand(B, device=dev, generator=g) > 0.05).float()
    cells, ref = [], []
    for _ in range(2):
        w_ih, w_hh, bias = rn(1024, 128, k=0.08).to(bf), rn(1024, 256, k=0.08).to(bf), rn(1024, k=0.3)
        h, c = rn(B, 256, k=0.5).to(bf), rn(B, 256)
        cells.append((w_ih, w_hh, bias, h, c))
        gates = torch.cat([x.float(), h.float() * keep[:, None]], 1) @ torch.cat([w_ih, w_hh], 1).float().t() + bias
        i, f, gg, o = gates.chunk(4, 1)
        c2 = torch.sigmoid(f) * (c * keep[:, None]) + torch.sigmoid(i) * torch.tanh(gg)
        ref.append((torch.sigmoid(o) * torch.tanh(c2), c2))
    img = pack_rc_weights([(c_[0], c_[1]) for c_ in cells])
    assert img.numel() * 2 == lib.fdyn_policy_recurrent_image_bytes()
    bias = torch.stack([c_[2] for c_ in cells]).contiguous()
    xi = rc_pack_x(x)
    hi = [rc_pack_h(c_[3]) for c_ in cells]
    ci = [rc_pack_c(c_[4]) for c_ in cells]
    assert torch.equal(rc_unpack_h(hi[0]), cells[0][3]) and torch.equal(rc_unpack_c(ci[1]), cells[1][4])
    ho = hi if inplace else [torch.empty_like(t) for t in hi]
    co = ci if inplace else [torch.empty_like(t) for t in ci]
    _lib.check(lib.fdyn_policy_recurrent(xi.data_ptr(), keep.data_ptr(), img.data_ptr(), bias.data_ptr(),
                                         hi[0].data_ptr(), ci[0].data_ptr(), ho[0].data_ptr(), co[0].data_ptr(),
                                         hi[1].data_ptr(), ci[1].data_ptr(), ho[1].data_ptr(), co[1].data_ptr(), B,
                                         _lib.current_stream()), "policy_recurrent")
    torch.cuda.synchronize()
    for k in range(2):
        h2, c2 = rc_unpack_h(ho[k]).float(), rc_unpack_c(co[k])
        assert torch.isfinite(h2).all() and torch.isfinite(c2).all()
        eh, ec = (h2 - ref[k][0]).abs().max().item(), (c2 - ref[k][1]).abs().max().item()
        assert eh < 6e-3 and ec < 2e-4, (k, eh, ec)            # h' is rounded to bf16 (2^-9 relative), c' stays fp32


def test_recurrent_state_in_place_equals_ping_pong():
    """One recurrent-state set updated IN PLACE (what RecurrentPPO uses at batches the one-wave-per-SIMD cell kernel serves)
    gives bit for bit what the ping-pong pair gives; shapes whose kernel splits a row over workgroups refuse aliased state."""
    from hcrl_amd import _lib
    from hcrl_amd.policy import RNNStates
    torch.manual_seed(1)
    p = RateLSTMPolicy(compute_dtype=torch.bfloat16).cuda()
    p.prepare_inference()
    B = 65536
    assert p.recurrent_inplace_ok(B, "cuda") and not p.recurrent_inplace_ok(4096, "cuda")
    obs = torch.randn(B, 18, device="cuda")
    mk = lambda: RNNStates(*(torch.randn(B, 256, device="cuda").to(dt) * 0.5 for dt in (torch.bfloat16, torch.float32) * 2))  # noqa: E731
    st = mk()
    st2 = RNNStates(*(t.clone() for t in st))
    out = RNNStates(*(torch.empty_like(t) for t in st))
    start = (torch.rand(B, device="cuda") < 0.02).float()
    with torch.no_grad():
        a1, v1, _, s1 = p.step(obs, st, start, deterministic=True, out_states=out)
        a2, v2, _, s2 = p.step(obs, st2, start, deterministic=True, out_states=st2)
    assert all(a.data_ptr() == b.data_ptr() for a, b in zip(s2, st2))          # really in place
    assert all(torch.equal(a, b) for a, b in zip(s1, s2)) and torch.equal(a1, a2) and torch.equal(v1, v2)
    lib, n = _lib.load(), 512
    x, h, c = torch.zeros(n, 128, device="cuda", dtype=torch.bfloat16), torch.zeros(n, 256, device="cuda", dtype=torch.bfloat16), torch.zeros(n, 256, device="cuda")
    w, b = torch.zeros(1024, 384, device="cuda", dtype=torch.bfloat16), torch.zeros(1024, device="cuda")
    rc = lib.fdyn_lstm_cell_mfma(x.data_ptr(), 128, h.data_ptr(), 256, c.data_ptr(), None, w.data_ptr(), b.data_ptr(), h.data_ptr(),
                                 c.data_ptr(), None, n, 256, _lib.current_stream())
    assert rc == _lib.FDYN_ERR_BAD_SIZE


@pytest.mark.parametrize("B,inplace", [(65536, False), (65536, True), (32768, True), (512, False)])
def test_cell_pair_in_one_launch_equals_two_launches(B, inplace):
    """fdyn_lstm_cell_mfma_pair (actor and critic cell, one launch of 2 x B / 256 workgroups where the one-wave-per-SIMD kernel
    applies; two calls elsewhere) against two calls of fdyn_lstm_cell_mfma: the same kernel on the same operands, bit for bit --
    out of place and with the state updated in place; aliased state at a shape the paired kernel does not serve is refused."""
    from hcrl_amd import _lib
    torch.manual_seed(B + int(inplace))
    lib, bf, st = _lib.load(), torch.bfloat16, _lib.current_stream()
    x = (torch.randn(B, 128, device="cuda") * 0.7).to(bf)
    keep = (torch.rand(B, device="cuda") > 0.05).float()
    W = [(torch.randn(1024, 384, device="cuda") * 0.06).to(bf) for _ in range(2)]
    bias = [torch.randn(1024, device="cuda") * 0.3 for _ in range(2)]
    h0 = [(torch.randn(B, 256, device="cuda") * 0.5).to(bf) for _ in range(2)]
    c0 = [torch.randn(B, 256, device="cuda") for _ in range(2)]
    ref = []
    for g in range(2):
        h, c = torch.empty_like(h0[g]), torch.empty_like(c0[g])
        _lib.check(lib.fdyn_lstm_cell_mfma(x.data_ptr(), 128, h0[g].data_ptr(), 256, c0[g].data_ptr(), keep.data_ptr(), W[g].data_ptr(),
                                           bias[g].data_ptr(), h.data_ptr(), c.data_ptr(), None, B, 256, st), "lstm_cell_mfma")
        ref += [h, c]
    hin, cin = [t.clone() for t in h0], [t.clone() for t in c0]
    hout = hin if inplace else [torch.empty_like(t) for t in h0]
    cout = cin if inplace else [torch.empty_like(t) for t in c0]
    args = [t.data_ptr() for g in range(2) for t in (hin[g], cin[g], W[g], bias[g], hout[g], cout[g])]
    _lib.check(lib.fdyn_lstm_cell_mfma_pair(x.data_ptr(), 128, keep.data_ptr(), 256, B, 256, *args, st), "lstm_cell_mfma_pair")
    torch.cuda.synchronize()
    for g in range(2):
        assert torch.equal(hout[g], ref[2 * g]) and torch.equal(cout[g], ref[2 * g + 1]), g
    assert not torch.equal(ref[0], ref[2])                                       # the two cells really differ
    if B == 512:
        args = [t.data_ptr() for g in range(2) for t in (hin[g], cin[g], W[g], bias[g], hin[g], cin[g])]
        assert lib.fdyn_lstm_cell_mfma_pair(x.data_ptr(), 128, keep.data_ptr(), 256, B, 256, *args, st) == _lib.FDYN_ERR_BAD_SIZE


def test_policy_entry_points_validate_their_arguments():
    """The policy-step entry points added in round 3 return the library's error codes instead of launching on bad arguments:
    NULL operands, negative / unsupported sizes; an empty batch is a no-op."""
    from hcrl_amd import _lib
    lib, bf, st = _lib.load(), torch.bfloat16, _lib.current_stream()
    B = 512
    x = torch.zeros(B, 128, device="cuda", dtype=bf)
    h = [torch.zeros(B, 256, device="cuda", dtype=bf) for _ in range(4)]
    c = [torch.zeros(B, 256, device="cuda") for _ in range(4)]
    W = torch.zeros(1024, 384, device="cuda", dtype=bf); b = torch.zeros(1024, device="cuda")
    good = [h[0].data_ptr(), c[0].data_ptr(), W.data_ptr(), b.data_ptr(), h[1].data_ptr(), c[1].data_ptr(),
            h[2].data_ptr(), c[2].data_ptr(), W.data_ptr(), b.data_ptr(), h[3].data_ptr(), c[3].data_ptr()]
    pair = lib.fdyn_lstm_cell_mfma_pair
    assert pair(x.data_ptr(), 128, None, 256, B, 256, *good, st) == 0                       # keep may be NULL (nobody restarted)
    assert pair(x.data_ptr(), 128, None, 256, 0, 256, *good, st) == 0                       # empty batch
    assert pair(None, 128, None, 256, B, 256, *good, st) == _lib.FDYN_ERR_NULL
    assert pair(x.data_ptr(), 128, None, 256, B, 256, *good[:8], None, *good[9:], st) == _lib.FDYN_ERR_NULL
    assert pair(x.data_ptr(), 128, None, 256, -1, 256, *good, st) == _lib.FDYN_ERR_BAD_SIZE
    assert pair(x.data_ptr(), 128, None, 0, B, 256, *good, st) == _lib.FDYN_ERR_BAD_SIZE    # a pair of zero-state cells is not a thing
    assert pair(x.data_ptr(), 64, None, 256, B, 256, *good, st) == _lib.FDYN_ERR_BAD_SIZE   # unsupported input width
    w1 = torch.zeros(2, 128, 256, device="cuda", dtype=bf); b1 = torch.zeros(2, 128, device="cuda")
    w2 = torch.zeros(2, 64, 128, device="cuda", dtype=bf); b2 = torch.zeros(2, 64, device="cuda")
    wa = torch.zeros(4, 64, device="cuda", dtype=bf); ba = torch.zeros(4, device="cuda", dtype=bf)
    wv = torch.zeros(64, device="cuda", dtype=bf); bv = torch.zeros(1, device="cuda", dtype=bf)
    ls = torch.zeros(4, device="cuda"); act = torch.full((B, 4), 3.0, device="cuda"); lp = torch.empty(B, device="cuda"); val = torch.empty(B, device="cuda")
    th = lib.fdyn_policy_trunks_heads
    args = [h[0].data_ptr(), h[2].data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), wa.data_ptr(), ba.data_ptr(),
            wv.data_ptr(), bv.data_ptr(), ls.data_ptr(), 7, None, 1, act.data_ptr(), lp.data_ptr(), val.data_ptr()]
    assert th(*args, 0, st) == 0 and th(*args, -5, st) == _lib.FDYN_ERR_BAD_SIZE
    assert th(*args[:6], None, *args[7:], B, st) == _lib.FDYN_ERR_NULL
    assert th(*args, B, st) == 0                                                              # step counter NULL = counter 0
    torch.cuda.synchronize()
    assert float(act.abs().max()) == 0.0 and float(val.abs().max()) == 0.0                  # zero weights, deterministic: zeros out


@pytest.mark.parametrize("B", [300, 65536])
def test_heads_behind_the_trunks_equal_the_two_launch_path(B, monkeypatch):
    """fdyn_policy_trunks_heads (trunks -> output heads -> Gaussian sampling in one launch, lat never leaves the registers)
    against fdyn_policy_trunks + fdyn_policy_heads: same Philox key, same bf16 rounding of the trunk outputs; only the order of
    the 64-term head sums differs (fp32)."""
    torch.manual_seed(3)
    p = RateLSTMPolicy(compute_dtype=torch.bfloat16).cuda()
    with torch.no_grad():
        p.action_net.weight.mul_(30.0)                      # the 0.01-gain initialisation would leave the means ~0: make them visible
        p.log_std.fill_(-0.7)
    p.prepare_inference()
    obs = torch.randn(B, 18, device="cuda")
    st = p.initial_state(B, "cuda")
    st = type(st)(*((t + 0.3 * torch.randn_like(t)).to(dt) for t, dt in zip(st, (torch.bfloat16, torch.float32) * 2)))
    start = (torch.rand(B, device="cuda") < 0.1).float()
    with torch.no_grad():
        a1, v1, lp1, _ = p.step(obs, st, start, bump_noise=False)
        monkeypatch.setenv("FDYN_NO_TRUNK_HEADS", "1")
        a0, v0, lp0, _ = p.step(obs, st, start, bump_noise=False)
        monkeypatch.delenv("FDYN_NO_TRUNK_HEADS")
        ad, vd, _, _ = p.step(obs, st, start, deterministic=True, bump_noise=False)
    assert (a1 - a0).abs().max() < 2e-5 and (v1 - v0).abs().max() < 2e-5 and torch.equal(lp1, lp0)
    assert (a1 - ad).abs().max() > 0.1 and (a1.std(0) > 0.3).all()          # really sampled, really different per row
    assert torch.equal(vd, v1)


def test_episode_flags_moves_the_noise_counter_for_any_flag_dtype():
    """The glue between an env step and the next policy step advances the action-noise counter whatever dtype the env's done
    flags have (uint8: one fused launch; bool: the tensor-op fallback) -- a counter that stops would replay the same noise on
    every step of every rollout with no error raised."""
    from hcrl_amd.fused import episode_flags
    n = 1000
    for dt in (torch.uint8, torch.bool):
        term = (torch.rand(n, device="cuda") < 0.1).to(dt)
        trunc = (torch.rand(n, device="cuda") < 0.1).to(dt)
        start, keep = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        counter = torch.zeros(1, dtype=torch.int32, device="cuda")
        for k in range(3):
            episode_flags(term, trunc, start, keep, counter)
        done = (term.bool() | trunc.bool()).float()
        assert int(counter) == 3 and torch.equal(start, done) and torch.equal(keep, 1.0 - done)


def test_event_capacity_is_whole_shards_or_refused():
    """Episode-end records live in FD_EV_SHARDS segments of ev_cap / FD_EV_SHARDS records: the C entry refuses a capacity that
    would give segments of zero records (it used to drop every record silently), the host class rounds a caller's figure up."""
    from hcrl_amd import _lib, layout as L
    from hcrl_amd.rate_env import GpuRateVecEnv
    env = GpuRateVecEnv(10, "medium", 10.0, 0.02, "step", seed=0, precision="mixed", event_capacity=10)
    assert env.ev_cap % L.FD_EV_SHARDS == 0 and env.ev_cap >= L.FD_EV_SHARDS and env._ev_cap_shard >= 1
    env.reset()
    a = torch.zeros((10, 4), device=env.device)
    a[:, 0] = 1.0                                           # hard-over aileron: every env crashes within the episode
    ends = 0
    for _ in range(400):
        env.step_device(a)
        ints, _ = env.episode_events()
        ends += ints.shape[0]
    assert ends >= 10                                       # records were kept, not dropped
    lib = _lib.load()
    fn = lib.fdyn_rate_env_step_mixed
    cur, nxt = env._ev_counts[0], env._ev_counts[1]
    args = [env.x.data_ptr(), env.e.data_ptr(), env.ei.data_ptr(), None, env.params.data_ptr(), env.n_types, env.env_consts.data_ptr(),
            a.data_ptr(), env.pid_state.data_ptr(), env.pid_cfg.data_ptr(), env.casc_consts.data_ptr(), None, None, None, 1, 0, 1, 0.0,
            env.obs.data_ptr(), env.rewards.data_ptr(), env.rewards_full.data_ptr(), env.terminated.data_ptr(), env.truncated.data_ptr(),
            cur.data_ptr(), nxt.data_ptr(), env.ev_int.data_ptr(), env.ev_flt.data_ptr(), 10, env.n, _lib.current_stream()]
    assert fn(*args) == _lib.FDYN_ERR_BAD_SIZE


@pytest.mark.parametrize("B", [512, 65536])
def test_done_flags_inside_the_policy_step_equal_the_glue_launch(B):
    """policy.step(done_flags=(terminated, truncated)) -- the previous env step's flags turned into episode_start / keep and the
    noise counter moved on by the step's FIRST kernel (fdyn_policy_features_flags; a launch of fused.episode_flags where that
    kernel does not serve the batch) -- against fused.episode_flags followed by a plain step: same masks, same counter, same
    actions (same noise), same new state."""
    from hcrl_amd.fused import episode_flags
    from hcrl_amd.policy import RNNStates
    torch.manual_seed(5)
    p = RateLSTMPolicy(compute_dtype=torch.bfloat16).cuda()
    with torch.no_grad():
        p.action_net.weight.mul_(30.0)
    p.prepare_inference()
    obs = torch.randn(B, 18, device="cuda")
    mk = lambda: RNNStates(*(torch.randn(B, 256, device="cuda").to(dt) * 0.5 for dt in (torch.bfloat16, torch.float32) * 2))  # noqa: E731
    st = mk()
    term = (torch.rand(B, device="cuda") < 0.05).to(torch.uint8)
    trunc = (torch.rand(B, device="cuda") < 0.05).to(torch.uint8)
    ctr = p.noise_counter("cuda")
    ctr.fill_(7)
    es0, kp0 = torch.full((B,), 0.5, device="cuda"), torch.full((B,), 0.5, device="cuda")       # garbage: must be overwritten
    with torch.no_grad():
        a1, v1, lp1, s1 = p.step(obs, st, es0, keep=kp0, bump_noise=False, done_flags=(term, trunc))
    c1 = int(ctr)
    ctr.fill_(7)
    es1, kp1 = torch.zeros(B, device="cuda"), torch.zeros(B, device="cuda")
    episode_flags(term, trunc, es1, kp1, ctr)
    with torch.no_grad():
        a0, v0, lp0, s0 = p.step(obs, st, es1, keep=kp1, bump_noise=False)
    done = ((term | trunc) != 0).float()
    assert c1 == 8 and int(ctr) == 8 and torch.equal(es0, done) and torch.equal(kp0, 1.0 - done) and torch.equal(es0, es1)
    assert torch.equal(a1, a0) and torch.equal(v1, v0) and torch.equal(lp1, lp0) and all(torch.equal(x, y) for x, y in zip(s1, s0))
