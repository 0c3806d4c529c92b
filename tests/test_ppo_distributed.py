"""CPU tests of the policy / PPO host logic, including the N>1 path on gloo with world_size 2.
The env here is a tiny torch test double (the real env is HIP-only and has no CPU path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hcrl_amd
from hcrl_amd.policy import RateLSTMPolicy, RNNStates
from hcrl_amd.ppo import FlatGrad, PPOConfig, RecurrentPPO, compute_gae


class DummyVecEnv:
    """Test double with the GpuRateVecEnv device API: linear dynamics, truncation every 20 steps."""

    def __init__(self, n, seed):
        self.num_envs, self.device = n, torch.device("cpu")
        self.g = torch.Generator().manual_seed(seed)
        self.k = torch.zeros(n, dtype=torch.int64)

    def reset(self):
        self.obs = torch.randn(self.num_envs, 18, generator=self.g) * 0.1
        self.k.zero_()
        return self.obs

    def step_device(self, actions, auto_reset=True):
        a = actions.clamp(-1, 1)
        self.obs = 0.9 * self.obs
        self.obs[:, :4] += 0.1 * a
        self.k += 1
        rew = -self.obs[:, :3].abs().sum(1)
        trunc = (self.k >= 20)
        fresh = torch.randn(self.num_envs, 18, generator=self.g) * 0.1
        self.obs = torch.where(trunc[:, None], fresh, self.obs)
        self.k[trunc] = 0
        return self.obs, rew, torch.zeros_like(trunc, dtype=torch.uint8), trunc.to(torch.uint8)


def test_parameter_count_matches_survey():
    assert RateLSTMPolicy().num_parameters() == 1_830_089          # ~1.83 M
    assert RateLSTMPolicy(use_lstm=False).num_parameters() < 200_000


def test_zero_state_extractor_equals_nn_lstm():
    torch.manual_seed(0)
    fe = RateLSTMPolicy().features_extractor
    obs = torch.randn(7, 18)
    x = fe.embedding(obs)
    out, _ = fe.lstm(x.unsqueeze(1))                               # what lstm_policy.py:75-92 computes
    assert torch.allclose(fe.output_proj(out.squeeze(1)), fe(obs), atol=1e-6)


def test_sequence_evaluation_equals_stepwise_rollout():
    torch.manual_seed(1)
    p = RateLSTMPolicy()
    T, B = 6, 5
    obs, starts = torch.randn(T, B, 18), torch.zeros(T, B)
    starts[0] = 1; starts[3, 2] = 1
    s = p.initial_state(B)
    acts, vals, lps = [], [], []
    for t in range(T):
        a, v, lp, s = p.step(obs[t], s, starts[t])
        acts.append(a); vals.append(v); lps.append(lp)
    v2, lp2, _ = p.evaluate_sequence(obs, torch.stack(acts), starts, p.initial_state(B))
    assert torch.allclose(torch.stack(vals), v2, atol=1e-6) and torch.allclose(torch.stack(lps), lp2, atol=1e-5)


def test_gae_against_per_env_loop():
    torch.manual_seed(2)
    T, N, g, lam = 9, 4, 0.99, 0.95
    rew, val = torch.randn(T, N), torch.randn(T, N)
    starts = (torch.rand(T, N) < 0.2).float()
    last_v, last_d = torch.randn(N), (torch.rand(N) < 0.5).float()
    adv, ret = compute_gae(rew, val, starts, last_v, last_d, g, lam)
    for n in range(N):
        a = 0.0
        for t in reversed(range(T)):
            nonterm = 1 - (last_d[n] if t == T - 1 else starts[t + 1, n])
            nv = last_v[n] if t == T - 1 else val[t + 1, n]
            delta = rew[t, n] + g * nv * nonterm - val[t, n]
            a = delta + g * lam * nonterm * a
            assert abs(float(adv[t, n]) - float(a)) < 1e-5
    assert torch.allclose(ret, adv + val)


def test_flat_grad_is_one_buffer():
    p = RateLSTMPolicy(use_lstm=False)
    fg = FlatGrad(p)
    assert fg.buf.numel() == p.num_parameters()
    loss = sum((q ** 2).sum() for q in p.parameters())
    loss.backward()
    assert torch.allclose(fg.buf, torch.cat([2 * q.detach().view(-1) for q in p.parameters()]))
    n = fg.clip_norm_(0.5)
    assert abs(float(fg.buf.norm()) - 0.5) < 1e-4 and n > 0.5


def test_single_process_learn_improves_dummy_task():
    env = DummyVecEnv(64, 0)
    m = RecurrentPPO(env, RateLSTMPolicy(use_lstm=False), PPOConfig(n_steps=16, n_epochs=2, n_minibatches=2), seed=0)
    m.learn(64 * 16 * 3, log_interval=0)
    assert m.num_timesteps == 64 * 16 * 3 and np.isfinite(m.last_stats["policy_loss"])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        env = DummyVecEnv(32, 100 + rank)                        # each rank owns a different env shard
        pol = RateLSTMPolicy(features_dim=32, lstm_hidden_size=32, policy_lstm_hidden=32, net_arch_pi=(16,), net_arch_vf=(16,))
        m = RecurrentPPO(env, pol, PPOConfig(n_steps=8, n_epochs=2, n_minibatches=2), seed=3)
        p0 = torch.cat([p.detach().view(-1) for p in m.policy.parameters()]).clone()
        m.learn(2 * 32 * 8 * 2, log_interval=0)                  # total over both ranks: 2 iterations each
        flat = torch.cat([p.detach().view(-1) for p in m.policy.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        obs_sum = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(obs_sum, m.buf_obs.sum().view(1))
        q.put((rank, bool(torch.equal(gathered[0], gathered[1])), float((flat - p0).abs().max()),
               float(obs_sum[0]), float(obs_sum[1]), m.num_timesteps))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_all_reduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=240) for _ in range(2)]
    [p.join(60) for p in procs]
    for rank, same, moved, o0, o1, steps in res:
        assert same, "replicas diverged: the gradient all-reduce is not keeping ranks in lock-step"
        assert moved > 0, "parameters never moved"
        assert o0 != o1, "ranks must roll out different env shards"
        assert steps == 2 * 32 * 8
