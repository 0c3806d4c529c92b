"""GPU: the policy / PPO / BC glue on the real device-resident env."""
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


def test_timeout_bootstrap_path():
    env = GpuRateVecEnv(256, "easy", 0.2, 0.02, "step", seed=2)      # 10-step episodes: every rollout sees truncations
    m = RecurrentPPO(env, RateLSTMPolicy(use_lstm=False), PPOConfig(n_steps=12, n_epochs=1, n_minibatches=2,
                                                                     bootstrap_timeouts=True))
    m.learn(256 * 12, log_interval=0)
    assert np.isfinite(m.last_stats["value_loss"])
