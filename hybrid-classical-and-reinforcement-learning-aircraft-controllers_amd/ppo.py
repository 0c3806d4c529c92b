"""Recurrent PPO over the device-resident vec-env, data-parallel across GPUs with one flat gradient all-reduce.

What it stands in for: the `RecurrentPPO(...)`/`PPO(...)` + `model.learn(...)` calls of the reference's trainers
(learned_controllers/train_rate.py:128-180, train_overnight.py:142-224) with the hyper-parameters of
learned_controllers/config/ppo_lstm.yaml:38-58 (lr 3e-4, gamma 0.99, lambda 0.95, clip 0.2, ent 0.01, vf 0.5,
max-grad-norm 0.5, n_epochs 10).  SB3 itself is third-party and absent; this is an independent implementation of the
published PPO-clip + GAE algorithm (parity with SB3 is unpinned).

MI355X-first choices:
  * rollouts never leave the GPU: env step = one fused HIP launch, policy step = a handful of MFMA GEMMs, the rollout
    buffer is a preallocated [T, N, ...] block in HBM (65 536 envs x 64 steps x 18 f32 = 302 MB -- trivial in 288 GB);
  * sequences are whole rollouts per env, minibatches are env slices (BPTT over T with episode-start masking);
  * gradients live in ONE flat fp32 buffer (parameters' .grad are views into it), so data parallelism is a single
    `all_reduce` of ~7.3 MB per optimizer step on RCCL/xGMI -- latency-bound, hence one call instead of per-tensor
    buckets; envs never communicate.
"""
import os
import time
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from .fused import episode_flags, ppo_loss
from .policy import RateLSTMPolicy, RNNStates


@dataclass
class PPOConfig:
    learning_rate: float = 3.0e-4
    n_steps: int = 64            # per-env rollout length (reference: 2048 with 4 envs; here N is 10^4..10^5)
    n_minibatches: int = 4       # env slices per epoch (reference expresses this as batch_size); big slices keep the
                                 # per-timestep BPTT GEMMs wide enough to fill 256 CUs (8 -> 2 slices: 235 -> 141 ms/iter)
    n_epochs: int = 10
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    clip_range_vf: Optional[float] = None
    ent_coef: float = 0.01
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    normalize_advantage: bool = True
    bootstrap_timeouts: object = False   # False | True (exact: V(terminal obs), host sync per step, no rollout graph) |
                                         # "approx" (device-side, graph-safe: gamma * V(s_t) of the step that timed out)
    reward_scale: float = 1.0    # rewards are multiplied by this before GAE (1.0 = the reference's raw rewards; its -100 crash
                                 # penalty makes value targets O(100), so the shared grad-norm clip starves the actor at large batch)

    @classmethod
    def from_dict(cls, d: dict, **over):
        keys = {"learning_rate", "n_steps", "n_epochs", "gamma", "gae_lambda", "clip_range", "clip_range_vf",
                "ent_coef", "vf_coef", "max_grad_norm", "n_minibatches", "bootstrap_timeouts", "reward_scale"}
        kw = {k: v for k, v in d.items() if k in keys}
        kw.update(over)
        return cls(**kw)


class FlatGrad:
    """All parameter gradients as views of one contiguous buffer => one collective per optimizer step."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.buf = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.events = None           # a list => every collective is bracketed by HIP events on the stream that waits for it
        off = 0
        for p in self.params:
            p.grad = self.buf[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.buf.zero_()

    def all_reduce_mean(self):
        # HCRL_FORCE_COLLECTIVE: a one-rank group still issues the call (bench.py --single-rank-group: the only rehearsal of
        # the RCCL path a one-GPU box allows)
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("HCRL_FORCE_COLLECTIVE") == "1"):
            timed = self.events is not None and self.buf.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM)
            if timed:
                e1.record()
                self.events.append((e0, e1))
            self.buf.div_(dist.get_world_size())

    def clip_norm_(self, max_norm: float):
        norm = self.buf.norm()
        self.buf.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
        return norm


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def compute_gae(rewards, values, episode_starts, last_values, last_dones, gamma, lam):
    """rewards/values/episode_starts [T,N]; episode_starts[t] = 1 if env was reset before step t.  Returns adv, returns."""
    if rewards.is_cuda:
        from .fused import gae
        return gae(rewards, values, episode_starts, last_values, last_dones, gamma, lam)
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in range(T - 1, -1, -1):
        if t == T - 1:
            nonterminal, next_v = 1.0 - last_dones, last_values
        else:
            nonterminal, next_v = 1.0 - episode_starts[t + 1], values[t + 1]
        delta = rewards[t] + gamma * next_v * nonterminal - values[t]
        last = delta + gamma * lam * nonterminal * last
        adv[t] = last
    return adv, adv + values


@torch.no_grad()
def episode_stats_from_rollout(rew, starts, final_start, run_ret, run_len):
    """Monitor-style bookkeeping over one rollout without a per-step host loop: `rew` [T, N] per-step rewards, `starts`
    [T, N] episode-start flags of the observations (starts[t] = done after step t-1), `final_start` [N] = done after the
    last step, `run_ret` / `run_len` [N] the return / length each env had accumulated before the rollout.
    -> (sum of returns, sum of lengths, count) of the episodes that ended inside the rollout as one [3] tensor, plus the
    new carries.  (What SB3's Monitor wrapper + `ep_info_buffer` feed `rollout/ep_rew_mean`, `rollout/ep_len_mean`.)"""
    T, N = rew.shape
    done = torch.cat([starts[1:], final_start[None]], 0) > 0
    cs = rew.double().cumsum(0)
    t = torch.arange(T, device=rew.device)[:, None].expand(T, N)
    last = torch.where(done, t, torch.full_like(t, -1)).cummax(0).values          # last step <= t that ended an episode
    prev = torch.cat([torch.full((1, N), -1, dtype=last.dtype, device=rew.device), last[:-1]], 0)   # ... strictly before t
    has = prev >= 0
    zero = torch.zeros((), dtype=cs.dtype, device=rew.device)
    ep_ret = cs - torch.where(has, cs.gather(0, prev.clamp(min=0)), zero) + torch.where(has, zero, run_ret.double()[None])
    ep_len = (t - prev).double() + torch.where(has, zero, run_len.double()[None])
    sums = torch.stack([(ep_ret * done).sum(), (ep_len * done).sum(), done.sum().double()])
    fl = last[-1]
    ended = fl >= 0
    new_ret = torch.where(ended, cs[-1] - cs.gather(0, fl.clamp(min=0)[None])[0], cs[-1] + run_ret.double())
    new_len = torch.where(ended, (T - 1 - fl).double(), T + run_len.double())
    return sums, new_ret, new_len


class RecurrentPPO:
    def __init__(self, env, policy: Optional[RateLSTMPolicy] = None, config: Optional[PPOConfig] = None, seed: int = 0,
                 use_graph: bool = True, use_update_graph: Optional[bool] = None):
        self.env, self.cfg = env, config or PPOConfig()
        self.use_graph, self._graph, self._graph_env = use_graph, None, None
        self.use_update_graph = use_graph if use_update_graph is None else use_update_graph
        self.device = env.device
        torch.manual_seed(seed)                          # identical initial weights on every rank
        self.policy = (policy or RateLSTMPolicy()).to(self.device)
        broadcast_parameters(self.policy)
        self.flat = FlatGrad(self.policy)
        # one fused multi-tensor kernel per optimizer step on the GPU (the foreach form is a dozen launches over 33 tensors)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=self.cfg.learning_rate, eps=1e-5,
                                    **({"fused": True} if self.device.type == "cuda" else {}))
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        torch.manual_seed(seed + 7919 * (rank + 1))      # different exploration noise per rank
        N, T = env.num_envs, self.cfg.n_steps
        if N % max(1, self.cfg.n_minibatches) != 0:
            # env slices are fixed-size (one captured update graph serves every slice): a remainder would silently drop
            # envs from every epoch
            raise ValueError(f"num_envs ({N}) must be a multiple of n_minibatches ({self.cfg.n_minibatches})")
        f32 = dict(dtype=torch.float32, device=self.device)
        self.buf_obs = torch.zeros((T, N, 18), **f32)
        self.buf_act = torch.zeros((T, N, 4), **f32)
        self.buf_rew, self.buf_val, self.buf_logp, self.buf_start = (torch.zeros((T, N), **f32) for _ in range(4))
        # time-limit bootstrap term gamma * V(.), kept OUT of buf_rew: GAE sees rew + boot, the logs see the env's own rewards
        self.buf_boot = torch.zeros((T, N), **f32) if self.cfg.bootstrap_timeouts else None
        self.obs = env.reset()                           # aliases the env's observation buffer (rewritten every step)
        if hasattr(env, "mark_episode_starts"):
            env.mark_episode_starts()
        self._alloc_states(N)
        self.episode_start = torch.ones(N, **f32)
        self.keep = torch.zeros(N, **f32)                # 1 - episode_start, kept in step by fused.episode_flags
        self.num_timesteps = 0
        self.ep_returns, self.ep_lengths = [], []
        self.last_stats = {}
        # Monitor-style episode statistics (rollout/ep_rew_mean, ep_len_mean): off unless a logger asks -- learn() turns
        # it on when it is given a callback; costs a dozen small launches and one host read per iteration
        self.track_episode_stats = False
        self._run_ret = torch.zeros(N, dtype=torch.float64, device=self.device)
        self._run_len = torch.zeros(N, dtype=torch.float64, device=self.device)
        self._ep_stats = {}

    def _alloc_states(self, n):
        """Fixed recurrent-state buffers: h in the policy's compute dtype (what the MFMA cell emits), c in fp32."""
        pol = self.policy
        hd = pol.compute_dtype if (pol.compute_dtype is not None and self.device.type == "cuda") else torch.float32
        z = lambda dt: torch.zeros(n, pol.hidden, dtype=dt, device=self.device)  # noqa: E731
        mk = lambda: RNNStates(z(hd), z(torch.float32), z(hd), z(torch.float32))  # noqa: E731
        # ping-pong: step t reads one set and writes the other -- or, where the cell kernel owns whole rows per wave, ONE set
        # updated in place (both entries are the same buffers)
        first = mk()
        self._state_bufs, self._cur = [first, first if pol.recurrent_inplace_ok(n, self.device) else mk()], 0
        self.rollout_states = mk()

    @property
    def states(self) -> RNNStates:
        return self._state_bufs[self._cur]

    def set_env(self, env):
        """Curriculum phases swap the env (train_rate.py:155-170); recurrent state and episode flags restart."""
        assert env.num_envs == self.env.num_envs
        self.env = env
        self.obs = env.reset()
        if hasattr(env, "mark_episode_starts"):
            env.mark_episode_starts()
        for t in self.states:
            t.zero_()
        self.episode_start.fill_(1.0)
        self.keep.zero_()

    def _rollout_body(self):
        """T policy+env steps, device ops only (no host sync): runs eagerly or under hipGraph capture.  All state lives in
        fixed buffers (self.obs / self.states / self.episode_start are updated IN PLACE) so a captured graph can be replayed."""
        env, pol, cfg = self.env, self.policy, self.cfg
        for dst, src in zip(self.rollout_states, self.states):
            dst.copy_(src)
        # the fused policy path + uint8 env flags: episode_start, keep and the action-noise counter move in ONE launch per step
        fused_glue = self.device.type == "cuda" and pol._fused_ok(self.obs)
        counter = pol.noise_counter(self.device) if fused_glue else None
        # where the env hands out uint8 done flags in fixed buffers, the step's FIRST kernel turns the previous step's flags into
        # episode_start / keep and moves the noise counter on (policy.step: done_flags): no glue launch between the steps
        in_step = fused_glue and all(getattr(getattr(env, k, None), "dtype", None) == torch.uint8 for k in ("terminated", "truncated"))
        for t in range(cfg.n_steps):
            nxt = self._state_bufs[1 - self._cur]             # the fused cells write the new state straight into it
            actions, values, logp, new_states = pol.step(self.obs, self.states, self.episode_start, out_states=nxt,
                                                         keep=self.keep, bump_noise=not fused_glue,
                                                         done_flags=(env.terminated, env.truncated) if in_step else None)
            self.buf_obs[t].copy_(self.obs); self.buf_act[t].copy_(actions); self.buf_val[t].copy_(values)
            self.buf_logp[t].copy_(logp); self.buf_start[t].copy_(self.episode_start)
            obs, rew, term, trunc = env.step_device(actions)          # clip happens in-kernel (rate_env.py:225)
            self.buf_rew[t].copy_(rew)
            if cfg.reward_scale != 1.0:
                self.buf_rew[t].mul_(cfg.reward_scale)
            if cfg.bootstrap_timeouts == "approx":
                # time-limit truncation is not failure.  Partial-episode bootstrap without a second critic pass: the value the
                # critic gave the observation this step started from stands in for V(terminal observation) (one 20 ms step
                # earlier), as large-batch GPU trainers do; SB3 (which the reference uses) evaluates the terminal observation
                # itself -- that is bootstrap_timeouts=True below, at the price of a host sync per step.
                timeout = (trunc & ~term).float() if trunc.dtype == torch.bool else ((trunc != 0) & (term == 0)).float()
                torch.mul(values, timeout, out=self.buf_boot[t])
                self.buf_boot[t].mul_(cfg.gamma)
            elif cfg.bootstrap_timeouts:
                # time-limit truncation is not failure: add gamma * V(s_T) (vec-env 'TimeLimit.truncated' handling).
                # The post-step critic state belongs to the finished episode, so it can value the terminal observation.
                ints, flts = env.episode_events()                     # host sync: this option disables graph replay
                self.buf_boot[t].zero_()
                if ints.shape[0]:
                    tr = ints[:, 2] == 0
                    if bool(tr.any()):
                        ids = ints[tr, 0].long()
                        v = pol.predict_values(flts[tr, 1:], new_states.index(ids), torch.zeros(ids.numel(), device=self.device))
                        self.buf_boot[t, ids] = cfg.gamma * v
            if obs.data_ptr() != self.obs.data_ptr():
                self.obs.copy_(obs)                               # (self.obs aliases the env's observation buffer on the GPU)
            for dst, src in zip(nxt, new_states):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)                                # un-fused policy paths return fresh tensors
            self._cur = 1 - self._cur
            if in_step:
                if t == cfg.n_steps - 1:                          # after the last step: the flags GAE and predict_values read
                    episode_flags(term, trunc, self.episode_start, self.keep, None)
            elif fused_glue:                                      # any flag dtype: its tensor-op fallback bumps the counter too
                episode_flags(term, trunc, self.episode_start, self.keep, counter)
            else:
                self.episode_start.copy_((term | trunc).float())
                self.keep.copy_(1.0 - self.episode_start)

    @torch.no_grad()
    def collect_rollout(self):
        pol, cfg = self.policy, self.cfg
        pol.prepare_inference()                           # bf16 weight snapshot (refreshed IN PLACE) for the fused MFMA path
        if self.use_graph and cfg.bootstrap_timeouts in (False, "approx") and self.device.type == "cuda" and cfg.n_steps % 2 == 0:
            if self._graph is None or self._graph_env is not self.env:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._rollout_body()                  # warm-up on a side stream (allocator, lazy init) -- a real rollout
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                    self._rollout_body()
                self._graph_env = self.env
            self._graph.replay()                          # the whole T-step rollout is ONE graph launch
        else:
            self._rollout_body()
        self.num_timesteps += cfg.n_steps * self.env.num_envs
        last_values = pol.predict_values(self.obs, self.states, self.episode_start)
        rew = self.buf_rew if self.buf_boot is None else self.buf_rew + self.buf_boot
        self.adv, self.ret = compute_gae(rew, self.buf_val, self.buf_start, last_values, self.episode_start,
                                         cfg.gamma, cfg.gae_lambda)

    def _minibatch_loss(self, obs, act, starts, adv, ret, old_logp, old_v, states):
        """Clipped-surrogate PPO loss of one env slice ([T, mb] tensors) + the detached statistics."""
        cfg, pol = self.cfg, self.policy
        values, mean = pol.sequence_heads(obs, starts, states)
        return ppo_loss(mean, values, pol.log_std, act, old_logp, adv, ret, old_v, cfg.normalize_advantage, cfg.clip_range,
                        cfg.clip_range_vf, cfg.vf_coef, cfg.ent_coef)

    def _build_update_graph(self, mb):
        """Forward + loss + backward of one env slice as ONE hipGraph over static input buffers: the ~1500 launches of a
        BPTT pass stop being paced by the host (at 16 384 envs the eager loop is launch-bound).  Gradients land in the
        flat buffer; the collective, the clip and the optimizer step stay outside the graph."""
        T, dev = self.cfg.n_steps, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        g = {"obs": torch.zeros((T, mb, 18), **f32), "act": torch.zeros((T, mb, 4), **f32)}
        for k in ("starts", "adv", "ret", "old_logp", "old_v"):
            g[k] = torch.zeros((T, mb), **f32)
        g["states"] = RNNStates(*[torch.zeros((mb, t.shape[1]), dtype=t.dtype, device=dev) for t in self.rollout_states])

        def body():
            self.flat.zero()
            loss, st = self._minibatch_loss(g["obs"], g["act"], g["starts"], g["adv"], g["ret"], g["old_logp"], g["old_v"], g["states"])
            loss.backward()
            return st

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # warm-up: allocator, library handles, lazy one-time choices
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            g["stats"] = body()
        g["graph"], g["mb"] = graph, mb
        return g

    def update(self):
        cfg, pol = self.cfg, self.policy
        N = self.env.num_envs
        mb = max(1, N // cfg.n_minibatches)
        stats = dict(policy_loss=0.0, value_loss=0.0, approx_kl=0.0, clip_frac=0.0, grad_norm=0.0, n=0)
        ug = None
        if self.use_update_graph and self.device.type == "cuda" and (pol.sequence_bptt or not pol.use_lstm):
            ug = getattr(self, "_update_graph", None)
            if ug is None or ug["mb"] != mb:
                try:
                    ug = self._update_graph = self._build_update_graph(mb)
                except Exception as e:                    # capture is an optimisation: fall back to the eager loop, loudly
                    import sys
                    print(f"[ppo] update-graph capture failed ({type(e).__name__}: {e}); running the update eagerly",
                          file=sys.stderr, flush=True)
                    self.use_update_graph, ug = False, None
                    torch.cuda.synchronize()
        for _ in range(cfg.n_epochs):
            perm = torch.randperm(N, device=self.device)
            for s in range(0, N - mb + 1, mb):
                idx = perm[s:s + mb]
                if ug is not None:
                    for k, src in (("obs", self.buf_obs), ("act", self.buf_act), ("starts", self.buf_start), ("adv", self.adv),
                                   ("ret", self.ret), ("old_logp", self.buf_logp), ("old_v", self.buf_val)):
                        torch.index_select(src, 1, idx, out=ug[k])
                    for dst, src in zip(ug["states"], self.rollout_states):
                        torch.index_select(src, 0, idx, out=dst)
                    ug["graph"].replay()
                    st = ug["stats"]
                else:
                    loss, st = self._minibatch_loss(self.buf_obs[:, idx], self.buf_act[:, idx], self.buf_start[:, idx],
                                                    self.adv[:, idx], self.ret[:, idx], self.buf_logp[:, idx],
                                                    self.buf_val[:, idx], self.rollout_states.index(idx))
                    self.flat.zero()
                    loss.backward()
                self.flat.all_reduce_mean()                    # THE collective: one flat ~7.3 MB all-reduce
                gn = self.flat.clip_norm_(cfg.max_grad_norm)
                self.opt.step()
                with torch.no_grad():
                    stats["policy_loss"] += st[0]; stats["value_loss"] += st[1]
                    stats["approx_kl"] += st[2]; stats["clip_frac"] += st[3]
                    stats["grad_norm"] += gn; stats["n"] += 1
        n = max(stats.pop("n"), 1)
        self.last_stats = {k: float(v.detach() if torch.is_tensor(v) else v) / n for k, v in stats.items()}
        self.last_stats["mean_reward_per_step"] = float(self.buf_rew.mean()) / cfg.reward_scale
        if self.track_episode_stats:
            sums, self._run_ret, self._run_len = episode_stats_from_rollout(self.buf_rew / cfg.reward_scale, self.buf_start,
                                                                            self.episode_start, self._run_ret, self._run_len)
            ret_sum, len_sum, count = sums.tolist()
            if count > 0:                                 # no episode ended in this rollout: keep the previous means
                self._ep_stats = {"ep_rew_mean": ret_sum / count, "ep_len_mean": len_sum / count}
            self.last_stats.update(self._ep_stats, episodes=count)
        return self.last_stats

    def learn(self, total_timesteps: int, log_interval: int = 1, callback=None):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        target = self.num_timesteps + total_timesteps // world
        it, t0 = 0, time.time()
        if callback is not None:
            self.track_episode_stats = True
        while self.num_timesteps < target:
            self.collect_rollout()
            st = self.update()
            it += 1
            if callback is not None:
                callback(self, st)
            if log_interval and it % log_interval == 0 and (not dist.is_initialized() or dist.get_rank() == 0):
                fps = self.cfg.n_steps * self.env.num_envs * world * it / max(time.time() - t0, 1e-9)
                print(f"[ppo] iter {it} steps {self.num_timesteps * world} fps {fps:,.0f} "
                      f"rew/step {st['mean_reward_per_step']:.3f} pl {st['policy_loss']:.4f} vl {st['value_loss']:.3f} "
                      f"kl {st['approx_kl']:.4f} clip {st['clip_frac']:.3f} gnorm {st['grad_norm']:.2f} "
                      f"std {float(self.policy.log_std.detach().exp().mean()):.3f}", flush=True)
        return self

    def save(self, path):
        torch.save({"policy": self.policy.state_dict(), "optimizer": self.opt.state_dict(),
                    "num_timesteps": self.num_timesteps, "config": self.cfg.__dict__}, path)

    def save_sb3_zip(self, path):
        """Policy weights + plain hyper-parameters in the Stable-Baselines3 archive layout (what the reference's
        `model.save(...)` writes, train_rate.py:353-355; see sb3_zip.py for what is and is not in the file)."""
        from .sb3_zip import save_sb3_zip
        return save_sb3_zip(path, self.policy, hyper=dict(self.cfg.__dict__), num_timesteps=self.num_timesteps)

    def load(self, path):
        from .sb3_zip import is_sb3_zip, read_sb3_zip
        if is_sb3_zip(path):                    # weights only: an SB3 archive's optimizer state is in SB3's parameter order
            sd, meta = read_sb3_zip(path, self.device)
            self.policy.load_state_dict(sd)
            self.num_timesteps = int(meta["data"].get("num_timesteps", 0) or 0)
            return self
        ck = torch.load(path, map_location=self.device, weights_only=True)
        self.policy.load_state_dict(ck["policy"]); self.opt.load_state_dict(ck["optimizer"])
        self.num_timesteps = ck["num_timesteps"]
