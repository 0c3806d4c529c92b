"""Device-resident rate-control environments.

`GpuRateVecEnv` is N `RateControlEnv`s (learned_controllers/envs/rate_env.py:17-470) stepped by ONE fused HIP launch
per vec-env step: action clip -> 20 RK4 sub-steps -> command update -> reward (+ settling bonus, crash penalty) ->
termination -> observation -> episode-end compaction -> in-kernel auto-reset.  It presents the vec-env surface the
reference's trainer drives (`num_envs`, `reset()`, `step(actions)`, `step_async/step_wait`, `close`, `seed`,
`get_attr/set_attr/env_method`; learned_controllers/utils/training_utils.py:49-69) but hands back device tensors, so
a policy on the same GPU never crosses PCIe.  `numpy_io=True` returns NumPy like the reference's vec-env does.

Episode sampling: `parity` mode replays NumPy MT19937 streams pre-sampled on the host in the reference's exact call
order (samplers.presample_reset_pool; identical seeds => identical episodes); `device` mode draws in-kernel (Philox).
"""
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib, layout as L
from .config import cascade_consts, pid_table
from .params import param_table
from .samplers import COMMAND_TYPE, EpisodeStreams, env_consts, presample_reset_pool


class GpuRateVecEnv:
    def __init__(self, num_envs: int, difficulty: str = "medium", episode_length: float = 10.0, dt: float = 0.02,
                 command_type: str = "step", seed: Optional[int] = None, precision: str = "mixed",
                 sampling: str = "device", pool_depth: int = 8, types: Sequence = ("rc_plane",),
                 type_index: Optional[np.ndarray] = None, event_capacity: Optional[int] = None,
                 numpy_io: bool = False, device=None, residual_scale: float = 0.0, sensor_noise: Optional[dict] = None):
        self.lib = _lib.load()
        self.device = device or _lib.require_gpu()
        self.num_envs = self.n = int(num_envs)
        self.precision, self.dtype = precision, _lib.state_dtype(precision)
        self.difficulty, self.episode_length, self.dt, self.command_type = difficulty, episode_length, dt, command_type
        self.seed_value = 0 if seed is None else int(seed)
        self.numpy_io = numpy_io
        # > 0: ResidualRateControlEnv semantics (residual_rate_env.py:17-182): actions are corrections added to the fused PID
        self.residual_scale = float(residual_scale)
        dev, n = self.device, self.n
        self.env_consts_host = env_consts(difficulty, episode_length, dt, command_type)
        self.env_consts = torch.as_tensor(self.env_consts_host, device=dev)
        self.params = torch.as_tensor(param_table(types), device=dev).contiguous()
        self.n_types = self.params.shape[0]
        self.type_index = None if type_index is None else torch.as_tensor(np.asarray(type_index, np.uint8), device=dev)
        self.x = torch.zeros((L.FD_NX, n), dtype=self.dtype, device=dev)
        # env words: fp32 in the fp32-evaluation variants (272 of the 660 B per env-step as fp64 rows); FD_E_SETTLE_TIMER then
        # counts settled steps and FD_E_TIME is step * dt (include/fdyn.h)
        self.e = torch.zeros((L.FD_NE, n), dtype=torch.float64 if precision == "f64" else torch.float32, device=dev)
        self.ei = torch.zeros((L.FD_NEI, n), dtype=torch.int32, device=dev)
        self.obs = torch.zeros((n, L.FD_OBS_DIM), dtype=torch.float32, device=dev)
        self.rewards = torch.zeros(n, dtype=torch.float32, device=dev)
        self.rewards_full = torch.zeros(n, dtype=self.dtype, device=dev)
        self.terminated = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.actions_taken = torch.zeros((n, L.FD_ACT_DIM), dtype=torch.float32, device=dev)
        # compacted episode-end records: FD_EV_SHARDS segments, one counter each (one fleet-wide atomic word saturates)
        full = int(self.lib.fdyn_event_capacity(n))
        if event_capacity is None:
            self.ev_cap = full
        else:                                    # a caller's figure: whole shards (the kernel divides it by FD_EV_SHARDS), at most `full`
            self.ev_cap = min(full, -(-max(int(event_capacity), 1) // L.FD_EV_SHARDS) * L.FD_EV_SHARDS)
        self._ev_cap_shard = self.ev_cap // L.FD_EV_SHARDS
        self._ev_counts = torch.zeros((2, L.FD_EV_SHARDS), dtype=torch.int32, device=dev)   # ping-pong sets (kernel clears the other)
        self._ev_slot = 0
        self._ev_cur = self._ev_counts[0]
        self.ev_int = torch.zeros((max(self.ev_cap, 1), L.FD_EV_NI), dtype=torch.int32, device=dev)
        self.ev_flt = torch.zeros((max(self.ev_cap, 1), L.FD_EV_NF), dtype=torch.float32, device=dev)
        # fused PID demonstrator (learned_controllers/utils/pid_demonstrations.py:41-77)
        self.pid_state = torch.zeros((3 * L.FD_NPS, n), dtype=torch.float32, device=dev)
        self.pid_cfg = torch.as_tensor(pid_table(), device=dev)
        self.casc_consts = torch.as_tensor(cascade_consts(), device=dev)
        # episode sampling
        self.sampling = sampling
        self.pool, self.pool_depth, self._streams = None, 0, None
        if sampling == "parity":
            seeds = [None if seed is None else seed + i for i in range(n)]       # make_env: rng_seed = seed + rank
            self.pool_depth = int(pool_depth)
            self.pool = torch.as_tensor(presample_reset_pool(seeds, pool_depth, difficulty, command_type), device=dev)
            if command_type == "random":                                          # per-step random-walk deltas
                self._streams = [EpisodeStreams(difficulty, command_type, s) for s in seeds]
        elif sampling != "device":
            raise ValueError("sampling must be 'parity' or 'device'")
        # optional sensor layer between the physics and the policy (interfaces/sensor.py:137-243 noise model on the obs)
        self.sensor = None
        if sensor_noise is not None:
            from .sensors import ObservationNoise
            self.sensor = ObservationNoise(sensor_noise, n, dev)
            self._done_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._reset_fn = getattr(self.lib, f"fdyn_rate_env_reset_{precision}")
        self._step_fn = getattr(self.lib, f"fdyn_rate_env_step_{precision}")
        self._pending = None

    # ---- vec-env surface ------------------------------------------------------------------------------------
    def reset(self, mask: Optional[torch.Tensor] = None):
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        rc = self._reset_fn(_lib.ptr(self.x), _lib.ptr(self.e), _lib.ptr(self.ei), _lib.ptr(self.pid_state),
                            _lib.ptr(m), _lib.ptr(self.env_consts), _lib.ptr(self.pool), self.pool_depth,
                            self.seed_value, _lib.ptr(self.obs), self.n, _lib.current_stream())
        _lib.check(rc, "RateControlEnv.reset")
        if self.sensor is not None:
            if m is None:
                self.sensor.reset()
                self.sensor.apply(self.obs)
            else:       # a masked reset restarts the bias walk of the reset envs only; obs rows of the others are unchanged
                keep, keep_bias, mb = self.obs.clone(), self.sensor.gyro_bias.clone(), m.bool()
                self.sensor.apply(self.obs, m)
                self.obs.copy_(torch.where(mb[:, None], self.obs, keep))
                self.sensor.gyro_bias.copy_(torch.where(mb[None, :], self.sensor.gyro_bias, keep_bias))
        return self._out(self.obs)

    def step_device(self, actions: Optional[torch.Tensor], auto_reset: bool = True, rw_delta=None):
        """One fused launch.  `actions` [N,4] fp32 on the device, or None => the fused rate-PID demonstrator."""
        if actions is not None:
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
            assert actions.shape == (self.n, L.FD_ACT_DIM)
        if rw_delta is not None:                               # random-walk command increments [3][N] in the state dtype
            assert rw_delta.shape == (3, self.n) and rw_delta.dtype == self.dtype and rw_delta.device == self.x.device
        # `actions_taken` is written only where the action is made in-kernel (fused PID demonstrator, residual env): with a
        # policy's own actions it would be a 16-B-per-env copy of the input
        acts_out = self.actions_taken if (actions is None or self.residual_scale > 0.0) else None
        cur, nxt = self._ev_counts[self._ev_slot], self._ev_counts[1 - self._ev_slot]
        self._ev_cur, self._ev_slot = cur, 1 - self._ev_slot
        rc = self._step_fn(_lib.ptr(self.x), _lib.ptr(self.e), _lib.ptr(self.ei), _lib.ptr(self.type_index),
                           _lib.ptr(self.params), self.n_types, _lib.ptr(self.env_consts), _lib.ptr(actions),
                           _lib.ptr(self.pid_state), _lib.ptr(self.pid_cfg), _lib.ptr(self.casc_consts),
                           _lib.ptr(acts_out), _lib.ptr(rw_delta), _lib.ptr(self.pool), self.pool_depth,
                           self.seed_value, int(auto_reset), self.residual_scale, _lib.ptr(self.obs), _lib.ptr(self.rewards),
                           _lib.ptr(self.rewards_full), _lib.ptr(self.terminated), _lib.ptr(self.truncated),
                           cur.data_ptr(), nxt.data_ptr(), _lib.ptr(self.ev_int), _lib.ptr(self.ev_flt), self.ev_cap,
                           self.n, _lib.current_stream())
        _lib.check(rc, "RateControlEnv.step")
        if self.sensor is not None:
            mask = None
            if auto_reset:                                     # obs rows of finished envs are first observations
                torch.bitwise_or(self.terminated, self.truncated, out=self._done_mask)
                mask = self._done_mask
            self.sensor.apply(self.obs, mask)
        return self.obs, self.rewards, self.terminated, self.truncated

    def mark_episode_starts(self):
        """After a reset every env is at the start of an episode: say so in the done flags a rollout loop reads as "the previous
        step ended an episode" (SB3's `_last_episode_starts = ones`); the next step overwrites them."""
        self.terminated.fill_(1)
        self.truncated.zero_()

    def step(self, actions, auto_reset: bool = True):
        if isinstance(actions, np.ndarray):
            if self.numpy_io and actions.shape == (self.n, L.FD_ACT_DIM):        # pinned staging: one async H2D, no pageable copy
                io = self._host_io()
                np.copyto(io["act_np"], actions, casting="same_kind")
                io["act_dev"].copy_(io["act"], non_blocking=True)
                actions = io["act_dev"]
            else:
                actions = torch.as_tensor(actions, dtype=torch.float32, device=self.device)
        rw = None
        if self._streams is not None:                                # parity-mode random-walk deltas for THIS step
            d = np.stack([s.random_walk_delta(self.dt) for s in self._streams], axis=1)
            rw = torch.as_tensor(np.ascontiguousarray(d), device=self.device).to(self.dtype)
        obs, rew, term, trunc = self.step_device(actions, auto_reset, rw)
        if self.numpy_io:
            # everything the host needs leaves in ONE stream-ordered batch of async copies into pinned buffers, one sync
            io = self._host_io()
            torch.bitwise_or(term, trunc, out=io["done_dev"])
            io["obs"].copy_(obs, non_blocking=True); io["rew"].copy_(rew, non_blocking=True)
            io["done"].copy_(io["done_dev"], non_blocking=True); io["count"].copy_(self._ev_cur, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            k = 0
            if int(io["count"].sum()) > 0:                           # the records of the few episodes that just ended
                ints, flts = self._gather_events(io["count"].tolist())
                k = ints.shape[0]
                io["ev_int"][:k].copy_(ints, non_blocking=True)
                io["ev_flt"][:k].copy_(flts, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()
            infos = self._infos_from(io["ev_int"][:k].numpy(), io["ev_flt"][:k].numpy())
            return io["obs"].numpy().copy(), io["rew"].numpy().copy(), io["done"].numpy().astype(bool), infos
        return obs, rew, (term | trunc).bool(), None

    def _host_io(self):
        """Pinned host staging buffers of the NumPy surface (allocated on first use)."""
        if getattr(self, "_io", None) is None:
            pin = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)                       # noqa: E731
            act = pin((self.n, L.FD_ACT_DIM), torch.float32)
            self._io = {"act": act, "act_np": act.numpy(), "act_dev": torch.empty((self.n, L.FD_ACT_DIM), dtype=torch.float32, device=self.device),
                        "obs": pin(tuple(self.obs.shape), self.obs.dtype), "rew": pin(tuple(self.rewards.shape), self.rewards.dtype),
                        "done": pin((self.n,), torch.uint8), "done_dev": torch.empty(self.n, dtype=torch.uint8, device=self.device),
                        "count": pin((L.FD_EV_SHARDS,), torch.int32), "ev_int": pin(tuple(self.ev_int.shape), self.ev_int.dtype),
                        "ev_flt": pin(tuple(self.ev_flt.shape), self.ev_flt.dtype)}
            self._info_list, self._info_dirty = [{} for _ in range(self.n)], []
        return self._io

    def _infos_from(self, ints, flts):
        """The vec-env info list without building N dicts per step: entries of envs that finished get a fresh dict, the ones
        written last step are reset, everything else stays the (empty) dict it was."""
        lst = self._info_list
        for i in self._info_dirty:
            lst[i] = {}
        dirty = []
        for (env, length, term), f in zip(ints, flts):
            lst[env] = {"episode": {"r": float(f[0]), "l": int(length)}, "terminal_observation": f[1:].copy(),
                        "TimeLimit.truncated": not bool(term)}
            dirty.append(int(env))
        self._info_dirty = dirty
        return list(lst)

    def episode_events_host(self):
        """(ints, floats) NumPy records of the episodes that ended in the last step."""
        ints, flts = self.episode_events()
        return ints.cpu().numpy(), flts.cpu().numpy()

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        a, self._pending = self._pending, None
        return self.step(a)

    @property
    def ev_count(self) -> torch.Tensor:
        """[1] int32: episodes that ended in the last step (sum over the record shards)."""
        return self._ev_cur.sum(dtype=torch.int32).reshape(1)

    def _gather_events(self, counts):
        """Dense (ints, floats) from the sharded record list; `counts` = the shard counters as a host list."""
        cs = self._ev_cap_shard
        segs = [(s * cs, min(int(c), cs)) for s, c in enumerate(counts) if c > 0]
        if not segs:
            return self.ev_int[:0], self.ev_flt[:0]
        if len(segs) == 1:
            lo, k = segs[0]
            return self.ev_int[lo:lo + k], self.ev_flt[lo:lo + k]
        idx = torch.cat([torch.arange(lo, lo + k, device=self.device) for lo, k in segs])
        return self.ev_int.index_select(0, idx), self.ev_flt.index_select(0, idx)

    def episode_events(self):
        """Compacted records of the episodes that ended in the last step (one small D2H copy of the shard counters)."""
        return self._gather_events(self._ev_cur.tolist())

    def episode_infos(self):
        """Per-env info dicts in the vec-env convention: `episode` = {r, l}, `terminal_observation`."""
        infos = [{} for _ in range(self.n)]
        ints, flts = self.episode_events()
        ints, flts = ints.cpu().numpy(), flts.cpu().numpy()
        for (env, length, term), f in zip(ints, flts):
            infos[env] = {"episode": {"r": float(f[0]), "l": int(length)}, "terminal_observation": f[1:].copy(),
                          "TimeLimit.truncated": not bool(term)}
        return infos

    def _out(self, t):
        return t.cpu().numpy() if self.numpy_io else t

    def close(self):
        pass

    def seed(self, seed=None):
        self.seed_value = 0 if seed is None else int(seed)
        return [self.seed_value + i for i in range(self.n)]

    def get_attr(self, name, indices=None):
        return [getattr(self, name)] * self.n

    def set_attr(self, name, value, indices=None):
        setattr(self, name, value)

    def env_method(self, name, *a, indices=None, **kw):
        return [getattr(self, name)(*a, **kw)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.n

    @property
    def rate_command(self) -> torch.Tensor:
        return self.e[L.FD_E_CMD_P:L.FD_E_CMD_R + 1].T
