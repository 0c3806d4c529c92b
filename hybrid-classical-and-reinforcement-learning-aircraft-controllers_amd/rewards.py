"""Reward functions on recorded data: host mirror of learned_controllers/envs/rewards.py over `fdyn_rate_reward_seq_*`.

Inside training the reward is part of the fused env step (`rate_env_step_kernel`, default weights).  These classes keep the
reference's objects for everything around it -- reward shaping studies, re-scoring logged flights, the reference's
example_usage.py:104-143 -- with the same constructors, `compute(...)` signatures, `reset()` and carried state:

  RateTrackingReward(w_tracking, ..., settling_threshold).compute(p_err, q_err, r_err, action, prev_action, airspeed,
      altitude, roll, pitch) -> (total, components dict)                                          rewards.py:48-137
  SettlingTimeBonus(settling_threshold, min_settle_time, bonus_multiplier).compute(p_err, q_err, r_err, p_cmd, q_cmd,
      r_cmd, dt) -> bonus                                                                          rewards.py:168-221
  score_sequences(...)  the batched form: n sequences x T steps in one launch, components included.

No CPU path: the arithmetic runs in the HIP kernel (one lane per sequence) and fails loudly without the library.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, layout as L

DEFAULT_PARAMS = {"w_tracking": 0.5, "w_smoothness": 0.01, "w_stability": 0.3, "w_oscillation": 0.1, "w_survival": 1.0,
                  "settling_threshold": 0.05, "min_settle_time": 0.2, "bonus_multiplier": 2.0}
_ORDER = ("w_tracking", "w_smoothness", "w_stability", "w_oscillation", "w_survival", "settling_threshold", "min_settle_time",
          "bonus_multiplier")
COMPONENTS = ("tracking", "smoothness", "stability", "oscillation", "survival")


def params_block(**overrides) -> torch.Tensor:
    """[FD_NRW] fp64 parameter block (host tensor; copied to the device by the caller)."""
    vals = dict(DEFAULT_PARAMS, **overrides)
    return torch.tensor([float(vals[k]) for k in _ORDER], dtype=torch.float64)


def score_sequences(errs, actions, prev0, flight, cmd, dt: float, params: Optional[torch.Tensor] = None,
                    rstate: Optional[torch.Tensor] = None, want_components: bool = True):
    """errs [T,3,n], actions [T,4,n], prev0 [4,n], flight [T,4,n] (airspeed, altitude, roll, pitch), cmd [3,n] -- device
    tensors, fp64 or fp32 -> dict(tracking [T,n], components [T,5,n] | None, settle [T,n], settled [T,n] uint8, rstate [8,n]).
    `rstate` carries RateTrackingReward.prev_errors / sign_changes and the SettlingTimeBonus timer between calls."""
    lib = _lib.load()
    T, _, n = errs.shape
    dt_ = errs.dtype
    assert dt_ in (torch.float64, torch.float32) and errs.is_cuda, "device tensors, fp64 or fp32"
    assert actions.shape == (T, 4, n) and prev0.shape == (4, n) and flight.shape == (T, L.FD_NRF, n) and cmd.shape == (3, n)
    dev = errs.device
    args = [t.to(dt_).contiguous() for t in (errs, actions, prev0, flight, cmd)]
    params = (params_block() if params is None else params).to(dev, torch.float64).contiguous()
    assert params.numel() == L.FD_NRW
    if rstate is None:
        rstate = torch.zeros((L.FD_NRS, n), dtype=dt_, device=dev)
    assert rstate.shape == (L.FD_NRS, n) and rstate.dtype == dt_ and rstate.is_contiguous()
    tracking = torch.empty((T, n), dtype=dt_, device=dev)
    comps = torch.empty((T, L.FD_NRC, n), dtype=dt_, device=dev) if want_components else None
    settle = torch.empty((T, n), dtype=dt_, device=dev)
    settled = torch.empty((T, n), dtype=torch.uint8, device=dev)
    fn = lib.fdyn_rate_reward_seq_f64 if dt_ == torch.float64 else lib.fdyn_rate_reward_seq_f32
    _lib.check(fn(*(_lib.ptr(a) for a in args), _lib.ptr(params), _lib.ptr(rstate), float(dt), T, n, _lib.ptr(tracking),
                  _lib.ptr(comps), _lib.ptr(settle), _lib.ptr(settled), _lib.current_stream()), "rate_reward_seq")
    return {"tracking": tracking, "components": comps, "settle": settle, "settled": settled, "rstate": rstate}


class _Scorer:
    """One sequence, one step per call: shared plumbing of the two reference classes."""

    def __init__(self, **params):
        self._params = params_block(**params)
        self._dev = torch.device("cuda", torch.cuda.current_device())
        self._rstate = torch.zeros((L.FD_NRS, 1), dtype=torch.float64, device=self._dev)

    def _step(self, err, action, prev_action, flight, cmd, dt):
        col = lambda v, rows: torch.as_tensor(np.asarray(v, dtype=np.float64).reshape(1, rows, 1), device=self._dev)   # noqa: E731
        out = score_sequences(col(err, 3), col(action, 4), col(prev_action, 4)[0], col(flight, 4), col(cmd, 3)[0], dt,
                              self._params, self._rstate)
        return out

    def reset(self):
        self._rstate.zero_()


class RateTrackingReward(_Scorer):
    """learned_controllers/envs/rewards.py:6-146."""

    def __init__(self, w_tracking: float = 0.5, w_smoothness: float = 0.01, w_stability: float = 0.3, w_oscillation: float = 0.1,
                 w_survival: float = 1.0, settling_threshold: float = 0.1):
        self.w_tracking, self.w_smoothness, self.w_stability = w_tracking, w_smoothness, w_stability
        self.w_oscillation, self.w_survival, self.settling_threshold = w_oscillation, w_survival, settling_threshold
        super().__init__(w_tracking=w_tracking, w_smoothness=w_smoothness, w_stability=w_stability, w_oscillation=w_oscillation,
                         w_survival=w_survival)

    @property
    def prev_errors(self) -> np.ndarray:
        return self._rstate[L.FD_RS_PERR_P:L.FD_RS_PERR_R + 1, 0].cpu().numpy()

    @property
    def sign_changes(self) -> np.ndarray:
        return self._rstate[L.FD_RS_SIGN_P:L.FD_RS_SIGN_R + 1, 0].cpu().numpy()

    def compute(self, p_error: float, q_error: float, r_error: float, action, prev_action, airspeed: float, altitude: float,
                roll: float, pitch: float) -> Tuple[float, Dict[str, float]]:
        out = self._step([p_error, q_error, r_error], action, prev_action, [airspeed, altitude, roll, pitch], [0.0, 0.0, 0.0], 0.0)
        total = float(out["tracking"][0, 0])
        c = out["components"][0, :, 0].cpu().numpy()
        comps = {name: float(c[k]) for k, name in enumerate(COMPONENTS)}
        comps["total"] = total
        comps["tracking_error_mse"] = (p_error ** 2 + q_error ** 2 + r_error ** 2) / 3.0                      # :76
        return total, comps


class SettlingTimeBonus(_Scorer):
    """learned_controllers/envs/rewards.py:149-221."""

    def __init__(self, settling_threshold: float = 0.05, min_settle_time: float = 0.2, bonus_multiplier: float = 2.0):
        self.settling_threshold, self.min_settle_time, self.bonus_multiplier = settling_threshold, min_settle_time, bonus_multiplier
        super().__init__(settling_threshold=settling_threshold, min_settle_time=min_settle_time, bonus_multiplier=bonus_multiplier)

    @property
    def settle_timer(self) -> float:
        return float(self._rstate[L.FD_RS_SETTLE_TIMER, 0])

    @property
    def is_settled(self) -> bool:
        return bool(self._rstate[L.FD_RS_IS_SETTLED, 0] != 0)

    def compute(self, p_error: float, q_error: float, r_error: float, p_cmd: float, q_cmd: float, r_cmd: float, dt: float) -> float:
        z4 = [0.0, 0.0, 0.0, 0.0]
        out = self._step([p_error, q_error, r_error], z4, z4, [20.0, 100.0, 0.0, 0.0], [p_cmd, q_cmd, r_cmd], dt)
        return float(out["settle"][0, 0])
