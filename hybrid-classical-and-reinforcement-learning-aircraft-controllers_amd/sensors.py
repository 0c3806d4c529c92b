"""Sensor layer for fleets: host mirror of interfaces/sensor.py (SensorInterface :7-101, PerfectSensorInterface :104-134,
NoisySensorInterface :137-268) over `fdyn_sensor_update_*`.

The reference wraps ONE AircraftState per update; here `update` takes the word-major state block of N aircraft
(`BatchedSixDOF.x`, `GpuRateVecEnv.x`: [12][N]) and `get_state` hands back the measurement block [FD_NMS][N]
(12 state words + airspeed + altitude).  Method names, config keys, defaults and error behaviour follow the reference.
"""
from typing import Optional

import numpy as np
import torch

from . import _lib, layout as L

_DEFAULTS = (("gps_position_stddev", 1.0, L.FD_SN_GPS_POS), ("gps_velocity_stddev", 0.1, L.FD_SN_GPS_VEL),
             ("attitude_stddev", 0.01, L.FD_SN_ATTITUDE), ("imu_gyro_stddev", 0.01, L.FD_SN_GYRO),
             ("airspeed_stddev", 0.5, L.FD_SN_AIRSPEED), ("altitude_stddev", 0.5, L.FD_SN_ALTITUDE))


def noise_block(noise_config: Optional[dict] = None) -> np.ndarray:
    """[FD_NSN] float64 from the reference's config keys (sensor.py:172-182); the bias walks are its constants (:230-231)."""
    cfg = noise_config or {}
    c = np.zeros(L.FD_NSN, dtype=np.float64)
    for key, default, slot in _DEFAULTS:
        c[slot] = cfg.get(key, default)
    c[L.FD_SN_GYRO_BIAS_WALK], c[L.FD_SN_ACCEL_BIAS_WALK] = 0.0001, 0.001
    c[L.FD_SN_ENABLED] = float(bool(cfg.get("enabled", True)))
    return c


class SensorInterface:
    """sensor.py:7-101."""

    def get_state(self):
        raise NotImplementedError

    def update(self, true_state, derived=None) -> None:
        raise NotImplementedError

    def reset(self) -> None:
        raise NotImplementedError

    def get_sensor_type(self) -> str:
        raise NotImplementedError

    def is_perfect(self) -> bool:
        return self.get_sensor_type() == "perfect"

    def get_noise_parameters(self) -> dict:
        return {}

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(type={self.get_sensor_type()})"


class PerfectSensorInterface(SensorInterface):
    """sensor.py:104-134: hands the truth through (the state block itself, no copy)."""

    def __init__(self):
        self._state = None

    def get_state(self):
        if self._state is None:
            raise RuntimeError("Sensor not yet updated with state")
        return self._state

    def update(self, true_state, derived=None) -> None:
        self._state = true_state

    def reset(self) -> None:
        self._state = None

    def get_sensor_type(self) -> str:
        return "perfect"


class NoisySensorInterface(SensorInterface):
    """sensor.py:137-268 for N aircraft.  `noise_config` keys as the reference (`enabled`, `imu_gyro_stddev`,
    `imu_accel_stddev`, `gps_position_stddev`, `gps_velocity_stddev`, `airspeed_stddev`, `altitude_stddev`,
    `attitude_stddev`, `seed`).  Noise comes from in-kernel Philox keyed by (seed, aircraft, update count); pass `z`
    ([FD_NSZ][N] standard normals in the reference's draw order) to `update` to replay a NumPy stream instead."""

    def __init__(self, noise_config: dict, n: int = 1, precision: str = "f64", device=None):
        self.lib = _lib.load()
        self.device = device or _lib.require_gpu()
        self._config = noise_config
        self._enabled = noise_config.get("enabled", True)
        self._accel_noise = noise_config.get("imu_accel_stddev", 0.1)          # kept for get_noise_parameters only (:176)
        self.n, self.dtype = int(n), _lib.state_dtype(precision)
        self._fn = self.lib.fdyn_sensor_update_f32 if self.dtype == torch.float32 else self.lib.fdyn_sensor_update_f64
        self._cfg_host = noise_block(noise_config)
        self._cfg = torch.as_tensor(self._cfg_host, device=self.device)
        seed = noise_config.get("seed", None)
        self._seed = int(seed) if seed is not None else int(np.random.SeedSequence().entropy & (2 ** 63 - 1))
        self._bias = torch.zeros((L.FD_NSB, self.n), dtype=self.dtype, device=self.device)
        self._meas = torch.zeros((L.FD_NMS, self.n), dtype=self.dtype, device=self.device)
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._state = None

    def get_state(self) -> torch.Tensor:
        if self._state is None:
            raise RuntimeError("Sensor not yet updated with state")
        return self._state

    def update(self, true_state: torch.Tensor, derived: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None) -> None:
        assert true_state.shape == (L.FD_NX, self.n) and true_state.dtype == self.dtype
        assert derived is None or (derived.shape == (L.FD_ND, self.n) and derived.dtype == self.dtype)
        assert z is None or (z.shape == (L.FD_NSZ, self.n) and z.dtype == self.dtype)
        self._step.add_(1)
        rc = self._fn(_lib.ptr(true_state), _lib.ptr(derived), _lib.ptr(self._bias), _lib.ptr(self._cfg), _lib.ptr(z),
                      self._seed, self._step.data_ptr(), _lib.ptr(self._meas), self.n, _lib.current_stream())
        _lib.check(rc, "NoisySensorInterface.update")
        self._state = self._meas

    def reset(self) -> None:
        self._bias.zero_()
        self._state = None

    @property
    def gyro_bias(self) -> torch.Tensor:
        return self._bias[:3]

    @property
    def accel_bias(self) -> torch.Tensor:
        return self._bias[3:]

    def get_sensor_type(self) -> str:
        return "noisy"

    def get_noise_parameters(self) -> dict:
        c = self._cfg_host
        return {"enabled": self._enabled, "imu_gyro_stddev": c[L.FD_SN_GYRO], "imu_accel_stddev": self._accel_noise,
                "gps_position_stddev": c[L.FD_SN_GPS_POS], "gps_velocity_stddev": c[L.FD_SN_GPS_VEL],
                "airspeed_stddev": c[L.FD_SN_AIRSPEED], "altitude_stddev": c[L.FD_SN_ALTITUDE],
                "attitude_stddev": c[L.FD_SN_ATTITUDE]}


class ObservationNoise:
    """The noise model applied in place to rate-control observations [N][18] (`fdyn_sensor_observe`): what
    `GpuRateVecEnv(sensor_noise={...})` runs after every reset / step so the policy sees measured, not true, rates."""

    def __init__(self, noise_config: dict, n: int, device=None):
        self.lib = _lib.load()
        self.device = device or _lib.require_gpu()
        self.n = int(n)
        self._cfg = torch.as_tensor(noise_block(noise_config), device=self.device)
        seed = noise_config.get("seed", None)
        self._seed = int(seed) if seed is not None else int(np.random.SeedSequence().entropy & (2 ** 63 - 1))
        self.gyro_bias = torch.zeros((3, self.n), dtype=torch.float32, device=self.device)
        self._step = torch.zeros(1, dtype=torch.int32, device=self.device)

    def apply(self, obs: torch.Tensor, reset_mask: Optional[torch.Tensor] = None, z: Optional[torch.Tensor] = None):
        assert obs.shape == (self.n, L.FD_OBS_DIM) and obs.dtype == torch.float32 and obs.is_contiguous()
        assert reset_mask is None or (reset_mask.shape == (self.n,) and reset_mask.dtype == torch.uint8)
        assert z is None or (z.shape == (L.FD_NSZ, self.n) and z.dtype == torch.float32)
        self._step.add_(1)
        rc = self.lib.fdyn_sensor_observe(_lib.ptr(obs), _lib.ptr(self.gyro_bias), _lib.ptr(reset_mask), _lib.ptr(self._cfg),
                                          _lib.ptr(z), self._seed, self._step.data_ptr(), self.n, _lib.current_stream())
        _lib.check(rc, "ObservationNoise.apply")
        return obs

    def reset(self):
        self.gyro_bias.zero_()
