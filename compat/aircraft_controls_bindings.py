"""Source-compatible stand-in for the reference's pybind11 module `aircraft_controls_bindings` v1.1.0
(cpp/bindings/bindings.cpp:13-164; classes cpp/include/pid_controller.h:22-161), computed by the HIP library.

Put this directory on `sys.path` and the reference's `import aircraft_controls_bindings as acb` resolves here.
Every `compute` is one launch of `fdyn_pid_compute_batch` (n = 1, or n = 3 for the multi-axis controller): the
arithmetic is the bit-faithful fp32 PID of cpp/src/pid_controller.cpp:24-60 running on the GPU; all floats are
narrowed to 32 bits on entry exactly like the pybind module.  This scalar API exists for source compatibility
(~tens of microseconds per call); the data path for fleets is the fused cascade / env kernels.
"""
import os
import sys

import numpy as np
import torch

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)
import hcrl_amd  # noqa: E402,F401
from hcrl_amd import _lib, layout as L  # noqa: E402

__version__ = "1.1.0"
_f32 = np.float32


class PIDGains:
    def __init__(self, kp=0.0, ki=0.0, kd=0.0):
        self.kp, self.ki, self.kd = float(_f32(kp)), float(_f32(ki)), float(_f32(kd))

    def __repr__(self):
        return f"PIDGains(kp={self.kp:f}, ki={self.ki:f}, kd={self.kd:f})"


class PIDConfig:
    def __init__(self):
        self.gains = PIDGains()
        self.output_min, self.output_max = -1.0, 1.0
        self.integral_min, self.integral_max = -10.0, 10.0
        self.derivative_filter_alpha = 0.1

    def _row(self):
        g = self.gains
        return np.array([g.kp, g.ki, g.kd, self.output_min, self.output_max, self.integral_min, self.integral_max,
                         self.derivative_filter_alpha], dtype=np.float32)

    def _copy(self):
        c = PIDConfig()
        c.gains = PIDGains(self.gains.kp, self.gains.ki, self.gains.kd)
        c.output_min, c.output_max, c.integral_min, c.integral_max = self.output_min, self.output_max, self.integral_min, self.integral_max
        c.derivative_filter_alpha = self.derivative_filter_alpha
        return c

    def __repr__(self):
        return (f"PIDConfig(kp={self.gains.kp:f}, ki={self.gains.ki:f}, kd={self.gains.kd:f}, "
                f"output_range=[{self.output_min:f}, {self.output_max:f}])")


class _Bank:
    """n independent PIDs resident on the device; value semantics for configs (pid_controller.cpp:12-19)."""

    def __init__(self, configs):
        self.lib, self.dev = _lib.load(), _lib.require_gpu()
        self.cfgs = [c._copy() for c in configs]
        self.n = len(configs)
        self.state = torch.zeros((L.FD_NPS, self.n), dtype=torch.float32, device=self.dev)
        self.out = torch.zeros(self.n, dtype=torch.float32, device=self.dev)
        self.error = np.zeros(self.n, np.float32)
        self.last_out = np.zeros(self.n, np.float32)
        self._upload()

    def _upload(self):
        self.cfg_dev = torch.as_tensor(np.stack([c._row() for c in self.cfgs]), device=self.dev).contiguous()

    def compute(self, sp, ms, dt):
        sp32, ms32 = np.asarray(sp, np.float32), np.asarray(ms, np.float32)
        spd, msd = torch.as_tensor(sp32, device=self.dev), torch.as_tensor(ms32, device=self.dev)
        _lib.check(self.lib.fdyn_pid_compute_batch(_lib.ptr(self.cfg_dev), 1, _lib.ptr(self.state), _lib.ptr(spd), _lib.ptr(msd),
                                                   float(_f32(dt)), _lib.ptr(self.out), self.n, _lib.current_stream()))
        self.error = sp32 - ms32
        self.last_out = self.out.cpu().numpy()
        return self.last_out

    def reset(self):
        self.state.zero_()
        self.error[:] = 0
        self.last_out[:] = 0


class PIDController:
    def __init__(self, config=None):
        self._b = _Bank([config or PIDConfig()])

    def compute(self, setpoint, measurement, dt):
        return float(self._b.compute([setpoint], [measurement], dt)[0])

    def reset(self):
        self._b.reset()

    def set_gains(self, gains):
        self._b.cfgs[0].gains = PIDGains(gains.kp, gains.ki, gains.kd)
        self._b._upload()

    def get_gains(self):
        g = self._b.cfgs[0].gains
        return PIDGains(g.kp, g.ki, g.kd)

    def get_error(self):
        return float(self._b.error[0])

    def get_integral(self):
        return float(self._b.state[L.FD_PS_INTEGRAL, 0])

    def get_derivative(self):
        return float(self._b.state[L.FD_PS_DFILT, 0])

    def get_output(self):
        return float(self._b.last_out[0])

    def __repr__(self):
        g = self.get_gains()
        return f"PIDController(kp={g.kp:f}, ki={g.ki:f}, kd={g.kd:f})"


class Vector3:
    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(_f32(x)), float(_f32(y)), float(_f32(z))

    def __repr__(self):
        return f"Vector3(x={self.x:f}, y={self.y:f}, z={self.z:f})"


class ControlOutput:
    def __init__(self, roll=0.0, pitch=0.0, yaw=0.0):
        self.roll, self.pitch, self.yaw = float(_f32(roll)), float(_f32(pitch)), float(_f32(yaw))

    def __repr__(self):
        return f"ControlOutput(roll={self.roll:f}, pitch={self.pitch:f}, yaw={self.yaw:f})"


class MultiAxisPIDController:
    def __init__(self, roll_config=None, pitch_config=None, yaw_config=None):
        self._b = _Bank([roll_config or PIDConfig(), pitch_config or PIDConfig(), yaw_config or PIDConfig()])

    def compute(self, setpoint, measurement, dt):
        o = self._b.compute([setpoint.x, setpoint.y, setpoint.z], [measurement.x, measurement.y, measurement.z], dt)
        return ControlOutput(o[0], o[1], o[2])

    def reset(self):
        self._b.reset()

    def set_gains(self, axis, gains):
        if axis in (0, 1, 2):                                         # invalid axis: no-op (pid_controller.cpp:120-127)
            self._b.cfgs[axis].gains = PIDGains(gains.kp, gains.ki, gains.kd)
            self._b._upload()

    def get_gains(self, axis):
        if axis in (0, 1, 2):
            g = self._b.cfgs[axis].gains
            return PIDGains(g.kp, g.ki, g.kd)
        return PIDGains()                                             # invalid axis: zero gains (:129-136)

    def get_error(self):
        return Vector3(*self._b.error)

    def get_integral(self):
        return Vector3(*self._b.state[L.FD_PS_INTEGRAL].cpu().numpy())

    def get_output(self):
        return ControlOutput(*self._b.last_out)

    def __repr__(self):
        return (f"MultiAxisPIDController(roll_kp={self.get_gains(0).kp:f}, pitch_kp={self.get_gains(1).kp:f}, "
                f"yaw_kp={self.get_gains(2).kp:f})")
