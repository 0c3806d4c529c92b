"""GPU parity of the sensor layer (csrc/sensor_kernels.hip, C-ABI fdyn_sensor_update_* / fdyn_sensor_observe) against the
reference's NoisySensorInterface outputs (tests/golden/sensor_noisy.npz) and the CPU oracle.

Tolerances.  With the standard normals supplied (parity mode) the fp64 kernel is bit-exact (one multiply and one or two
adds per word, contraction off).  The in-kernel Philox mode is checked statistically.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from hcrl_amd import layout as L
from hcrl_amd.rate_env import GpuRateVecEnv
from hcrl_amd.sensors import NoisySensorInterface, ObservationNoise, PerfectSensorInterface, noise_block

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cfg_dict(c, seed=0):
    return {"gps_position_stddev": c[0], "gps_velocity_stddev": c[1], "attitude_stddev": c[2], "imu_gyro_stddev": c[3],
            "airspeed_stddev": c[4], "altitude_stddev": c[5], "enabled": bool(c[8]), "seed": seed}


@pytest.mark.parametrize("tag", ["default", "custom", "disabled"])
def test_update_replays_reference_sensor_bit_exact(tag):
    g = load_golden("sensor_noisy.npz")
    x, z, meas, bias, va = (g[f"{tag}_{k}"] for k in ("x", "z", "meas", "bias", "airspeed_altitude"))
    n = 3                                                       # the same aircraft in every lane
    sens = NoisySensorInterface(_cfg_dict(g[f"{tag}_cfg"]), n=n, precision="f64")
    with pytest.raises(RuntimeError):
        sens.get_state()
    assert sens.get_sensor_type() == "noisy" and not sens.is_perfect()
    for k in range(len(x)):
        xt = torch.as_tensor(np.repeat(x[k][:, None], n, 1), device=DEV)
        d = torch.zeros((L.FD_ND, n), dtype=torch.float64, device=DEV)
        d[L.FD_D_AIRSPEED], d[L.FD_D_ALTITUDE] = float(va[k, 0]), float(va[k, 1])
        sens.update(xt, d, torch.as_tensor(np.repeat(z[k][:, None], n, 1), device=DEV))
        got = sens.get_state().cpu().numpy()
        assert np.array_equal(got, np.repeat(meas[k][:, None], n, 1)), (k, got[:, 0] - meas[k])
        assert np.array_equal(torch.cat([sens.gyro_bias, sens.accel_bias]).cpu().numpy(), np.repeat(bias[k][:, None], n, 1)), k
    sens.reset()
    assert float(sens.gyro_bias.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        sens.get_state()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_update_batch_against_oracle(oracle, precision):
    rs = np.random.RandomState(3)
    n = 5000
    dt = np.float64 if precision == "f64" else np.float32
    x = rs.normal(0, 20, (L.FD_NX, n)).astype(dt)
    z = rs.normal(size=(L.FD_NSZ, n)).astype(dt)
    b0 = (rs.normal(0, 0.01, (L.FD_NSB, n))).astype(dt)
    cfg = {"imu_gyro_stddev": 0.02, "gps_position_stddev": 1.7, "airspeed_stddev": 0.4, "seed": 1}
    sens = NoisySensorInterface(cfg, n=n, precision=precision)
    sens._bias.copy_(torch.as_tensor(b0, device=DEV))
    sens.update(torch.as_tensor(x, device=DEV), None, torch.as_tensor(z, device=DEV))
    got, gb = sens.get_state().cpu().numpy().astype(np.float64), sens._bias.cpu().numpy().astype(np.float64)
    c = noise_block(cfg)
    tol = 0.0 if precision == "f64" else 2e-6
    for i in range(0, n, 7):
        xi = x[:, i].astype(np.float64)
        bias = b0[:, i].astype(np.float64).copy()
        va = np.sqrt(xi[3] * xi[3] + xi[4] * xi[4] + xi[5] * xi[5])
        want = oracle.sensor_update(xi, va, -xi[2], bias, c, z[:, i].astype(np.float64))
        # airspeed computed in-kernel from x (no derived rows given): sqrt may differ in the last ulp from libm
        assert np.all(np.abs(got[:12, i] - want[:12]) <= tol * np.maximum(np.abs(want[:12]), 1.0)), (i, got[:, i] - want)
        assert np.all(np.abs(got[12:, i] - want[12:]) <= max(tol, 1e-15) * np.maximum(np.abs(want[12:]), 1.0))
        assert np.all(np.abs(gb[:, i] - bias) <= tol), i


def test_philox_mode_statistics_and_bias_walk():
    n, K = 1 << 16, 64
    cfg = {"seed": 5, "imu_gyro_stddev": 0.03, "gps_position_stddev": 2.0, "gps_velocity_stddev": 0.2,
           "attitude_stddev": 0.015, "airspeed_stddev": 0.7, "altitude_stddev": 1.2}
    sens = NoisySensorInterface(cfg, n=n, precision="f64")
    x = torch.zeros((L.FD_NX, n), dtype=torch.float64, device=DEV)
    x[L.FD_X_U] = 20.0
    first = None
    for k in range(K):
        sens.update(x)
        if k == 0:
            first = sens.get_state().clone()
    m = first
    want_std = [2.0] * 3 + [0.2] * 3 + [0.015] * 3 + [0.03] * 3 + [0.7, 1.2]
    truth = [0.0] * 3 + [20.0, 0.0, 0.0] + [0.0] * 6 + [20.0, 0.0]
    for row in range(L.FD_NMS):
        d = m[row] - truth[row]
        assert abs(float(d.mean())) < 5 * want_std[row] / np.sqrt(n) + 1e-4, row
        assert abs(float(d.std()) / want_std[row] - 1.0) < 0.02, (row, float(d.std()))
    # rows are independent draws, successive updates differ
    assert abs(float(torch.corrcoef(torch.stack([m[0], m[1]]))[0, 1])) < 0.02
    assert float((sens.get_state()[0] - m[0]).abs().mean()) > 0.5
    # bias random walks: variance K * w^2  (w = 1e-4 gyro, 1e-3 accel; sensor.py:230-231)
    assert abs(float(sens.gyro_bias.std()) / (1e-4 * np.sqrt(K)) - 1.0) < 0.03
    assert abs(float(sens.accel_bias.std()) / (1e-3 * np.sqrt(K)) - 1.0) < 0.03
    # same seed, same stream; another seed, another stream
    again = NoisySensorInterface(cfg, n=n, precision="f64"); again.update(x)
    other = NoisySensorInterface({**cfg, "seed": 6}, n=n, precision="f64"); other.update(x)
    assert torch.equal(again.get_state(), first) and not torch.equal(other.get_state(), first)


def test_perfect_sensor_passes_the_block_through():
    s = PerfectSensorInterface()
    with pytest.raises(RuntimeError):
        s.get_state()
    x = torch.zeros((L.FD_NX, 4), dtype=torch.float64, device=DEV)
    s.update(x)
    assert s.get_state() is x and s.is_perfect() and s.get_noise_parameters() == {}


def test_observation_noise_against_numpy_and_reset_mask():
    rs = np.random.RandomState(1)
    n = 4096
    cfg = {"seed": 2, "imu_gyro_stddev": 0.02, "attitude_stddev": 0.01, "airspeed_stddev": 0.5, "altitude_stddev": 0.8}
    c = noise_block(cfg).astype(np.float32)
    obs0 = rs.normal(0, 1, (n, L.FD_OBS_DIM)).astype(np.float32)
    z = rs.normal(size=(L.FD_NSZ, n)).astype(np.float32)
    b0 = rs.normal(0, 0.01, (3, n)).astype(np.float32)
    mask = (rs.uniform(size=n) < 0.3).astype(np.uint8)
    on = ObservationNoise(cfg, n)
    on.gyro_bias.copy_(torch.as_tensor(b0, device=DEV))
    obs = torch.as_tensor(obs0.copy(), device=DEV)
    on.apply(obs, torch.as_tensor(mask, device=DEV), torch.as_tensor(z, device=DEV))
    got, gb = obs.cpu().numpy(), on.gyro_bias.cpu().numpy()
    b = np.where(mask[None, :] != 0, np.float32(0), b0)
    want = obs0.copy()
    rate = (obs0[:, 0:3] + c[L.FD_SN_GYRO] * z[L.FD_SZ_GYRO:L.FD_SZ_GYRO + 3].T) + b.T
    want[:, 0:3] = rate
    want[:, 6:9] = obs0[:, 3:6] - rate
    want[:, 11:14] = obs0[:, 11:14] + c[L.FD_SN_ATTITUDE] * z[L.FD_SZ_ATT:L.FD_SZ_ATT + 3].T
    want[:, 9] = obs0[:, 9] + c[L.FD_SN_AIRSPEED] * z[L.FD_SZ_AIRSPEED]
    want[:, 10] = obs0[:, 10] + c[L.FD_SN_ALTITUDE] * z[L.FD_SZ_ALTITUDE]
    assert np.array_equal(got, want)
    assert np.array_equal(gb, b + c[L.FD_SN_GYRO_BIAS_WALK] * z[L.FD_SZ_GYRO_BIAS:L.FD_SZ_GYRO_BIAS + 3])
    assert np.array_equal(got[:, 3:6], obs0[:, 3:6]) and np.array_equal(got[:, 14:], obs0[:, 14:])


def test_env_with_sensor_noise_keeps_truth_inside_and_measured_outside():
    n = 8192
    noisy = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=4, precision="mixed", sampling="device",
                          sensor_noise={"seed": 9, "imu_gyro_stddev": 0.02})
    clean = GpuRateVecEnv(n, "medium", 10.0, 0.02, "step", seed=4, precision="mixed", sampling="device")
    o_n, o_c = noisy.reset().clone(), clean.reset().clone()
    d = o_n - o_c
    assert abs(float(d[:, 0:3].std()) / 0.02 - 1.0) < 0.05 and abs(float(d[:, 9].std()) / 0.5 - 1.0) < 0.05
    assert torch.equal(o_n[:, 3:6], o_c[:, 3:6]) and torch.equal(o_n[:, 14:], o_c[:, 14:])
    assert torch.allclose(o_n[:, 6:9], o_n[:, 3:6] - o_n[:, 0:3], atol=1e-6)
    a = torch.zeros((n, 4), device=DEV); a[:, 3] = 0.6
    for _ in range(30):
        noisy.step_device(a)
        clean.step_device(a)
    # the physics and the reward see the truth: identical states and rewards; only the observation is measured
    assert torch.equal(noisy.x, clean.x) and torch.equal(noisy.rewards, clean.rewards)
    d = noisy.obs - clean.obs
    assert abs(float(d[:, 1].std()) / 0.02 - 1.0) < 0.05 and float(noisy.sensor.gyro_bias.abs().max()) > 0.0
