"""GPU parity AT SCALE: the benched precision variants against the on-box CPU oracle on thousands of DISTINCT aircraft.

The fixture tests (test_gpu_parity.py) pin 32-64 reference-generated trajectories.  These tests widen the sample the
north-star gate (1e-4 relative over 1000 steps, |a-b| / max(|b|, 1), roll / yaw modulo 2 pi) rests on:

  * cfg 2 (SURVEY §8d recipe, fresh seed): 4096 distinct initial conditions and control settings, 1000 RK4 steps at
    dt = 1 ms and at dt = 10 ms, `mixed` / `f32` and `f64` against `orc_sixdof_step_batch` (OpenMP, seconds);
  * the env: 4096 parity-sampled envs x 520 steps (so every surviving env is truncated at step 500 and auto-resets,
    and random actions crash some earlier) against `orc_env_step_batch` + `orc_env_reset`: rewards, flags, obs, states;
  * the register-capped (`OCC2`) build of the env step, which the launcher selects only above 65 536 envs:
    131 072 envs, first half bit-equal to a 65 536-env launch of the uncapped build on the same rows.

Measured percentiles go to stdout and to gpurun_out/drift.json (copied to profiles/drift.json, which bench.py quotes).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, rel_err, STATE_ANGLE_COLS
from hcrl_amd import layout as L, samplers
from hcrl_amd.fleet import BatchedSixDOF
from hcrl_amd.params import AircraftParams
from hcrl_amd.rate_env import GpuRateVecEnv

pytestmark = pytest.mark.gpu

N = 4096
_DRIFT = {}


def _record(precision, key, per_item, **extra):
    q = {f"p{p}": float(np.percentile(per_item, p)) for p in (50, 90, 99)}
    q["max"] = float(per_item.max())
    q["n"] = int(per_item.size)
    q.update(extra)
    _DRIFT.setdefault(precision, {})[key] = q
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "drift.json"), "w") as f:
        json.dump(_DRIFT, f, indent=1, sort_keys=True)
    return q


def _cfg2_inputs(n, seed):
    """SURVEY §8d cfg 2: FlightEnvelopeSampler.sample's distribution and draw order, fixed controls."""
    rs = np.random.RandomState(seed)
    x0 = np.zeros((n, 12))
    x0[:, 3] = rs.uniform(15.0, 30.0, n)
    x0[:, 2] = -rs.uniform(50.0, 200.0, n)
    x0[:, 6] = rs.uniform(-np.radians(15), np.radians(15), n)
    x0[:, 7] = rs.uniform(-np.radians(15), np.radians(15), n)
    x0[:, 8] = rs.uniform(0.0, 2 * np.pi, n)
    x0[:, 9:12] = rs.uniform(-0.1, 0.1, (n, 3))
    u = np.concatenate([rs.uniform(-0.3, 0.3, (n, 3)), rs.uniform(0.3, 0.9, (n, 1))], 1)   # elevator, aileron, rudder, throttle
    return x0, u


_ORACLE_RUNS = {}


def _oracle_cfg2(oracle, dt, chunk=50, steps=1000):
    """The oracle's trajectory (checkpoints every `chunk` steps) plus what the REFERENCE MODEL says about each aircraft:
    regular = none of the reference's guards came within a margin of engaging during the flight: u_safe (|u| < min_u,
              simplified_6dof.py:368), the alpha clip (:370), the pitch clamp (:268, :463), the rate and velocity clamps
              (:262, :273), ground contact (:276) -- i.e. the dynamics stayed smooth;
    amp     = worst deviation of the oracle under a 1e-12 relative perturbation of the initial state (3 random sign
              patterns) / 1e-12 -- smooth amplification;
    model   = the oracle with the ARGUMENT of every step rounded to fp32 resolution (x~ = x (1 +- 6e-8), the increment of
              the step from x~ applied to the unrounded x): the floor of ANY scheme that evaluates the derivatives in fp32
              and accumulates in fp64 -- three noise sequences, worst deviation."""
    if dt in _ORACLE_RUNS:
        return _ORACLE_RUNS[dt]
    x0, u = _cfg2_inputs(N, seed=20261004)
    P = AircraftParams().to_block()
    us = np.ascontiguousarray(u.T)
    threads = min(16, int(oracle.lib.orc_max_threads()))
    step1 = lambda arr, k=1: oracle.lib.orc_sixdof_step_batch(oracle.dp(P), oracle.dp(arr), oracle.dp(us), N, dt * k, k, threads)  # noqa: E731
    rs = np.random.RandomState(1)
    base = np.ascontiguousarray(x0.T)
    p12 = [np.ascontiguousarray((x0 * (1 + 1e-12 * rs.choice([-1.0, 1.0], x0.shape))).T) for _ in range(3)]
    rnd = [np.ascontiguousarray(x0.T.copy()) for _ in range(3)]
    traj, amp, model, irregular = [], np.zeros(N), np.zeros(N), np.zeros(N, bool)
    min_u, max_alpha, max_pitch, max_rate, max_vel = (P[L.FD_P_MIN_U_VELOCITY], P[L.FD_P_MAX_ALPHA_RAD], P[L.FD_P_MAX_PITCH_RAD],
                                                      P[L.FD_P_MAX_RATE_RAD], P[L.FD_P_MAX_VELOCITY])
    for k in range(steps):
        step1(base)
        irregular |= (base[3] < 5 * min_u) | (np.abs(np.arctan2(base[5], np.maximum(base[3], min_u))) > max_alpha - np.radians(1.0)) | \
                     (np.abs(base[7]) > max_pitch - np.radians(1.0)) | (np.abs(base[9:12]).max(0) > 0.99 * max_rate) | \
                     (np.abs(base[3:6]).max(0) > 0.99 * max_vel) | (-base[2] < 0.5)
        for j in range(3):
            xt = np.ascontiguousarray(rnd[j] * (1 + 6e-8 * rs.choice([-1.0, 1.0], base.shape)))
            xt0 = xt.copy()
            step1(xt)
            inc = xt - xt0
            for c in STATE_ANGLE_COLS:                          # the step re-wraps roll / yaw: increment modulo 2 pi
                inc[c] = (inc[c] + np.pi) % (2 * np.pi) - np.pi
            rnd[j] = np.ascontiguousarray(rnd[j] + inc)
        if k % chunk == chunk - 1:
            traj.append(base.T.copy())
            for b in p12:
                step1(b, chunk)
                amp = np.maximum(amp, rel_err(b.T, base.T, STATE_ANGLE_COLS).max(1) / 1e-12)
            for b in rnd:
                model = np.maximum(model, rel_err(b.T, base.T, STATE_ANGLE_COLS).max(1))
    _ORACLE_RUNS[dt] = (x0, u, traj, amp, model, ~irregular)
    return _ORACLE_RUNS[dt]


@pytest.mark.parametrize("dt", [0.001, 0.01])
@pytest.mark.parametrize("precision", ["f64", "mixed", "f32"])
def test_cfg2_4096_distinct_aircraft_1000_steps_vs_oracle(oracle, precision, dt):
    """Forward error of every aircraft, reported as percentiles over ALL 4096, and gated where a gate can mean something.

    Ten seconds of open-loop flight with fixed +-0.3 surface deflections sends most of this fleet into tumbles, the rate and
    pitch clamps, ground contact and the u ~ 0 sign switch of u_safe: there the REFERENCE does not reproduce itself -- the
    oracle (C, libm) against the reference (NumPy) run here differs by up to 1.9 on 36 of 1024 aircraft
    (tests/test_oracle_scale.py).  So:
      f64   : <= 1e-9 wherever the oracle's own amplification of a 1e-12 input perturbation is <= 1e3 (97 % of the fleet at
              dt = 10 ms, all of it at 1 ms), and for EVERY aircraft error / amplification <= 1e-10;
      mixed : the north-star gate 1e-4 on EVERY aircraft at dt = 1 ms (1000 steps = 1 s of flight), and at dt = 10 ms on
              every aircraft that stays clear of the reference's guards (`regular`).  Over the WHOLE fleet its error
              distribution must not exceed 5 x the floor of fp32 evaluation (`model`: the oracle itself with the argument
              of each step rounded to fp32; the derivative arithmetic itself adds its own rounding, measured 2-3 x at dt = 1 ms,
              1.1 x at 10 ms) at the median and the 90th percentile;
      f32   : un-gated throughput variant, bulk bounded loosely."""
    x0, u, traj, amp, model, regular = _oracle_cfg2(oracle, dt)
    fl = BatchedSixDOF(N, precision)
    fl.reset(x0)
    fl.set_controls(u)
    chunk = 50
    worst = np.zeros(N)
    for k in range(len(traj)):
        for _ in range(chunk):
            fl.step(dt)
        worst = np.maximum(worst, rel_err(fl.state_numpy(), traj[k], STATE_ANGLE_COLS).max(1))
    well12 = amp <= 1e3
    extra = {"n_amp_le_1e3": int(well12.sum()), "max_where_amp_le_1e3": float(worst[well12].max()),
             "n_regular": int(regular.sum()), "max_regular": float(worst[regular].max()),
             "fp32_argument_model_p50": float(np.percentile(model, 50)), "fp32_argument_model_p90": float(np.percentile(model, 90)),
             "fp32_argument_model_p99": float(np.percentile(model, 99)), "n_over_1e-4": int((worst > 1e-4).sum()),
             "n_model_over_1e-4": int((model > 1e-4).sum())}
    q = _record(precision, f"cfg2_dt{dt:g}_1000_steps", worst, **extra)
    print(f"\n[drift@scale] cfg2 {precision} dt={dt:g}: {N} distinct aircraft x 1000 steps vs oracle: "
          f"p50 {q['p50']:.2e}  p90 {q['p90']:.2e}  p99 {q['p99']:.2e}  max {q['max']:.2e}; {extra['n_over_1e-4']} aircraft over 1e-4\n"
          f"   amplification <= 1e3 on {extra['n_amp_le_1e3']} aircraft, GPU max there {extra['max_where_amp_le_1e3']:.2e}; "
          f"clear of every reference guard: {extra['n_regular']} aircraft, GPU max there {extra['max_regular']:.2e}\n"
          f"   floor of fp32 evaluation (oracle with fp32-rounded step arguments): p50 {extra['fp32_argument_model_p50']:.2e} "
          f"p90 {extra['fp32_argument_model_p90']:.2e} p99 {extra['fp32_argument_model_p99']:.2e}; {extra['n_model_over_1e-4']} aircraft over 1e-4")
    if precision == "f64":
        assert worst[well12].max() <= 1e-9 and (worst / np.maximum(amp, 1.0)).max() <= 1e-10
        assert well12.sum() >= 0.95 * N
    elif precision == "mixed":
        if dt <= 0.001:
            assert q["max"] <= 1e-4, (q, np.argsort(worst)[-5:])          # the gate, every one of the 4096 aircraft
        assert regular.sum() >= 100 and worst[regular].max() <= 1e-4, (regular.sum(), worst[regular].max())
        assert q["p50"] <= 5 * extra["fp32_argument_model_p50"] and q["p90"] <= 5 * extra["fp32_argument_model_p90"]
    else:                                              # un-gated throughput variant: report, bound the bulk loosely
        assert q["p90"] < 1e-2


def _env_actions(n, steps, seed):
    """Smooth bounded surfaces around trim plus an occasional hard-over, so most envs fly and some crash."""
    rs = np.random.RandomState(seed)
    base = np.concatenate([(rs.rand(n, 3) - 0.5) * 0.4, 0.45 + 0.3 * rs.rand(n, 1)], 1)
    acts = np.empty((steps, n, 4), np.float32)
    for k in range(steps):
        jitter = np.concatenate([(rs.rand(n, 3) - 0.5) * 0.3, (rs.rand(n, 1) - 0.5) * 0.2], 1)
        a = base + jitter
        wild = rs.rand(n) < 0.002                     # ~1 env-step in 500 gets a saturating input (clip path, crashes)
        a[wild, :3] = rs.uniform(-1.5, 1.5, (int(wild.sum()), 3))
        acts[k] = a.astype(np.float32)
    return acts


class _OracleFleet:
    """N oracle envs in SoA form with vec-env auto-reset from the parity pool; `perturb` scales every freshly reset state by
    (1 + perturb * random sign) -- the copy that measures the reference's own sensitivity."""

    def __init__(self, oracle, pool, P, EC, perturb=0.0, seed=3):
        self.o, self.pool, self.P, self.EC, self.perturb = oracle, pool, P, EC, perturb
        self.n, self.depth = pool.shape[0], pool.shape[1]
        self.rs = np.random.RandomState(seed)
        self.xs, self.es = np.zeros((12, self.n)), np.zeros((L.FD_NE, self.n))
        self.eis = np.zeros((L.FD_NEI, self.n), np.int32)
        self.obs = np.zeros((self.n, 18), np.float32)
        self.rew, self.te, self.tr = np.zeros(self.n), np.zeros(self.n, np.int32), np.zeros(self.n, np.int32)
        self.episode = np.zeros(self.n, np.int64)
        self.threads = min(16, int(oracle.lib.orc_max_threads()))
        for i in range(self.n):
            self._reset(i)

    def _reset(self, i):
        o = self.o
        x, e, ei = np.zeros(12), np.zeros(L.FD_NE), self.eis[:, i].copy()
        o.lib.orc_env_reset(o.dp(self.EC), o.dp(x), o.dp(e), o.ip(ei), o.dp(self.pool[i, self.episode[i] % self.depth].copy()),
                            o.fp(self.obs[i]))
        if self.perturb:
            x *= 1.0 + self.perturb * self.rs.choice([-1.0, 1.0], 12)
        self.xs[:, i], self.es[:, i], self.eis[:, i] = x, e, ei
        self.episode[i] += 1

    def step(self, a):
        o = self.o
        o.lib.orc_env_step_batch(o.dp(self.P), o.dp(self.EC), o.dp(self.xs), o.dp(self.es), o.ip(self.eis),
                                 o.fp(np.ascontiguousarray(a)), o.fp(self.obs), o.dp(self.rew), o.ip(self.te), o.ip(self.tr),
                                 self.n, self.threads)
        term, trunc = self.te.astype(bool), self.tr.astype(bool)
        for i in np.nonzero(term | trunc)[0]:          # vec-env auto-reset: next record of env i's pool
            self._reset(i)
        return term.copy(), trunc.copy()


@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_env_4096_envs_520_steps_with_auto_reset_vs_oracle(oracle, precision):
    """Rewards, done flags, observations and states of 4096 distinct envs over 520 steps (every surviving env is truncated at
    step 500 and auto-resets; ~0.2 % of the env-steps carry a saturating action and crash some earlier).  An env leaves the
    comparison at the first done-flag mismatch (different episodes afterwards).  f64: exact flags, 1e-9.  mixed: the 1e-4
    gate wherever the reference itself holds 1e-5 under a 1e-7 perturbation of each freshly reset state (`d7`), and within
    30 x max(d7, 1e-6) everywhere."""
    steps, depth = 520, 6
    env = GpuRateVecEnv(N, "medium", 10.0, 0.02, "step", seed=977, precision=precision, sampling="parity", pool_depth=depth)
    pool = env.pool.cpu().numpy()                      # [N][depth][FD_NR] -- the records both sides reset from
    P, EC = AircraftParams().to_block(), samplers.env_consts("medium", 10.0, 0.02, "step")
    obs_g = env.reset().cpu().numpy()
    ref, per = _OracleFleet(oracle, pool, P, EC), _OracleFleet(oracle, pool, P, EC, perturb=1e-7)
    assert rel_err(obs_g, ref.obs).max() < 1e-6
    acts = _env_actions(N, steps, seed=5)
    alive = np.ones(N, bool)                           # envs whose done flags have agreed at every step so far
    alive7 = np.ones(N, bool)                          # ... between the oracle and its perturbed copy
    worst_state, worst_rew, worst_obs, d7 = np.zeros(N), np.zeros(N), np.zeros(N), np.zeros(N)
    n_done = 0
    for k in range(steps):
        a = acts[k]
        obs_t, _r, term, trunc = env.step_device(torch.as_tensor(a, device=env.device))
        te, tr = ref.step(a)
        te7, tr7 = per.step(a)
        n_done += int((te | tr).sum())
        tg, ug = term.cpu().numpy().astype(bool), trunc.cpu().numpy().astype(bool)
        alive &= (tg == te) & (ug == tr)
        alive7 &= (te7 == te) & (tr7 == tr)
        rg = env.rewards_full.to(torch.float64).cpu().numpy()
        worst_rew = np.maximum(worst_rew, np.where(alive, np.abs(rg - ref.rew) / np.maximum(np.abs(ref.rew), 1.0), 0.0))
        if k % 10 == 9 or k == steps - 1:
            sg = env.x.to(torch.float64).T.cpu().numpy()
            worst_state = np.maximum(worst_state, np.where(alive, rel_err(sg, ref.xs.T, STATE_ANGLE_COLS).max(1), 0.0))
            worst_obs = np.maximum(worst_obs, np.where(alive, rel_err(obs_t.cpu().numpy(), ref.obs, angle_cols=(11, 13)).max(1), 0.0))
            d7 = np.maximum(d7, np.where(alive7, rel_err(per.xs.T, ref.xs.T, STATE_ANGLE_COLS).max(1), 1.0))
    assert n_done > N                                  # every env was truncated at step 500 at the latest; some crashed earlier
    d7 = np.where(alive7, d7, 1.0)                     # the oracle lost its own perturbed copy: not reproducible at all
    lost = int((~alive).sum())
    well7 = d7 <= 1e-5
    ratio = worst_state / np.maximum(d7, 1e-6)
    qs = _record(precision, "env_520_steps_state", worst_state[alive], flag_mismatch_envs=lost, episode_ends=n_done,
                 n_d7_le_1e5=int(well7.sum()), max_where_d7_le_1e5=float(worst_state[well7 & alive].max()),
                 oracle_lost_its_perturbed_copy=int((~alive7).sum()), err_over_d7_max=float(ratio[alive].max()))
    qr = _record(precision, "env_520_steps_reward", worst_rew[alive])
    qo = _record(precision, "env_520_steps_obs", worst_obs[alive])
    print(f"\n[drift@scale] env {precision}: {N} envs x {steps} steps ({n_done} episode ends, auto-reset), "
          f"{lost} envs left the comparison at a done-flag mismatch ({int((~alive7).sum())} for the oracle vs its own 1e-7-perturbed copy)\n"
          f"   state  p50 {qs['p50']:.2e} p99 {qs['p99']:.2e} max {qs['max']:.2e}; d7<=1e-5 on {int(well7.sum())} envs, GPU max there "
          f"{qs['max_where_d7_le_1e5']:.2e}; err / max(d7, 1e-6) max {qs['err_over_d7_max']:.1f}\n"
          f"   reward p50 {qr['p50']:.2e} p99 {qr['p99']:.2e} max {qr['max']:.2e}\n"
          f"   obs    p50 {qo['p50']:.2e} p99 {qo['p99']:.2e} max {qo['max']:.2e}")
    if precision == "f64":
        assert lost == 0 and qs["max"] < 1e-9 and qr["max"] < 1e-9 and qo["max"] < 1e-6
    else:
        assert lost <= max(N // 500, 2 * int((~alive7).sum())), lost
        assert worst_state[well7 & alive].max() <= 1e-4 and ratio[alive].max() <= 30.0
        assert np.percentile(worst_rew[alive], 99) <= 1e-4 and np.percentile(worst_obs[alive], 99) <= 1e-4


@pytest.mark.parametrize("precision", ["mixed", "f32"])
def test_register_capped_env_kernel_equals_uncapped_on_the_same_rows(precision):
    """`rate_env_step_kernel<.., OCC2 = true>` (selected for more than one wave per SIMD, i.e. > 65 536 envs) against the
    uncapped build: 131 072 envs whose first 65 536 rows are the rows of a 65 536-env run -- same seeds, same actions,
    300 steps with in-kernel auto-reset (device sampling is keyed by (seed, env id, episode), so row i is row i)."""
    small, big = 65536, 131072
    a = GpuRateVecEnv(small, "medium", 2.0, 0.02, "step", seed=11, precision=precision, sampling="device")
    b = GpuRateVecEnv(big, "medium", 2.0, 0.02, "step", seed=11, precision=precision, sampling="device")
    oa, ob = a.reset(), b.reset()
    assert torch.equal(oa, ob[:small])
    g = torch.Generator(device=a.device).manual_seed(3)
    ends = 0
    for k in range(300):
        act = torch.cat([(torch.rand((big, 3), device=a.device, generator=g) - 0.5) * 0.8,
                         0.3 + 0.5 * torch.rand((big, 1), device=a.device, generator=g)], 1).contiguous()
        oa, ra, ta, ua = a.step_device(act[:small].contiguous())
        ob, rb, tb, ub = b.step_device(act)
        assert torch.equal(oa, ob[:small]) and torch.equal(ra, rb[:small]), (k, float((oa - ob[:small]).abs().max()), float((ra - rb[:small]).abs().max()))
        assert torch.equal(ta, tb[:small]) and torch.equal(ua, ub[:small]), k
        ends += int((ta | ua).sum())
    assert torch.equal(a.x, b.x[:, :small]) and torch.equal(a.e, b.e[:, :small]) and torch.equal(a.ei, b.ei[:, :small])
    assert ends > small                                # 2 s episodes: every env was truncated + auto-reset at least twice
