"""Host-side logic that needs no GPU: mission bookkeeping, command rows, sensor-noise blocks, callbacks and checkpoint
look-up, metric containers.  (The reference's counterparts: controllers/mission_planner.py, controllers/*_agent.py command
validation, interfaces/sensor.py:172-182, learned_controllers/utils/training_utils.py:72-156, eval/metrics.py:8-93.)"""
import numpy as np
import pytest

from hcrl_amd import layout as L
from hcrl_amd.flight_types import AircraftState, ControlCommand, ControlMode, Waypoint


def _state(n, e, d, t=0.0):
    return AircraftState(time=t, position=np.array([n, e, d], float))


def test_mission_planner_sequences_waypoints_and_reports():
    from hcrl_amd.mission import MissionPlanner, MissionState
    with pytest.raises(ValueError):
        MissionPlanner([])
    wps = [Waypoint.from_ned(100, 0, -100), Waypoint.from_ned(100, 100, -100), Waypoint.from_altitude(0, 100, 100, speed=15.0)]
    m = MissionPlanner(wps, acceptance_radius=15.0)
    assert m.state == MissionState.IDLE and m.get_current_waypoint() is None and not m.update(_state(100, 0, -100))
    m.start()
    assert m.is_active() and m.get_waypoint_command().mode == ControlMode.WAYPOINT
    assert not m.update(_state(0, 0, -100, 0.0)) and m.mission_start_time == 0.0
    assert abs(m.get_distance_to_current_waypoint(_state(0, 0, -100)) - 100.0) < 1e-12
    assert m.update(_state(90, 0, -100, 5.0)) and m.current_waypoint_index == 1 and m.waypoint_distances == [10.0]
    assert not m.update(_state(90, 0, -100, 5.1))                       # 100 m from the next one
    assert m.update(_state(100, 95, -90, 9.0))                          # 3-D distance: sqrt(25 + 100) < 15
    assert abs(m.get_progress_percentage() - 200.0 / 3) < 1e-9
    assert m.update(_state(5, 100, -100, 14.0)) and m.is_complete() and m.get_mission_duration() == 14.0
    assert m.get_waypoint_command() is None and not m.update(_state(5, 100, -100, 15.0))
    s = m.get_summary()
    assert s["waypoints_reached"] == 3 and s["state"] == "complete" and abs(s["total_distance_m"] - 200.0) < 1e-9
    assert s["waypoint_arrival_times"] == [5.0, 9.0, 14.0] and "3/3" not in repr(m)
    m.reset()
    assert m.state == MissionState.IDLE and m.waypoints_reached == 0 and m.mission_start_time is None
    m.start(); m.abort()
    assert m.is_aborted() and m.get_current_waypoint() is None


def test_command_rows_and_validation():
    from hcrl_amd.agents import command_row, LEVELS
    assert LEVELS[ControlMode.RATE] == L.FD_LEVEL_RATE and LEVELS[ControlMode.WAYPOINT] == L.FD_LEVEL_WAYPOINT
    r = command_row(ControlCommand(mode=ControlMode.RATE, roll_rate=0.1, pitch_rate=-0.2, yaw_rate=0.3, throttle=0.7))
    assert np.array_equal(r, [0.1, -0.2, 0.3, 0.7])
    a = command_row(ControlCommand(mode=ControlMode.ATTITUDE, roll_angle=0.2, pitch_angle=0.1))
    assert np.isnan(a[2]) and a[3] == 0.0                                 # no yaw command, no throttle
    h = command_row(ControlCommand(mode=ControlMode.HSA, heading=1.0, speed=20.0, altitude=120.0))
    assert np.array_equal(h, [1.0, 20.0, 120.0, 0.0])
    w = command_row(ControlCommand(mode=ControlMode.WAYPOINT, waypoint=Waypoint.from_altitude(10, 20, 100)))
    assert np.array_equal(w[:3], [10, 20, 100]) and np.isnan(w[3])       # speed None -> keep current airspeed
    for bad in (ControlCommand(mode=ControlMode.RATE, roll_rate=0.1), ControlCommand(mode=ControlMode.HSA, heading=0.0),
                ControlCommand(mode=ControlMode.WAYPOINT)):
        with pytest.raises(ValueError):
            command_row(bad)


def test_noise_block_defaults_and_overrides():
    from hcrl_amd.sensors import noise_block
    c = noise_block()
    assert c.shape == (L.FD_NSN,) and c[L.FD_SN_GPS_POS] == 1.0 and c[L.FD_SN_GYRO] == 0.01 and c[L.FD_SN_ENABLED] == 1.0
    assert c[L.FD_SN_GYRO_BIAS_WALK] == 0.0001 and c[L.FD_SN_ACCEL_BIAS_WALK] == 0.001      # sensor.py:230-231 constants
    c = noise_block({"enabled": False, "airspeed_stddev": 0.9})
    assert c[L.FD_SN_ENABLED] == 0.0 and c[L.FD_SN_AIRSPEED] == 0.9 and c[L.FD_SN_ALTITUDE] == 0.5


def test_callbacks_and_best_checkpoint_lookup(tmp_path):
    from hcrl_amd.training_utils import CallbackList, CheckpointCallback, ProgressLogger, find_best_checkpoint

    class FakeEnv:
        num_envs = 4

    class FakeModel:
        env, num_timesteps = FakeEnv(), 0
        saved = []

        def save(self, path):
            self.saved.append(path)
            open(path, "w").write("x")

    m = FakeModel()
    cb = CallbackList([CheckpointCallback(save_freq=10, save_path=str(tmp_path / "ck"), name_prefix="rc"),
                       ProgressLogger(str(tmp_path / "tb"))])
    for it in range(1, 8):                       # 8 vec-steps per iteration: checkpoints when n_calls crosses 10, 20, ...
        m.num_timesteps = it * 8 * 4
        cb(m, {"policy_loss": 0.1 * it})
    # n_calls = 8, 16, ..., 56: a checkpoint whenever n_calls // 10 grows -> at 16, 24, 32, 40 and 56 vec-steps
    assert [p.split("/")[-1] for p in m.saved] == [f"rc_{4 * c}_steps.pt" for c in (16, 24, 32, 40, 56)]
    assert len(open(tmp_path / "tb" / "progress.jsonl").read().splitlines()) == 7
    assert find_best_checkpoint(str(tmp_path)) is None
    np.savez(tmp_path / "evaluations.npz", timesteps=np.array([100, 200, 300]), results=np.array([[1.0, 3.0], [5.0, 5.0], [4.0, 2.0]]),
             ep_lengths=np.ones((3, 2)))
    assert find_best_checkpoint(str(tmp_path)) == (200, 5.0)


def test_metrics_container_and_aggregation():
    from hcrl_amd.eval_metrics import FIELD_ORDER, RateControlMetrics, aggregate_metrics
    assert len(FIELD_ORDER) == L.FD_NM and FIELD_ORDER[L.FD_M_RMSE] == "tracking_rmse" and FIELD_ORDER[L.FD_M_SUCCESS] == "success"
    a = RateControlMetrics.from_vector(np.arange(17.0))
    assert a.settling_time_roll == 0.0 and a.total_reward == 16.0 and a.success is True and list(a.to_dict()) == list(FIELD_ORDER)
    b = RateControlMetrics()
    agg = aggregate_metrics([a, b])
    assert agg.total_reward == 8.0 and agg.success == 0.5 and agg.tracking_rmse == 6.5


# ---- RLAgentInterface helpers + SurfaceAgent against the reference's outputs (tests/golden/agent_spaces.json) -------------
def _spaces_fixture():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "agent_spaces.json")))


def test_rl_agent_interface_spaces_and_observations_match_reference():
    """interfaces/agent.py:154-323 (the reference's tests: tests/test_interfaces.py:43-150)."""
    import torch
    from hcrl_amd.agents import RLAgentInterface, preprocess_observations
    from hcrl_amd.flight_types import AircraftState, ControlCommand, ControlMode

    with pytest.raises(TypeError):
        RLAgentInterface()                                   # abstract

    class LevelAgent(RLAgentInterface):
        def __init__(self, level):
            self.level = level

        def get_control_level(self):
            return self.level

        def reset(self, initial_state):
            self.last_state = initial_state

        def get_action(self, observation):
            return ControlCommand(mode=self.level)

    fx = _spaces_fixture()
    rows = np.array(fx["states"])
    states = [AircraftState(position=r[0:3], velocity=r[3:6], attitude=r[6:9], angular_rate=r[9:12], airspeed=r[12], altitude=r[13])
              for r in rows]
    for name, want in fx["levels"].items():
        ag = LevelAgent(ControlMode[name])
        for getter, key in ((ag.get_observation_space, "observation_space"), (ag.get_action_space, "action_space")):
            got = getter()
            assert set(got) == set(want[key]), (name, key)
            for k, v in want[key].items():
                g = got[k]
                assert (tuple(g) == tuple(v)) if k == "shape" else np.array_equal(np.asarray(g, dtype=object if isinstance(g, str) else None), np.asarray(v, dtype=object if isinstance(v, str) else None)), (name, key, k)
        for st, obs in zip(states, want["observations"]):
            assert np.array_equal(ag.preprocess_observation(st), np.array(obs)), name
        # the fleet form: all states at once
        batch = preprocess_observations(ControlMode[name], torch.as_tensor(rows[:, :12].T.copy()), torch.as_tensor(rows[:, 12]),
                                        torch.as_tensor(rows[:, 13]))
        assert np.array_equal(batch.numpy(), np.array(want["observations"])), name
        assert repr(ag) == want["repr"]
        with pytest.raises(NotImplementedError, match="does not support level switching") as e:
            ag.switch_control_level(ControlMode.HSA)
        assert str(e.value) == want["switch_error"]
        assert ag.update({"reward": 1.0}) is None and ag.save("x") is None and ag.load("x") is None
        ag.reset(states[0])
        assert ag.last_state is states[0] and ag.get_action(np.zeros(10)).mode == ControlMode[name]


def test_surface_agent_matches_reference():
    """controllers/surface_agent.py:8-103."""
    import torch
    from hcrl_amd.agents import SurfaceAgent
    from hcrl_amd.flight_types import ControlCommand, ControlMode
    for want in _spaces_fixture()["surface_agent"]:
        ag = SurfaceAgent(want["config"])
        assert repr(ag) == want["repr"] and ag.get_control_level().name == want["level"]
        lim = ag.limits()
        for case in want["cases"]:
            e, a, r, t = case["command"]
            out = ag.compute_action(ControlCommand(mode=ControlMode.SURFACE, elevator=e, aileron=a, rudder=r, throttle=t), None)
            assert [float(out.elevator), float(out.aileron), float(out.rudder), float(out.throttle)] == case["surfaces"]
            fleet = torch.tensor([case["command"]], dtype=torch.float64)
            assert torch.minimum(torch.maximum(fleet, lim[0]), lim[1])[0].tolist() == case["surfaces"]
        with pytest.raises(AssertionError, match="expects SURFACE mode"):
            ag.compute_action(ControlCommand(mode=ControlMode.RATE, roll_rate=0.0, pitch_rate=0.0, yaw_rate=0.0), None)
        ag.reset()


def test_recover_training_picks_copies_and_verifies(tmp_path, capsys):
    """learned_controllers/recover_training.py:18-152 over this trainer's files."""
    import torch
    from hcrl_amd import recover_training, sb3_zip
    from hcrl_amd.policy import RateLSTMPolicy
    ck, ev = tmp_path / "ckpt", tmp_path / "best"
    ck.mkdir(); ev.mkdir()
    pol = RateLSTMPolicy(use_lstm=False)
    for steps in (2000, 4000, 6000):
        torch.save({"policy": pol.state_dict(), "num_timesteps": steps}, ck / f"rate_controller_{steps}_steps.pt")
    (ck / "unrelated.pt").write_bytes(b"x")
    np.savez(ev / "evaluations.npz", timesteps=np.array([1500, 4100, 5900]), results=np.array([[1.0, 2.0], [9.0, 7.0], [3.0, 4.0]]),
             ep_lengths=np.zeros((3, 2)))
    assert [s for s, _ in recover_training.list_checkpoints(str(ck))] == [2000, 4000, 6000]
    out = tmp_path / "rec" / "recovered.pt"
    step = recover_training.main(["--output", str(out), "--checkpoint-dir", str(ck), "--eval-dir", str(ev)])
    assert step == 4000 and torch.load(out, weights_only=True)["num_timesteps"] == 4000
    text = capsys.readouterr().out
    assert "Best evaluation: step 4100 (reward: 8.00)" in text and "RECOVERY COMPLETE!" in text and "--resume" in text
    assert recover_training.main(["--checkpoint-step", "6000", "--output", str(out), "--checkpoint-dir", str(ck)]) == 6000
    with pytest.raises(SystemExit):                                       # a step that was never saved: lists what exists
        recover_training.main(["--checkpoint-step", "123", "--output", str(out), "--checkpoint-dir", str(ck)])
    assert "rate_controller_2000_steps.pt" in capsys.readouterr().out
    # no evaluation file: earliest checkpoint; an archive in SB3's layout verifies too
    sb3_zip.save_sb3_zip(ck / "rate_controller_1000_steps.zip", pol)
    assert recover_training.main(["--output", str(tmp_path / "r.zip"), "--checkpoint-dir", str(ck), "--eval-dir", str(tmp_path / "none")]) == 1000
    (ck / "rate_controller_500_steps.pt").write_bytes(b"not a checkpoint")
    with pytest.raises(SystemExit):                                       # copy succeeds, verification fails
        recover_training.main(["--checkpoint-step", "500", "--output", str(out), "--checkpoint-dir", str(ck)])


def test_aircraft_interface_contract():
    """interfaces/aircraft.py:20-175 defaults, exercised the way the reference's tests/test_interfaces.py:163-272 does."""
    from hcrl_amd.backend import AircraftInterface
    from hcrl_amd.flight_types import ControlSurfaces

    with pytest.raises(TypeError):
        AircraftInterface()

    class Craft(AircraftInterface):
        def __init__(self, backend_type="simulation"):
            self.backend_type, self.state, self.controls = backend_type, AircraftState(), ControlSurfaces()

        def step(self, dt):
            self.state.time += dt
            self.state.altitude += dt * 10.0
            return self.state

        def set_controls(self, surfaces):
            self.controls = surfaces

        def reset(self, initial_state=None):
            self.state = initial_state if initial_state is not None else AircraftState(altitude=100.0, airspeed=20.0)
            return self.state

        def get_state(self):
            return self.state

        def get_backend_type(self):
            return self.backend_type

    sim, hw, hil = Craft(), Craft("hardware"), Craft("hil")
    assert sim.reset().altitude == 100.0
    sim.set_controls(ControlSurfaces(elevator=0.1, throttle=0.7))
    assert sim.controls.elevator == 0.1 and sim.step(dt=0.1).altitude == 101.0 and sim.get_state().time == 0.1
    assert not sim.is_real_hardware() and hw.is_real_hardware() and hil.is_real_hardware()
    assert sim.supports_reset() and not hw.supports_reset()
    assert sim.get_dt_nominal() == 0.01 and sim.get_info() == {"backend_type": "simulation", "dt_nominal": 0.01}
    assert sim.close() is None and repr(sim) == "Craft(type=simulation)"


def test_bench_usable_cores_respects_affinity_and_quota(monkeypatch, tmp_path):
    """bench.py's CPU-baseline thread count: OpenMP's maximum capped by the affinity mask and the cgroup CPU quota."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n_aff = len(os.sched_getaffinity(0))
    assert 1 <= bench.usable_cores(10 ** 6) <= n_aff and bench.usable_cores(1) == 1
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            import io
            return io.StringIO("200000 100000\n")                    # a quota of two cores
        return real_open(path, *a, **k)
    monkeypatch.setattr("builtins.open", fake_open)
    assert bench.usable_cores(10 ** 6) == min(2, n_aff)
    assert isinstance(bench.cpu_model(), str)
