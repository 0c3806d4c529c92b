#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

Runs only where /root/reference exists (the build container); the GPU box never sees it.
Needs `make -C oracle ref` first (the reference's own C++ PID compiled from its sources into
oracle/_ref/, because `import controllers` requires `aircraft_controls_bindings`).

What is imported from the reference (its real code, unmodified, read-only):
  simulation.simplified_6dof / simulation.simulation_backend        (physics)
  controllers.*  (types, rate/attitude/hsa/waypoint agents, mission planner, config_loader)
  learned_controllers/envs/rewards.py, learned_controllers/data/generators.py  -- loaded BY FILE PATH,
      because `learned_controllers.envs.__init__` imports gymnasium, which is absent from this image and
      stays absent (no stand-in is written for it).

What is therefore NOT a direct reference output: the RateControlEnv glue (rate_env.py:151-300,437-460).
`env_*` fixtures are produced by driving the reference's backend + reward + generator objects in the
order rate_env.py does; the result is pinned end-to-end against the values recorded in SURVEY.md §8a
(cmd, obs0[9:14], termination at step 150, return -15.188049), asserted below.

Fixtures are DATA (inputs + expected outputs) only.
"""
import importlib.util
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
sys.path.insert(0, REF)

import aircraft_controls_bindings as acb  # noqa: E402  (reference C++ PID, built by oracle/Makefile)
from controllers.types import (  # noqa: E402
    AircraftState, ControlCommand, ControlMode, ControlSurfaces, ControllerConfig, Waypoint)
from controllers.rate_agent import RateAgent  # noqa: E402
from controllers.attitude_agent import AttitudeAgent  # noqa: E402
from controllers.hsa_agent import HSAAgent  # noqa: E402
from controllers.waypoint_agent import WaypointAgent  # noqa: E402
from controllers.mission_planner import MissionPlanner  # noqa: E402
from controllers.config_loader import load_controller_config, load_mission_config  # noqa: E402
from simulation.simplified_6dof import Simplified6DOF, AircraftParams  # noqa: E402
from simulation.simulation_backend import SimulationAircraftBackend  # noqa: E402


def _load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


rewards_mod = _load_by_path("_ref_rewards", "learned_controllers/envs/rewards.py")
gen_mod = _load_by_path("_ref_generators", "learned_controllers/data/generators.py")


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


def state_vec(sim):
    return np.array(sim._state, dtype=np.float64).copy()


def derived(st):
    return np.array([st.airspeed, st.altitude, st.ground_speed, st.heading], dtype=np.float64)


def mk_state(x):
    return AircraftState(time=0.0, position=np.array(x[0:3]), velocity=np.array(x[3:6]),
                         attitude=np.array(x[6:9]), angular_rate=np.array(x[9:12]),
                         airspeed=float(np.linalg.norm(x[3:6])), altitude=float(-x[2]))


# --------------------------------------------------------------------------------------------
# cfg 2 : open-loop physics (SURVEY §8d cfg 2 input recipe)
# --------------------------------------------------------------------------------------------
def sample_cfg2(n, seed=0):
    """RandomState(seed) draws in the order of FlightEnvelopeSampler.sample, then controls."""
    rs = np.random.RandomState(seed)
    x0 = np.zeros((n, 12))
    ctrl = np.zeros((n, 4))  # [elevator, aileron, rudder, throttle]
    for i in range(n):
        V = rs.uniform(15.0, 30.0)
        h = rs.uniform(50.0, 200.0)
        roll = rs.uniform(np.radians(-15), np.radians(15))
        pitch = rs.uniform(np.radians(-15), np.radians(15))
        yaw = rs.uniform(0, 2 * np.pi)
        p, q, r = rs.uniform(-0.1, 0.1), rs.uniform(-0.1, 0.1), rs.uniform(-0.1, 0.1)
        x0[i] = [0, 0, -h, V, 0, 0, roll, pitch, yaw, p, q, r]
    for i in range(n):
        ctrl[i, 0] = rs.uniform(-0.3, 0.3)
        ctrl[i, 1] = rs.uniform(-0.3, 0.3)
        ctrl[i, 2] = rs.uniform(-0.3, 0.3)
        ctrl[i, 3] = rs.uniform(0.3, 0.9)
    return x0, ctrl


def gen_open_loop():
    print("cfg2 open loop")
    n = 32
    x0, ctrl = sample_cfg2(n, seed=0)
    for tag, dt, steps, every, params in (
            ("dt0p001", 0.001, 1000, 20, None),
            ("dt0p01", 0.01, 1000, 20, None),
            ("cessna_dt0p01", 0.01, 400, 20, "cessna")):
        traj = np.zeros((n, steps // every + 1, 12))
        der = np.zeros((n, steps // every + 1, 4))
        for i in range(n):
            if params == "cessna":
                sim = Simplified6DOF(SimulationAircraftBackend({'aircraft_type': 'cessna'})._physics.params)
            else:
                sim = Simplified6DOF(AircraftParams())
            sim.reset(mk_state(x0[i]))
            sim.set_controls(ControlSurfaces(elevator=ctrl[i, 0], aileron=ctrl[i, 1],
                                             rudder=ctrl[i, 2], throttle=ctrl[i, 3]))
            traj[i, 0] = state_vec(sim)
            der[i, 0] = derived(sim.get_state())
            for k in range(1, steps + 1):
                st = sim.step(dt)
                if k % every == 0:
                    traj[i, k // every] = state_vec(sim)
                    der[i, k // every] = derived(st)
        save(f"open_loop_{tag}.npz", x0=x0, ctrl=ctrl, dt=dt, steps=steps, every=every,
             traj=traj, derived=der)

    # backend sub-stepping (simulation_backend.py:82-101), as the env uses it: step(0.02) -> 20 x 1 ms
    steps, every = 200, 10
    traj = np.zeros((n, steps // every + 1, 12))
    for i in range(n):
        be = SimulationAircraftBackend({'aircraft_type': 'rc_plane', 'dt_physics': 0.001})
        be.reset(mk_state(x0[i]))
        be.set_controls(ControlSurfaces(elevator=ctrl[i, 0], aileron=ctrl[i, 1],
                                        rudder=ctrl[i, 2], throttle=ctrl[i, 3]))
        traj[i, 0] = state_vec(be._physics)
        for k in range(1, steps + 1):
            be.step(0.02)
            if k % every == 0:
                traj[i, k // every] = state_vec(be._physics)
    save("open_loop_backend_dt0p02.npz", x0=x0, ctrl=ctrl, dt=0.02, dt_physics=0.001,
         steps=steps, every=every, traj=traj)
    # int(dt/dt_physics) truncation cases (simulation_backend.py:95)
    cases = [(0.03, 0.001), (0.01, 0.001), (0.0005, 0.001), (0.07, 0.01), (0.02, 0.003), (0.1, 0.001)]
    nsub = np.array([max(1, int(dt / dtp)) for dt, dtp in cases])
    save("substep_counts.npz", cases=np.array(cases), nsub=nsub)


def gen_stress():
    """Wild ICs/controls that hit every clamp branch (simplified_6dof.py:258-291, 364-376, 463, 485-501)."""
    print("stress")
    rs = np.random.RandomState(7)
    n = 48
    x0 = np.zeros((n, 12))
    ctrl = np.zeros((n, 4))
    for i in range(n):
        x0[i, 0:2] = rs.uniform(-50, 50, 2)
        x0[i, 2] = -rs.uniform(0.05, 30.0)                      # near the ground
        x0[i, 3] = rs.uniform(-5, 60) if i % 3 else rs.uniform(-1e-7, 1e-7)  # |u|<1e-6 branch
        x0[i, 4] = rs.uniform(-20, 20)
        x0[i, 5] = rs.uniform(-20, 40)
        x0[i, 6] = rs.uniform(-np.pi, np.pi)
        x0[i, 7] = rs.uniform(-1.6, 1.6)                        # beyond the 85 deg clamp
        x0[i, 8] = rs.uniform(-7, 7)                            # unwrapped yaw
        x0[i, 9:12] = rs.uniform(-8, 8, 3)                      # beyond 360 deg/s
        ctrl[i, 0:3] = rs.uniform(-1.5, 1.5, 3)                 # beyond [-1,1]
        ctrl[i, 3] = rs.uniform(-0.2, 1.3)
    x0[0, 3:6] = 0.0                                            # zero airspeed
    x0[1, 3:6] = [120.0, -130.0, 110.0]                         # beyond velocity clamp
    dt, steps, every = 0.01, 300, 10
    traj = np.zeros((n, steps // every + 1, 12))
    der = np.zeros((n, steps // every + 1, 4))
    import logging
    logging.disable(logging.CRITICAL)
    for i in range(n):
        sim = Simplified6DOF(AircraftParams())
        sim.reset(mk_state(x0[i]))
        sim.set_controls(ControlSurfaces(elevator=ctrl[i, 0], aileron=ctrl[i, 1],
                                         rudder=ctrl[i, 2], throttle=ctrl[i, 3]))
        traj[i, 0] = state_vec(sim)
        der[i, 0] = derived(sim.get_state())
        for k in range(1, steps + 1):
            st = sim.step(dt)
            if k % every == 0:
                traj[i, k // every] = state_vec(sim)
                der[i, k // every] = derived(st)
    logging.disable(logging.NOTSET)
    save("stress_dt0p01.npz", x0=x0, ctrl=ctrl, dt=dt, steps=steps, every=every, traj=traj, derived=der)

    # single-evaluation fixtures of _dynamics (simplified_6dof.py:333-503)
    m = 256
    xs = np.zeros((m, 12))
    cs = np.zeros((m, 4))
    ds = np.zeros((m, 12))
    sim = Simplified6DOF(AircraftParams())
    for i in range(m):
        xs[i, 0:3] = rs.uniform(-100, 100, 3)
        xs[i, 3] = rs.uniform(-10, 60) if i % 5 else rs.uniform(-1e-6, 1e-6)
        xs[i, 4:6] = rs.uniform(-30, 30, 2)
        xs[i, 6:9] = rs.uniform(-4, 4, 3)
        xs[i, 9:12] = rs.uniform(-7, 7, 3)
        cs[i, 0:3] = rs.uniform(-1, 1, 3)
        cs[i, 3] = rs.uniform(0, 1)
        sim.set_controls(ControlSurfaces(elevator=cs[i, 0], aileron=cs[i, 1], rudder=cs[i, 2],
                                         throttle=cs[i, 3]))
        ds[i] = sim._dynamics(xs[i], sim._controls)
    save("dynamics_eval.npz", x=xs, ctrl=cs, xdot=ds)


# --------------------------------------------------------------------------------------------
# PID (reference C++ via its own pybind module) -- bit-exact fp32 sequences
# --------------------------------------------------------------------------------------------
def gen_pid():
    print("pid")
    rs = np.random.RandomState(11)
    n_seq, T = 24, 200
    cfgs = np.zeros((n_seq, 8), dtype=np.float32)  # kp ki kd omin omax imin imax alpha
    sp = rs.uniform(-2, 2, (n_seq, T)).astype(np.float32)
    ms = rs.uniform(-2, 2, (n_seq, T)).astype(np.float32)
    dts = np.full((n_seq, T), 0.01, dtype=np.float32)
    dts[1] = 0.001
    dts[2] = 0.02
    dts[3, ::7] = 0.0          # dt <= 1e-6 branch
    dts[4] = rs.uniform(1e-4, 0.05, T).astype(np.float32)
    out = np.zeros((n_seq, T), dtype=np.float32)
    integ = np.zeros((n_seq, T), dtype=np.float32)
    deriv = np.zeros((n_seq, T), dtype=np.float32)
    for s in range(n_seq):
        c = acb.PIDConfig()
        if s % 4 != 3:
            c.gains = acb.PIDGains(float(rs.uniform(0, 8)), float(rs.uniform(0, 2)), float(rs.uniform(0, 0.3)))
            lim = float(rs.uniform(0.2, 3))
            c.output_min, c.output_max = -lim, lim
            il = float(rs.uniform(0.5, 25))
            c.integral_min, c.integral_max = -il, il
            if s % 5 == 0:
                c.derivative_filter_alpha = float(rs.uniform(0.05, 1.0))
        else:
            c.gains = acb.PIDGains(1.3, 0.4, 0.012)
            c.integral_min, c.integral_max = -25.0, 25.0
        cfgs[s] = [c.gains.kp, c.gains.ki, c.gains.kd, c.output_min, c.output_max,
                   c.integral_min, c.integral_max, c.derivative_filter_alpha]
        pid = acb.PIDController(c)
        for t in range(T):
            out[s, t] = pid.compute(float(sp[s, t]), float(ms[s, t]), float(dts[s, t]))
            integ[s, t] = pid.get_integral()
            deriv[s, t] = pid.get_derivative()
    save("pid_sequences.npz", cfg=cfgs, setpoint=sp, measurement=ms, dt=dts, output=out,
         integral=integ, derivative=deriv)


# --------------------------------------------------------------------------------------------
# agents, one level at a time, on random state/command sequences (stateful PIDs run in sequence)
# --------------------------------------------------------------------------------------------
def rand_states(rs, T):
    xs = np.zeros((T, 12))
    xs[:, 0:2] = rs.uniform(-300, 300, (T, 2))
    xs[:, 2] = -rs.uniform(20, 200, T)
    xs[:, 3] = rs.uniform(10, 30, T)
    xs[:, 4:6] = rs.uniform(-3, 3, (T, 2))
    xs[:, 6] = rs.uniform(-1.0, 1.0, T)
    xs[:, 7] = rs.uniform(-0.5, 0.5, T)
    xs[:, 8] = rs.uniform(-np.pi, np.pi, T)
    xs[:, 9:12] = rs.uniform(-2, 2, (T, 3))
    return xs


def full_state(x):
    """AircraftState with the derived fields as get_state computes them (simplified_6dof.py:295-331)."""
    sim = Simplified6DOF(AircraftParams())
    sim.reset(mk_state(x))
    return sim.get_state()


def surf(s):
    return [float(s.elevator), float(s.aileron), float(s.rudder), float(s.throttle)]


def gen_agents():
    print("agents")
    rs = np.random.RandomState(21)
    T = 160
    cfg = ControllerConfig()
    flight_cfg = load_controller_config("cascaded_pid.yaml")
    xs = rand_states(rs, T)
    states = [full_state(x) for x in xs]
    der = np.array([derived(s) for s in states])

    # rate
    cmd_rate = rs.uniform(-4, 4, (T, 3))
    thr = rs.uniform(-0.1, 1.1, T)
    out_rate = np.zeros((2, T, 4))
    for j, dt in enumerate((0.01, None)):
        ag = RateAgent(cfg)
        for t in range(T):
            c = ControlCommand(mode=ControlMode.RATE, roll_rate=cmd_rate[t, 0], pitch_rate=cmd_rate[t, 1],
                               yaw_rate=cmd_rate[t, 2], throttle=thr[t])
            out_rate[j, t] = surf(ag.compute_action(c, states[t], dt=dt))
    # attitude
    cmd_att = np.stack([rs.uniform(-0.8, 0.8, T), rs.uniform(-0.8, 0.8, T), rs.uniform(-7, 7, T)], 1)
    out_att = np.zeros((T, 4))
    ag = AttitudeAgent(cfg)
    for t in range(T):
        c = ControlCommand(mode=ControlMode.ATTITUDE, roll_angle=cmd_att[t, 0], pitch_angle=cmd_att[t, 1],
                           yaw_angle=cmd_att[t, 2], throttle=thr[t])
        out_att[t] = surf(ag.compute_action(c, states[t], 0.01))
    # hsa (with and without the YAML flight config)
    cmd_hsa = np.stack([rs.uniform(-4, 4, T), rs.uniform(10, 25, T), rs.uniform(50, 150, T)], 1)
    out_hsa = np.zeros((2, T, 4))
    for j, fc in enumerate((flight_cfg, None)):
        ag = HSAAgent(cfg, fc)
        for t in range(T):
            c = ControlCommand(mode=ControlMode.HSA, heading=cmd_hsa[t, 0], speed=cmd_hsa[t, 1],
                               altitude=cmd_hsa[t, 2])
            out_hsa[j, t] = surf(ag.compute_action(c, states[t], 0.01))
    # waypoint (PP, LOS, default)
    wps = np.stack([rs.uniform(-300, 300, T), rs.uniform(-300, 300, T), rs.uniform(50, 150, T),
                    rs.uniform(12, 20, T)], 1)
    wps[::9, 0:2] = xs[::9, 0:2] + rs.uniform(-0.5, 0.5, (len(xs[::9]), 2))   # <1 m branch
    wps[1::9, 0:2] = xs[1::9, 0:2] + rs.uniform(-40, 40, (len(xs[1::9]), 2))  # proximity / slow-down branches
    out_wp = np.zeros((3, T, 4))
    for j, g in enumerate(("PP", "LOS", "XX")):
        ag = WaypointAgent(cfg, guidance_type=g, flight_config=flight_cfg)
        for t in range(T):
            wp = Waypoint.from_altitude(wps[t, 0], wps[t, 1], wps[t, 2], speed=wps[t, 3])
            c = ControlCommand(mode=ControlMode.WAYPOINT, waypoint=wp)
            out_wp[j, t] = surf(ag.compute_action(c, states[t], 0.01))
    save("agents.npz", x=xs, derived=der, cmd_rate=cmd_rate, throttle=thr, out_rate=out_rate,
         cmd_att=cmd_att, out_att=out_att, cmd_hsa=cmd_hsa, out_hsa=out_hsa, wps=wps, out_wp=out_wp)


# --------------------------------------------------------------------------------------------
# cfg 1 : examples/01_hello_controls.py:55-121 loop, restated here over the reference classes
# --------------------------------------------------------------------------------------------
def gen_cfg1():
    print("cfg1 rate PID")
    backend = SimulationAircraftBackend({'aircraft_type': 'rc_plane'})
    agent = RateAgent(ControllerConfig())
    x0 = np.array([0, 0, -100.0, 20.0, 0, 0, 0, 0, 0, 0, 0, 0])
    backend.reset(mk_state(x0))
    dt, steps = 0.01, 500
    cmd = ControlCommand(mode=ControlMode.RATE, roll_rate=np.radians(30), pitch_rate=0.0, yaw_rate=0.0,
                         throttle=0.7)
    traj = np.zeros((steps, 12))
    surfaces = np.zeros((steps, 4))
    state = backend.get_state()
    for i in range(steps):
        s = agent.compute_action(cmd, state, dt=dt)
        backend.set_controls(s)
        backend.step(dt)
        state = backend.get_state()
        traj[i] = state_vec(backend._physics)
        surfaces[i] = surf(s)
    # SURVEY §8a recorded values
    assert abs(np.degrees(traj[0, 9]) - 60.821838) < 1e-5 and abs(surfaces[0, 1] - 0.745605) < 1e-5
    assert abs(np.degrees(traj[100, 9]) - 65.718980) < 1e-5
    assert abs(np.degrees(traj[499, 9]) + 129.185359) < 1e-4
    save("cfg1_rate_pid.npz", x0=x0, dt=dt, steps=steps, cmd=np.array([np.radians(30), 0, 0, 0.7]),
         traj=traj, surfaces=surfaces)


# --------------------------------------------------------------------------------------------
# cfg 3 : examples/03_waypoint_square_demo.py:60-215 loop, restated over the reference classes
# --------------------------------------------------------------------------------------------
def gen_cfg3():
    print("cfg3 waypoint square")
    mc = load_mission_config("square_pattern.yaml")
    fc = load_controller_config("cascaded_pid.yaml")
    S, ALT, SPD, DT = mc.pattern_size, mc.altitude, mc.speed, mc.dt
    wps_arr = np.array([[0, 0, ALT, SPD], [S, 0, ALT, SPD], [S, S, ALT, SPD], [0, S, ALT, SPD], [0, 0, ALT, SPD]])
    waypoints = [Waypoint.from_altitude(w[0], w[1], w[2], speed=w[3]) for w in wps_arr]
    sim = Simplified6DOF(AircraftParams())
    x0 = np.array([0, 0, -ALT, SPD, 0, 0, 0, 0, 0, 0, 0, 0], dtype=np.float64)
    sim.reset(AircraftState(position=x0[0:3].copy(), velocity=x0[3:6].copy(), attitude=np.zeros(3),
                            angular_rate=np.zeros(3), airspeed=SPD, altitude=ALT))
    cc = ControllerConfig()
    cc.waypoint_acceptance_radius = fc.guidance.acceptance_radius
    agent = WaypointAgent(cc, guidance_type=mc.guidance, flight_config=fc)
    mission = MissionPlanner(waypoints, acceptance_radius=fc.guidance.acceptance_radius)
    mission.start()
    t, k, last = 0.0, 0, 0
    traj, surfs, idxs, events = [], [], [], []
    while t < mc.max_duration:
        state = sim.get_state()
        if k % 10 == 0:
            traj.append(state_vec(sim))
            idxs.append(mission.current_waypoint_index)
        mission.update(state)
        if mission.current_waypoint_index != last:
            wp = waypoints[last]
            err = np.sqrt((state.north - wp.north) ** 2 + (state.east - wp.east) ** 2 +
                          (state.altitude - wp.altitude) ** 2)
            events.append([k, t, last, err])
            last = mission.current_waypoint_index
        if mission.is_complete():
            break
        cmd = mission.get_waypoint_command()
        s = agent.compute_action(cmd, state, DT)
        if k % 10 == 0:
            surfs.append(surf(s))
        sim.set_controls(s)
        sim.step(DT)
        t += DT
        k += 1
    final = state_vec(sim)
    events = np.array(events)
    print("   steps", k, "events t:", events[:, 1], "err:", events[:, 3])
    assert k == 5079 or k == 5080, k
    assert np.allclose(events[:, 1], [0.0, 17.21, 31.35, 41.06, 50.79], atol=0.006)
    assert np.allclose(events[:, 3], [0.0, 39.899, 39.962, 39.841, 39.941], atol=2e-3)
    save("cfg3_waypoint_square.npz", x0=x0, dt=DT, waypoints=wps_arr, radius=fc.guidance.acceptance_radius,
         n_steps=k, traj=np.array(traj), surfaces=np.array(surfs), wp_index=np.array(idxs),
         events=events, final=final)


# --------------------------------------------------------------------------------------------
# samplers (learned_controllers/data/generators.py) and the env driven over reference components
# --------------------------------------------------------------------------------------------
def gen_samplers():
    print("samplers")
    out = {}
    for seed in (0, 42, 43, 1234):
        es = gen_mod.FlightEnvelopeSampler(rng_seed=seed)
        ics = []
        for _ in range(6):
            d = es.sample()
            ics.append([d["airspeed"], d["altitude"], *d["attitude"], *d["angular_rate"]])
        out[f"ic_seed{seed}"] = np.array(ics)
        for diff in ("easy", "medium", "hard"):
            env_rng = np.random.RandomState(seed)
            cg = gen_mod.RateCommandGenerator(difficulty=diff, rng_seed=seed)
            cmds = []
            for _ in range(6):
                k = env_rng.choice([1, 2, 3])
                c, _d = cg.generate_step_command(num_axes=k)
                cmds.append(c)
            out[f"step_{diff}_seed{seed}"] = np.array(cmds)
            cg = gen_mod.RateCommandGenerator(difficulty=diff, rng_seed=seed)
            ramps = []
            for _ in range(4):
                s, e, _d = cg.generate_ramp_command()
                ramps.append(e)
            out[f"ramp_{diff}_seed{seed}"] = np.array(ramps)
            cg = gen_mod.RateCommandGenerator(difficulty=diff, rng_seed=seed)
            sines = []
            for _ in range(4):
                f, a, _d = cg.generate_sine_command()
                sines.append([f, *a])
            out[f"sine_{diff}_seed{seed}"] = np.array(sines)
            cg = gen_mod.RateCommandGenerator(difficulty=diff, rng_seed=seed)
            rw = []
            for _ in range(8):
                d_, _d = cg.generate_random_walk(dt=0.02)
                rw.append(d_)
            out[f"rw_{diff}_seed{seed}"] = np.array(rw)
    save("samplers.npz", **out)


class RefComposedRateEnv:
    """RateControlEnv's step/reset order (rate_env.py:151-300,342-460) over the REFERENCE's own backend,
    reward and generator objects.  Only the glue below is ours; it is pinned against SURVEY §8a."""

    def __init__(self, difficulty="medium", episode_length=10.0, dt=0.02, command_type="step", rng_seed=None):
        self.dt, self.command_type = dt, command_type
        self.rng = np.random.RandomState(rng_seed)
        self.sim = SimulationAircraftBackend({'aircraft_type': 'rc_plane', 'dt_physics': 0.001})
        self.cmd_generator = gen_mod.RateCommandGenerator(difficulty=difficulty, rng_seed=rng_seed)
        self.envelope_sampler = gen_mod.FlightEnvelopeSampler(rng_seed=rng_seed)
        self.reward_tracker = rewards_mod.RateTrackingReward()
        self.settle_bonus = rewards_mod.SettlingTimeBonus()
        self.max_steps = int(episode_length / dt)
        self.max_rates = np.array([self.cmd_generator.max_roll_rate, self.cmd_generator.max_pitch_rate,
                                   self.cmd_generator.max_yaw_rate])
        self.schedule = None
        self.sine = None

    def reset(self, seed=None):
        if seed is not None:
            self.rng = np.random.RandomState(seed)
        ic = self.envelope_sampler.sample()
        self.sim.reset(AircraftState(time=0.0, position=np.array([0.0, 0.0, -ic["altitude"]]),
                                     velocity=np.array([ic["airspeed"], 0.0, 0.0]), attitude=ic["attitude"],
                                     angular_rate=ic["angular_rate"], airspeed=ic["airspeed"],
                                     altitude=ic["altitude"]))
        ct = self.command_type
        if ct == "step":
            k = self.rng.choice([1, 2, 3])
            self.rate_command, _ = self.cmd_generator.generate_step_command(num_axes=k)
            self.schedule = None
        elif ct == "ramp":
            s, e, _ = self.cmd_generator.generate_ramp_command()
            self.schedule = ("ramp", s, e)
            self.rate_command = s
        elif ct == "sine":
            f, a, _ = self.cmd_generator.generate_sine_command()
            self.sine = (f, a)
            self.rate_command = np.zeros(3)
        elif ct == "random":
            self.rate_command = np.zeros(3)
            self.schedule = ("random_walk",)
        else:
            self.rate_command = np.zeros(3)
        self.t, self.k = 0.0, 0
        self.prev_action = np.array([0.0, 0.0, 0.0, 0.5])
        self.reward_tracker.reset()
        self.settle_bonus.reset()
        self.state = self.sim.get_state()
        return self._obs()

    def _obs(self):
        s, c = self.state, self.rate_command
        return np.array([s.p, s.q, s.r, c[0], c[1], c[2], c[0] - s.p, c[1] - s.q, c[2] - s.r,
                         s.airspeed, s.altitude, s.roll, s.pitch, s.yaw, *self.prev_action], dtype=np.float32)

    def step(self, action):
        action = np.clip(action, np.array([-1.0, -1.0, -1.0, 0.0]), np.array([1.0, 1.0, 1.0, 1.0]))
        self.sim.set_controls(ControlSurfaces(aileron=float(action[0]), elevator=float(action[1]),
                                              rudder=float(action[2]), throttle=float(action[3])))
        s = self.state = self.sim.step(self.dt)
        self.t += self.dt
        self.k += 1
        if self.schedule is not None:
            if self.schedule[0] == "ramp":
                if self.t < 3.0:
                    a = self.t / 3.0
                    self.rate_command = (1 - a) * self.schedule[1] + a * self.schedule[2]
                else:
                    self.rate_command = self.schedule[2]
            else:
                d, _ = self.cmd_generator.generate_random_walk(dt=self.dt)
                self.rate_command += d
                np.clip(self.rate_command, -self.max_rates, self.max_rates, out=self.rate_command)
        elif self.sine is not None:
            self.rate_command = self.sine[1] * np.sin(2 * np.pi * self.sine[0] * self.t)
        c = self.rate_command
        pe, qe, re = c[0] - s.p, c[1] - s.q, c[2] - s.r
        reward, comps = self.reward_tracker.compute(pe, qe, re, action, self.prev_action, s.airspeed,
                                                    s.altitude, s.roll, s.pitch)
        settle = self.settle_bonus.compute(pe, qe, re, c[0], c[1], c[2], self.dt)
        reward += settle
        # what rate_env.py:276-279,421-433 hands out as info["reward_components"] / accumulates in episode_rewards
        self.last_components = np.array([comps["tracking"], comps["smoothness"], comps["stability"], comps["oscillation"],
                                         comps["survival"], settle])
        self.prev_action = action.copy()
        terminated = bool(s.altitude < 5.0 or abs(s.roll) > np.radians(120) or abs(s.pitch) > np.radians(80)
                          or s.airspeed < 8.0)
        truncated = self.k >= self.max_steps
        if terminated and not truncated:
            reward += -100.0
        return self._obs(), float(reward), terminated, truncated, bool(self.settle_bonus.is_settled)


def run_env_episode(env, policy, reset_seed, max_steps=600):
    obs = [env.reset(seed=reset_seed)]
    cmd0 = env.rate_command.copy()
    x0 = state_vec(env.sim._physics)
    acts, rews, flags, cmds, xs, comps = [], [], [], [], [], []
    pid = None
    for k in range(max_steps):
        if isinstance(policy, np.ndarray):
            a = policy[k] if policy.ndim == 2 else policy
        else:  # "pid": pid_demonstrations.py:47-77
            if pid is None:
                pid = RateAgent(ControllerConfig())
            c = ControlCommand(mode=ControlMode.RATE, roll_rate=env.rate_command[0],
                               pitch_rate=env.rate_command[1], yaw_rate=env.rate_command[2], throttle=0.6)
            s = pid.compute_action(c, env.sim.get_state(), dt=0.02)
            a = np.array([s.aileron, s.elevator, s.rudder, s.throttle], dtype=np.float32)
        a = np.asarray(a, dtype=np.float32)
        o, r, term, trunc, settled = env.step(a)
        obs.append(o)
        acts.append(a)
        rews.append(r)
        flags.append([term, trunc, settled])
        cmds.append(env.rate_command.copy())
        xs.append(state_vec(env.sim._physics))
        comps.append(env.last_components.copy())
        if term or trunc:
            break
    # reward_components [T][6]: the reference tracker's tracking / smoothness / stability / oscillation / survival + the settle bonus
    return dict(x0=x0, cmd0=cmd0, obs=np.array(obs), actions=np.array(acts), rewards=np.array(rews),
                flags=np.array(flags), cmds=np.array(cmds), states=np.array(xs), reward_components=np.array(comps))


def gen_env():
    print("env (reference components, composed)")
    # the SURVEY §8a episode
    env = RefComposedRateEnv(difficulty="easy", episode_length=10, dt=0.02, command_type="step", rng_seed=42)
    ep = run_env_episode(env, np.array([0.1, 0.0, 0.0, 0.6], dtype=np.float32), reset_seed=42)
    print("   cmd", ep["cmd0"], "obs0[9:14]", ep["obs"][0, 9:14], "len", len(ep["rewards"]), "ret",
          ep["rewards"].sum())
    assert np.allclose(ep["cmd0"], [-0.90996233, -0.79713236, -0.34282152], atol=1e-8)
    assert np.allclose(ep["obs"][0, 9:14], [20.6181, 192.6071, 0.1214717, 0.05165746, 0.980294], rtol=1e-6)
    assert len(ep["rewards"]) == 150 and abs(ep["rewards"].sum() + 15.188049) < 1e-5
    assert ep["flags"][-1, 0] and not ep["flags"][-1, 1]
    save("env_easy_step_seed42_const.npz", **ep)

    rs = np.random.RandomState(5)
    # random (smooth-ish) actions, medium/step, seed 7; PID actions on easy/medium/hard; time-varying commands
    env = RefComposedRateEnv(difficulty="medium", command_type="step", rng_seed=7)
    a = np.clip(np.cumsum(rs.normal(0, 0.05, (600, 4)), 0) + [0, 0, 0, 0.6], -1.2, 1.2).astype(np.float32)
    save("env_medium_step_seed7_rand.npz", **run_env_episode(env, a, reset_seed=7))
    for diff, ct, seed in (("easy", "step", 3), ("medium", "step", 11), ("hard", "random", 5),
                           ("medium", "ramp", 9), ("medium", "sine", 13)):
        env = RefComposedRateEnv(difficulty=diff, command_type=ct, rng_seed=seed)
        eps = [run_env_episode(env, "pid", reset_seed=seed)]
        # second episode on the same env object WITHOUT reseeding: the auto-reset path (sampler streams go on)
        eps.append(run_env_episode(env, "pid", reset_seed=None))
        d = {}
        for j, e in enumerate(eps):
            for k_, v in e.items():
                d[f"ep{j}_{k_}"] = v
        save(f"env_{diff}_{ct}_seed{seed}_pid.npz", **d)

    # residual env (residual_rate_env.py:99-157) composed over the same reference objects: PID + 0.3 * residual
    env = RefComposedRateEnv(difficulty="medium", command_type="step", rng_seed=17)
    obs = [env.reset(seed=17)]
    pid = RateAgent(ControllerConfig())
    res = np.clip(np.random.RandomState(23).normal(0, 0.4, (500, 4)), -1, 1).astype(np.float32)   # own stream: later fixtures keep theirs
    scale = 0.3
    rews, flags, pid_acts, comb_acts = [], [], [], []
    for k in range(500):
        c = ControlCommand(mode=ControlMode.RATE, roll_rate=env.rate_command[0], pitch_rate=env.rate_command[1],
                           yaw_rate=env.rate_command[2], throttle=0.6)
        s = pid.compute_action(c, env.sim.get_state(), dt=env.dt)
        pa = np.array([s.aileron, s.elevator, s.rudder, s.throttle], dtype=np.float32)
        comb = pa + res[k] * scale
        comb[:3] = np.clip(comb[:3], -1.0, 1.0)
        comb[3] = np.clip(comb[3], 0.0, 1.0)
        o, r, term, trunc, settled = env.step(comb)
        # the reference pins numpy<2 (requirements.txt:5): float32 scalar / python float -> float64 there, so only the
        # magnitude is float32; spelled out so NumPy 2's weak-scalar promotion does not round the reward to float32
        r = float(r) + 0.05 * (1.0 - float(np.sum(res[k][:3] ** 2)) / 3.0)
        obs.append(o); rews.append(float(r)); flags.append([term, trunc]); pid_acts.append(pa); comb_acts.append(comb.copy())
        if term or trunc:
            break
    save("env_residual_medium_step_seed17.npz", obs=np.array(obs), residual=res[:len(rews)], rewards=np.array(rews),
         flags=np.array(flags), pid_actions=np.array(pid_acts), combined=np.array(comb_acts), scale=scale)

    # reward / settle unit sequences (rewards.py:48-137,168-221)
    T = 300
    errs = rs.normal(0, 0.2, (T, 3)) * np.exp(-np.arange(T) / 80.0)[:, None]
    errs[::17] = 0.0                      # np.sign(0) == 0 branch
    acts = rs.uniform(-1, 1, (T, 4))
    prev = np.vstack([[0, 0, 0, 0.5], acts[:-1]])
    fl = np.stack([rs.uniform(5, 30, T), rs.uniform(0, 150, T), rs.uniform(-2, 2, T), rs.uniform(-1.4, 1.4, T)], 1)
    cmd = np.array([0.4, -0.02, 0.0])
    rt, sb = rewards_mod.RateTrackingReward(), rewards_mod.SettlingTimeBonus()
    r1, r2, st = np.zeros(T), np.zeros(T), np.zeros(T)
    comps = np.zeros((T, 5))
    for t in range(T):
        r1[t], c = rt.compute(errs[t, 0], errs[t, 1], errs[t, 2], acts[t], prev[t], fl[t, 0], fl[t, 1],
                              fl[t, 2], fl[t, 3])
        comps[t] = [c["tracking"], c["smoothness"], c["stability"], c["oscillation"], c["survival"]]
        r2[t] = sb.compute(errs[t, 0], errs[t, 1], errs[t, 2], cmd[0], cmd[1], cmd[2], 0.02)
        st[t] = float(sb.is_settled)
    save("rewards_sequence.npz", errs=errs, actions=acts, prev_actions=prev, flight=fl, cmd=cmd,
         tracking_reward=r1, components=comps, settle_reward=r2, settled=st)


def gen_eval():
    """Episode trajectories recorded the way eval_rate.py:129-235 (evaluate_pid_controller) does -- pre-step rates and
    command, post-step time, float64 RateAgent actions, throttle 0.5, PID dt = ControllerConfig.rate_loop_dt -- and the
    reference's MetricsCalculator.compute_metrics / aggregate over them (learned_controllers/eval/metrics.py:95-362)."""
    print("eval metrics (reference MetricsCalculator)")
    metrics_mod = _load_by_path("_ref_metrics", "learned_controllers/eval/metrics.py")
    calc = metrics_mod.MetricsCalculator()
    fields = list(metrics_mod.RateControlMetrics.__dataclass_fields__)
    assert fields[:3] == ["settling_time_roll", "settling_time_pitch", "settling_time_yaw"] and len(fields) == 17
    eps = []
    for diff, ct, seed in (("easy", "step", 3), ("medium", "step", 11), ("hard", "step", 21), ("medium", "ramp", 9),
                           ("medium", "sine", 13), ("hard", "random", 5), ("easy", "step", 42), ("medium", "step", 8)):
        env = RefComposedRateEnv(difficulty=diff, command_type=ct, rng_seed=seed)
        env.reset(seed=seed)
        pid = RateAgent(ControllerConfig())
        x0, cmd0 = state_vec(env.sim._physics), env.rate_command.copy()
        times, rates, cmds, acts, rews = [], [], [], [], []
        while True:
            st = env.sim.get_state()
            rc = env.rate_command.copy()
            c = ControlCommand(mode=ControlMode.RATE, roll_rate=rc[0], pitch_rate=rc[1], yaw_rate=rc[2], throttle=0.5)
            sf = pid.compute_action(c, st)
            a = np.array([sf.aileron, sf.elevator, sf.rudder, sf.throttle])
            _, r, term, trunc, _ = env.step(a)
            times.append(env.t); rates.append([st.p, st.q, st.r]); cmds.append(rc); acts.append(a); rews.append(r)
            if term or trunc:
                break
        eps.append(dict(times=np.array(times), rates=np.array(rates), commands=np.array(cmds), actions=np.array(acts),
                        rewards=np.array(rews), x0=x0, cmd0=cmd0, terminated=term))
    # synthetic edge cases: second-order responses with overshoot, negative / zero / tiny commands, episodes shorter
    # than the settling window, a late step, a response that never settles
    rs = np.random.RandomState(31)
    for L_, cmdv, zeta, noise in ((500, [0.8, -0.5, 0.0], 0.3, 0.0), (500, [0.005, 0.2, -0.9], 0.7, 0.01),
                                  (7, [0.4, 0.4, 0.4], 0.5, 0.0), (11, [0.3, -0.3, 0.02], 0.9, 0.0),
                                  (260, [-1.2, 0.05, 0.6], 0.15, 0.05), (500, [0.0, 0.0, 0.0], 0.5, 0.0),
                                  (2, [0.5, 0.0, 0.0], 0.5, 0.0), (137, [0.6, -0.6, 0.3], 0.05, 0.2)):
        t = 0.02 * np.cumsum(np.ones(L_))
        wn = np.array([6.0, 4.0, 9.0])
        wd = wn * np.sqrt(1 - zeta ** 2)
        resp = 1 - np.exp(-zeta * wn * t[:, None]) * (np.cos(wd * t[:, None]) + zeta / np.sqrt(1 - zeta ** 2) * np.sin(wd * t[:, None]))
        cmd = np.tile(np.array(cmdv, dtype=float), (L_, 1))
        if L_ == 260:
            cmd[:40] = 0.0                                              # late step
        rate = cmd * resp + noise * rs.normal(size=(L_, 3))
        act = np.clip(np.cumsum(rs.normal(0, 0.03, (L_, 4)), 0), -1, 1)
        eps.append(dict(times=np.cumsum(np.full(L_, 0.02)), rates=rate, commands=cmd, actions=act,
                        rewards=rs.normal(0.5, 0.3, L_), x0=np.zeros(12), cmd0=np.zeros(3), terminated=False))
    out, agg_in = {}, []
    for j, e in enumerate(eps):
        with np.errstate(all="ignore"):
            m = calc.compute_metrics(e["times"], e["rates"], e["commands"], e["actions"], e["rewards"])
        agg_in.append(m)
        for k_, v in e.items():
            out[f"ep{j}_{k_}"] = np.asarray(v)
        out[f"ep{j}_metrics"] = np.array([float(getattr(m, f)) for f in fields])
        print("   ep", j, "len", len(e["times"]), "settle", out[f"ep{j}_metrics"][:3], "success", m.success)
    # aggregate_metrics (eval_rate.py:238-263) is in a module that imports SB3 at the top, so it cannot be imported;
    # its arithmetic (np.mean per field, success rate) is applied to the reference's per-episode outputs here.
    out["n_episodes"] = np.array(len(eps))
    out["fields"] = np.array(fields)
    out["settling_threshold"] = np.array(calc.settling_threshold)
    out["settle_steps"] = np.array(int(calc.settling_duration / calc.dt))
    save("eval_metrics.npz", **out)


def gen_sensor():
    """NoisySensorInterface (interfaces/sensor.py:137-243) over an open-loop flight: measured states and bias walks, plus
    the standard normals its default_rng(seed) stream yields in the same call order (rng.normal(0, s, k) is
    0 + s * standard_normal(k) on the same bit stream), so a checker can replay the update without NumPy's generator."""
    print("sensor layer (reference NoisySensorInterface)")
    sensor_mod = _load_by_path("_ref_sensor", "interfaces/sensor.py")
    out = {}
    for tag, cfg in (("default", {"seed": 7}),
                     ("custom", {"seed": 11, "imu_gyro_stddev": 0.03, "gps_position_stddev": 2.5, "gps_velocity_stddev": 0.3,
                                 "airspeed_stddev": 0.8, "altitude_stddev": 1.5, "attitude_stddev": 0.02}),
                     ("disabled", {"seed": 3, "enabled": False})):
        sim = Simplified6DOF()
        sim.reset()
        sim.set_controls(ControlSurfaces(elevator=0.05, aileron=0.1, rudder=-0.05, throttle=0.7))
        sens = sensor_mod.NoisySensorInterface(dict(cfg))
        replay = np.random.default_rng(cfg["seed"])
        xs, zs, meas, bias, va = [], [], [], [], []
        for k in range(200):
            for _ in range(5):
                sim.step(0.002)
            st = sim.get_state()
            xs.append(state_vec(sim))
            va.append([st.airspeed, st.altitude])
            sens.update(st)
            m = sens.get_state()
            meas.append(np.concatenate([m.position, m.velocity, m.attitude, m.angular_rate, [m.airspeed, m.altitude]]))
            bias.append(np.concatenate([sens._gyro_bias, sens._accel_bias]))
            zs.append(replay.standard_normal(20) if cfg.get("enabled", True) else np.zeros(20))
        p_ = sens.get_noise_parameters()
        out[f"{tag}_x"], out[f"{tag}_z"], out[f"{tag}_meas"], out[f"{tag}_bias"] = map(np.array, (xs, zs, meas, bias))
        out[f"{tag}_airspeed_altitude"] = np.array(va)
        out[f"{tag}_cfg"] = np.array([p_["gps_position_stddev"], p_["gps_velocity_stddev"], p_["attitude_stddev"],
                                      p_["imu_gyro_stddev"], p_["airspeed_stddev"], p_["altitude_stddev"], 0.0001, 0.001,
                                      float(p_["enabled"])])
        print("  ", tag, "gyro bias after 200 updates", sens._gyro_bias)
    save("sensor_noisy.npz", **out)


def gen_telemetry():
    """The reference's TelemetryLogger (visualization/logger.py:9-152) fed a short two-aircraft flight: the JSON document it
    writes, kept as the fixture (data), plus the inputs."""
    import json, tempfile
    print("telemetry file format (reference TelemetryLogger)")
    logger_mod = _load_by_path("_ref_logger", "visualization/logger.py")
    inputs = {"aircraft": {}}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "flight.json")
        log = logger_mod.TelemetryLogger(path)
        log.register_aircraft("alpha", {"mission": "square", "speed": 15.0})
        for name, thr in (("alpha", 0.7), ("bravo", 0.5)):          # bravo is auto-registered by its first log call
            sim = Simplified6DOF()
            sim.reset()
            surf = ControlSurfaces(elevator=0.02, aileron=0.1, rudder=-0.03, throttle=thr)
            sim.set_controls(surf)
            rows = []
            for k in range(4):
                for _ in range(5):
                    sim.step(0.002)
                st = sim.get_state()
                log.log_state(name, st)
                log.log_command(name, ControlCommand(mode=ControlMode.RATE, roll_rate=0.1, pitch_rate=0.0, yaw_rate=0.0, throttle=thr), st.time)
                log.log_surfaces(name, surf, st.time)
                rows.append(np.concatenate([[st.time], state_vec(sim), [st.airspeed, st.altitude]]).tolist())
            inputs["aircraft"][name] = {"rows": rows, "surfaces": [surf.elevator, surf.aileron, surf.rudder, surf.throttle]}
        log.close()
        doc = json.load(open(path))
    with open(os.path.join(OUT, "telemetry_reference.json"), "w") as f:
        json.dump({"inputs": inputs, "document": doc}, f)
    print("   wrote telemetry_reference.json")


def gen_spaces():
    """RLAgentInterface's level-dependent helpers (interfaces/agent.py:154-323) for an agent at each of the five levels --
    space definitions, preprocess_observation on a few flown states, the default switch_control_level error -- and the
    Level-5 SurfaceAgent (controllers/surface_agent.py:8-103) on in-range / out-of-range commands."""
    import json
    from interfaces.agent import RLAgentInterface
    from controllers.surface_agent import SurfaceAgent
    print("agent observation / action spaces (reference RLAgentInterface) + SurfaceAgent")

    class LevelAgent(RLAgentInterface):
        def __init__(self, level):
            self.level = level

        def get_control_level(self):
            return self.level

        def reset(self, initial_state):
            pass

        def get_action(self, observation):
            return ControlCommand(mode=self.level)

    sim = Simplified6DOF()
    sim.reset()
    sim.set_controls(ControlSurfaces(elevator=-0.05, aileron=0.2, rudder=0.05, throttle=0.8))
    states = []
    for _ in range(3):
        for _ in range(40):
            sim.step(0.005)
        states.append(sim.get_state())

    def plain(v):
        if isinstance(v, np.ndarray):
            return v.tolist()
        if isinstance(v, tuple):
            return list(v)
        return v
    doc = {"states": [np.concatenate([state_vec_of(s), [s.airspeed, s.altitude]]).tolist() for s in states], "levels": {},
           "surface_agent": []}
    for mode in (ControlMode.WAYPOINT, ControlMode.HSA, ControlMode.ATTITUDE, ControlMode.RATE, ControlMode.SURFACE):
        ag = LevelAgent(mode)
        try:
            ag.switch_control_level(ControlMode.HSA)
            switch = None
        except NotImplementedError as e:
            switch = str(e)
        doc["levels"][mode.name] = {
            "observation_space": {k: plain(v) for k, v in ag.get_observation_space().items()},
            "action_space": {k: plain(v) for k, v in ag.get_action_space().items()},
            "observations": [ag.preprocess_observation(s).tolist() for s in states],
            "repr": repr(ag), "switch_error": switch}
    for cfg in (None, {"surface_limits": {"elevator_min": -0.5, "elevator_max": 0.4, "throttle_max": 0.9, "rudder_min": -0.2}}):
        ag = SurfaceAgent(cfg)
        cases = []
        for cmd in ((0.3, -0.2, 0.1, 0.6), (-1.7, 1.4, -0.9, 1.3), (0.45, -1.0, 0.25, -0.2)):
            out = ag.compute_action(ControlCommand(mode=ControlMode.SURFACE, elevator=cmd[0], aileron=cmd[1], rudder=cmd[2],
                                                   throttle=cmd[3]), states[0])
            cases.append({"command": list(cmd), "surfaces": [float(out.elevator), float(out.aileron), float(out.rudder), float(out.throttle)]})
        doc["surface_agent"].append({"config": cfg, "repr": repr(ag), "level": ag.get_control_level().name, "cases": cases})
    with open(os.path.join(OUT, "agent_spaces.json"), "w") as f:
        json.dump(doc, f)
    print("   wrote agent_spaces.json")


def state_vec_of(st):
    return np.concatenate([st.position, st.velocity, st.attitude, st.angular_rate])


if __name__ == "__main__":
    which = sys.argv[1:] or ["open", "stress", "pid", "agents", "cfg1", "cfg3", "samplers", "env", "eval", "sensor", "telemetry", "spaces"]
    for w in which:
        {"open": gen_open_loop, "stress": gen_stress, "pid": gen_pid, "agents": gen_agents, "cfg1": gen_cfg1,
         "cfg3": gen_cfg3, "samplers": gen_samplers, "env": gen_env, "eval": gen_eval, "sensor": gen_sensor,
         "telemetry": gen_telemetry, "spaces": gen_spaces}[w]()
    print("done")
