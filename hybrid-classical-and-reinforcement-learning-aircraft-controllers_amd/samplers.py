"""Host-side episode samplers for the rate-control env.

`RateCommandGenerator` / `FlightEnvelopeSampler` keep the reference's constructor arguments, method names and --
what parity needs -- its exact sequence of `np.random.RandomState` calls
(learned_controllers/data/generators.py:7-164 and :167-214), so that identical seeds give identical episodes.
MT19937 cannot be reproduced by a counter-based device generator, hence two modes downstream:
  * parity mode     -- `presample_reset_pool` draws each env's episodes here, in the reference's call order
                       (rate_env.py:64-77,171,306-336), and the pool is uploaded once; the kernel indexes it.
  * throughput mode -- the kernel draws from an in-kernel counter-based generator with the same distributions.
"""
from typing import Optional, Tuple

import numpy as np

from . import layout as L

DIFFICULTY_SCALE = {"easy": 0.3, "medium": 0.5, "hard": 0.7}
COMMAND_TYPE = {"step": L.FD_CMD_STEP, "ramp": L.FD_CMD_RAMP, "sine": L.FD_CMD_SINE, "random": L.FD_CMD_RANDOM_WALK}


class RateCommandGenerator:
    def __init__(self, max_roll_rate: float = np.radians(180), max_pitch_rate: float = np.radians(180),
                 max_yaw_rate: float = np.radians(160), difficulty: str = "medium",
                 rng_seed: Optional[int] = None):
        self.max_roll_rate, self.max_pitch_rate, self.max_yaw_rate = max_roll_rate, max_pitch_rate, max_yaw_rate
        self.difficulty = difficulty
        self.rng = np.random.RandomState(rng_seed)
        self._max_rates = np.array([max_roll_rate, max_pitch_rate, max_yaw_rate])
        self.difficulty_scale = DIFFICULTY_SCALE[difficulty]

    def _signed_magnitudes(self, out, axes, lo=0.3):
        for ax in axes:
            mag = self.rng.uniform(lo, 1.0) * self.difficulty_scale
            out[ax] = self.rng.choice([-1, 1]) * mag * self._max_rates[ax]
        return out

    def generate_step_command(self, num_axes: int = 1, hold_time: float = 2.0) -> Tuple[np.ndarray, str]:
        axes = self.rng.choice(3, size=min(num_axes, 3), replace=False)
        cmd = self._signed_magnitudes(np.zeros(3), axes)
        return cmd, "Step: " + ", ".join(("roll", "pitch", "yaw")[i] for i in axes)

    def generate_ramp_command(self, duration: float = 3.0):
        n = self.rng.choice([1, 2, 3])
        axes = self.rng.choice(3, size=n, replace=False)
        return np.zeros(3), self._signed_magnitudes(np.zeros(3), axes), f"Ramp: {duration:.1f}s"

    def generate_sine_command(self, frequency: Optional[float] = None, amplitude_scale: float = 0.5):
        if frequency is None:
            frequency = self.rng.uniform(0.1, 2.0)
        n = self.rng.choice([1, 2])
        axes = self.rng.choice(3, size=n, replace=False)
        amps = np.zeros(3)
        for ax in axes:
            amp = self.rng.uniform(0.3, 1.0) * amplitude_scale * self.difficulty_scale
            amps[ax] = amp * self._max_rates[ax]
        return frequency, amps, f"Sine: {frequency:.2f} Hz"

    def generate_multi_axis_command(self):
        cmd = np.zeros(3)
        for ax in range(3):
            mag = self.rng.uniform(0.4, 1.0) * self.difficulty_scale
            cmd[ax] = self.rng.choice([-1, 1]) * mag * self._max_rates[ax]
        return cmd, "Multi-axis coupled"

    def generate_random_walk(self, dt: float = 0.1, diffusion: float = 0.1):
        delta = self.rng.randn(3) * diffusion * np.sqrt(dt)
        delta *= self.difficulty_scale * self._max_rates
        return delta, "Random walk"


class FlightEnvelopeSampler:
    def __init__(self, airspeed_range=(15.0, 30.0), altitude_range=(50.0, 200.0),
                 attitude_range=(np.radians(-15), np.radians(15)), rng_seed: Optional[int] = None):
        self.airspeed_range, self.altitude_range, self.attitude_range = airspeed_range, altitude_range, attitude_range
        self.rng = np.random.RandomState(rng_seed)

    def sample(self) -> dict:
        u = self.rng.uniform
        airspeed = u(*self.airspeed_range)
        altitude = u(*self.altitude_range)
        roll = u(*self.attitude_range)
        pitch = u(*self.attitude_range)
        yaw = u(0, 2 * np.pi)
        p, q, r = u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1)
        return {"airspeed": airspeed, "altitude": altitude, "attitude": np.array([roll, pitch, yaw]),
                "angular_rate": np.array([p, q, r])}


class EpisodeStreams:
    """The three independent MT19937 streams one reference env owns (rate_env.py:64,73-77): env.rng,
    cmd_generator.rng and envelope_sampler.rng, all seeded with the same `rng_seed`."""

    def __init__(self, difficulty: str, command_type: str, rng_seed: Optional[int]):
        self.command_type = command_type
        self.rng = np.random.RandomState(rng_seed)
        self.cmd_generator = RateCommandGenerator(difficulty=difficulty, rng_seed=rng_seed)
        self.envelope_sampler = FlightEnvelopeSampler(rng_seed=rng_seed)

    def reseed_env_rng(self, seed):
        """`reset(seed=...)` re-seeds ONLY env.rng (rate_env.py:167-168)."""
        self.rng = np.random.RandomState(seed)

    def next_record(self) -> np.ndarray:
        """One reset: initial condition then command, in rate_env.py:171 / :188 order."""
        rec = np.zeros(L.FD_NR, dtype=np.float64)
        ic = self.envelope_sampler.sample()
        rec[L.FD_R_AIRSPEED], rec[L.FD_R_ALTITUDE] = ic["airspeed"], ic["altitude"]
        rec[L.FD_R_ROLL:L.FD_R_YAW + 1] = ic["attitude"]
        rec[L.FD_R_P:L.FD_R_R + 1] = ic["angular_rate"]
        ct = self.command_type
        if ct == "step":
            k = self.rng.choice([1, 2, 3])
            cmd, _ = self.cmd_generator.generate_step_command(num_axes=k)
            rec[L.FD_R_CMD0:L.FD_R_CMD2 + 1] = cmd
        elif ct == "ramp":
            _, end, _ = self.cmd_generator.generate_ramp_command()
            rec[L.FD_R_CMD0:L.FD_R_CMD2 + 1] = end
        elif ct == "sine":
            f, amps, _ = self.cmd_generator.generate_sine_command()
            rec[L.FD_R_CMD0:L.FD_R_CMD2 + 1] = amps
            rec[L.FD_R_CMD3] = f
        return rec

    def random_walk_delta(self, dt: float) -> np.ndarray:
        return self.cmd_generator.generate_random_walk(dt=dt)[0]


def presample_reset_pool(seeds, depth: int, difficulty: str, command_type: str,
                         reseed_first: bool = True) -> np.ndarray:
    """[n_envs][depth][FD_NR] float64 pool: env i's k-th episode, drawn in the reference's call order.
    `reseed_first` mirrors the vec-env's first `reset(seed=seed+rank)` (re-seeding env.rng with the same value
    it was built with is a no-op on the stream, kept for clarity)."""
    pool = np.zeros((len(seeds), depth, L.FD_NR), dtype=np.float64)
    for i, s in enumerate(seeds):
        st = EpisodeStreams(difficulty, command_type, None if s is None else int(s))
        if reseed_first and s is not None:
            st.reseed_env_rng(int(s))
        for k in range(depth):
            pool[i, k] = st.next_record()
    return pool


def env_consts(difficulty: str = "medium", episode_length: float = 10.0, dt: float = 0.02,
               command_type: str = "step", dt_physics: float = 0.001) -> np.ndarray:
    """[FD_NEC] float64 constants of one env configuration (rate_env.py:59-103)."""
    EC = np.zeros(L.FD_NEC, dtype=np.float64)
    EC[L.FD_EC_DT] = dt
    EC[L.FD_EC_DT_PHYSICS] = dt_physics
    EC[L.FD_EC_MAX_STEPS] = int(episode_length / dt)
    EC[L.FD_EC_CMD_TYPE] = COMMAND_TYPE.get(command_type, L.FD_CMD_STEP) if command_type in COMMAND_TYPE else -1
    EC[L.FD_EC_DIFFICULTY_SCALE] = DIFFICULTY_SCALE[difficulty]
    EC[L.FD_EC_MAX_RATE_P], EC[L.FD_EC_MAX_RATE_Q], EC[L.FD_EC_MAX_RATE_R] = (
        np.radians(180), np.radians(180), np.radians(160))
    return EC
