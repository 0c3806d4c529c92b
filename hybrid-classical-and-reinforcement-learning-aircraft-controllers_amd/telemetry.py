"""Telemetry on-disk format: host mirror of visualization/logger.py:9-152 (`TelemetryLogger`), plus a fleet recorder that
keeps the samples in device memory until the file is written.

File format (JSON, logger.py:139-146): {"metadata": {aircraft_id: {...}}, "data": {aircraft_id: {"states": [{time,
position, velocity, attitude, angular_rate, airspeed, altitude}], "commands": [{time, mode}], "surfaces": [{time, aileron,
elevator, rudder, throttle}], "times": [...]}}}.  A `.hdf5` / `.h5` path needs h5py (groups per aircraft with `times`,
`positions`, `attitudes` datasets, logger.py:128-138); without it the logger falls back to JSON with the reference's
warning, as the reference does.
"""
import json
from pathlib import Path
from typing import Any, Dict, Optional, Sequence

import numpy as np


class TelemetryLogger:
    def __init__(self, filepath: str):
        self.filepath = Path(filepath)
        self.filepath.parent.mkdir(parents=True, exist_ok=True)
        self._aircraft_data: Dict[str, Dict] = {}
        self._aircraft_metadata: Dict[str, Dict] = {}
        self._use_hdf5 = filepath.endswith(".hdf5") or filepath.endswith(".h5")
        self._h5file = None
        if self._use_hdf5:
            try:
                import h5py
                self._h5file = h5py.File(filepath, "w")
            except ImportError:
                print("Warning: h5py not available, falling back to JSON logging")
                self._use_hdf5 = False
                self.filepath = self.filepath.with_suffix(".json")

    @staticmethod
    def _as_list(value):
        return value.tolist() if hasattr(value, "tolist") else list(value)

    def _ensure_registered(self, aircraft_id: str):
        if aircraft_id not in self._aircraft_data:
            self.register_aircraft(aircraft_id)

    def register_aircraft(self, aircraft_id: str, metadata: Optional[Dict] = None):
        self._aircraft_data[aircraft_id] = {"states": [], "commands": [], "surfaces": [], "times": []}
        self._aircraft_metadata[aircraft_id] = metadata or {}
        if self._use_hdf5 and self._h5file:
            grp = self._h5file.create_group(aircraft_id)
            for key, val in (metadata or {}).items():
                grp.attrs[key] = str(val)

    def log_state(self, aircraft_id: str, state: Any):
        self._ensure_registered(aircraft_id)
        d = self._aircraft_data[aircraft_id]
        d["states"].append({"time": state.time, "position": self._as_list(state.position),
                            "velocity": self._as_list(state.velocity), "attitude": self._as_list(state.attitude),
                            "angular_rate": self._as_list(state.angular_rate), "airspeed": state.airspeed,
                            "altitude": state.altitude})
        d["times"].append(state.time)

    def log_command(self, aircraft_id: str, command: Any, time: float):
        self._ensure_registered(aircraft_id)
        mode = command.mode.name if hasattr(command.mode, "name") else str(command.mode)
        self._aircraft_data[aircraft_id]["commands"].append({"time": time, "mode": mode})

    def log_surfaces(self, aircraft_id: str, surfaces: Any, time: float):
        self._ensure_registered(aircraft_id)
        self._aircraft_data[aircraft_id]["surfaces"].append({"time": time, "aileron": surfaces.aileron,
                                                             "elevator": surfaces.elevator, "rudder": surfaces.rudder,
                                                             "throttle": surfaces.throttle})

    def close(self):
        if self._use_hdf5 and self._h5file:
            for aircraft_id, data in self._aircraft_data.items():
                grp = self._h5file[aircraft_id]
                if data["times"]:
                    grp.create_dataset("times", data=np.array(data["times"]))
                if data["states"]:
                    grp.create_dataset("positions", data=np.array([s["position"] for s in data["states"]]))
                    grp.create_dataset("attitudes", data=np.array([s["attitude"] for s in data["states"]]))
            self._h5file.close()
        else:
            with open(self.filepath, "w") as f:
                json.dump({"metadata": self._aircraft_metadata, "data": self._aircraft_data}, f, indent=2)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()
        return False


class FleetRecorder:
    """Samples of a whole fleet ([12][N] state blocks + [4][N] surfaces) appended on the device; `write` lays selected
    aircraft out in the TelemetryLogger file format (one small D2H copy per recorded sample block at the end, not per step)."""

    def __init__(self, fleet, capacity: int, every: int = 1):
        import torch
        self.fleet, self.every, self.capacity = fleet, int(every), int(capacity)
        n, dev = fleet.n, fleet.device
        self.states = torch.zeros((capacity, 12, n), dtype=fleet.dtype, device=dev)
        self.surfaces = torch.zeros((capacity, 4, n), dtype=fleet.dtype, device=dev)
        self.times = []
        self._calls = 0

    def sample(self, surfaces=None):
        """Record the fleet's current state (every `every`-th call); `surfaces` [4][N] in FD_U_* order, if any."""
        self._calls += 1
        if (self._calls - 1) % self.every or len(self.times) >= self.capacity:
            return
        k = len(self.times)
        self.states[k].copy_(self.fleet.x)
        if surfaces is not None:
            self.surfaces[k].copy_(surfaces)
        self.times.append(float(self.fleet.time))

    def write(self, filepath: str, aircraft: Optional[Sequence[int]] = None, metadata: Optional[Dict] = None, mode_name: str = "WAYPOINT"):
        from .flight_types import AircraftState, ControlSurfaces
        k = len(self.times)
        ids = list(range(self.fleet.n)) if aircraft is None else list(aircraft)
        xs = self.states[:k][:, :, ids].to("cpu").double().numpy()            # [k, 12, len(ids)]
        us = self.surfaces[:k][:, :, ids].to("cpu").double().numpy()
        with TelemetryLogger(filepath) as log:
            for j, a in enumerate(ids):
                name = f"aircraft_{a}"
                log.register_aircraft(name, dict(metadata or {}, index=a))
                for s in range(k):
                    x = xs[s, :, j]
                    st = AircraftState.from_vector(x, time=self.times[s])
                    st.airspeed, st.altitude = float(np.sqrt(x[3] ** 2 + x[4] ** 2 + x[5] ** 2)), float(-x[2])
                    log.log_state(name, st)
                    log.log_command(name, type("Cmd", (), {"mode": mode_name})(), self.times[s])
                    u = us[s, :, j]
                    log.log_surfaces(name, ControlSurfaces(elevator=float(u[0]), aileron=float(u[1]), rudder=float(u[2]),
                                                           throttle=float(u[3])), self.times[s])
        return log.filepath
