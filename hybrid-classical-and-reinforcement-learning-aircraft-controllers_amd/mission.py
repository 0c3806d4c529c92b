"""Waypoint sequencing for ONE aircraft: host mirror of controllers/mission_planner.py:10-290 (`MissionState`,
`MissionPlanner`).  Fleets keep the same bookkeeping on the device: `wp_idx` / `reached_total` of `BatchedCascade`, advanced
in-kernel by the same 3-D acceptance test (csrc/fdyn_core.hpp `waypoint_reached`)."""
from enum import Enum
from typing import List, Optional

import numpy as np

from .flight_types import AircraftState, ControlCommand, ControlMode, Waypoint


class MissionState(Enum):
    IDLE = "idle"
    ACTIVE = "active"
    COMPLETE = "complete"
    ABORTED = "aborted"


class MissionPlanner:
    def __init__(self, waypoints: List[Waypoint], acceptance_radius: float = 10.0, default_speed: Optional[float] = None):
        if not waypoints:
            raise ValueError("Mission must have at least one waypoint")
        self.waypoints, self.acceptance_radius, self.default_speed = waypoints, acceptance_radius, default_speed
        self.reset()

    # ---- life cycle (:78-97) ---------------------------------------------------------------------------------------------
    def start(self):
        if self.state == MissionState.IDLE:
            self.state, self.current_waypoint_index, self.waypoints_reached = MissionState.ACTIVE, 0, 0

    def abort(self):
        self.state = MissionState.ABORTED

    def reset(self):
        self.state = MissionState.IDLE
        self.current_waypoint_index = self.waypoints_reached = 0
        self.waypoint_arrival_times, self.waypoint_distances = [], []
        self.mission_start_time = self.mission_end_time = None

    # ---- guidance (:99-184) ----------------------------------------------------------------------------------------------
    def get_current_waypoint(self) -> Optional[Waypoint]:
        if self.state != MissionState.ACTIVE or self.current_waypoint_index >= len(self.waypoints):
            return None
        return self.waypoints[self.current_waypoint_index]

    def get_waypoint_command(self) -> Optional[ControlCommand]:
        wp = self.get_current_waypoint()
        return None if wp is None else ControlCommand(mode=ControlMode.WAYPOINT, waypoint=wp)

    def update(self, state: AircraftState) -> bool:
        """True when the current waypoint was reached at this state (and the mission advanced)."""
        if self.state != MissionState.ACTIVE:
            return False
        if self.mission_start_time is None:
            self.mission_start_time = state.time
        wp = self.get_current_waypoint()
        if wp is None:
            return False
        distance = self._calculate_distance_to_waypoint(state, wp)
        if not distance < self.acceptance_radius:
            return False
        self.waypoint_arrival_times.append(state.time)
        self.waypoint_distances.append(distance)
        self.waypoints_reached += 1
        self.current_waypoint_index += 1
        if self.current_waypoint_index >= len(self.waypoints):
            self.state, self.mission_end_time = MissionState.COMPLETE, state.time
        return True

    def _is_waypoint_reached(self, state: AircraftState, waypoint: Waypoint) -> bool:
        return self._calculate_distance_to_waypoint(state, waypoint) < self.acceptance_radius

    @staticmethod
    def _calculate_distance_to_waypoint(state: AircraftState, waypoint: Waypoint) -> float:
        return float(np.linalg.norm(np.array([waypoint.north - state.north, waypoint.east - state.east,
                                              waypoint.down - state.down])))

    def get_distance_to_current_waypoint(self, state: AircraftState) -> Optional[float]:
        wp = self.get_current_waypoint()
        return None if wp is None else self._calculate_distance_to_waypoint(state, wp)

    # ---- reporting (:217-290) ----------------------------------------------------------------------------------------------
    def get_progress_percentage(self) -> float:
        return 100.0 if not self.waypoints else (self.waypoints_reached / len(self.waypoints)) * 100.0

    def get_total_mission_distance(self) -> float:
        total = 0.0
        for a, b in zip(self.waypoints[:-1], self.waypoints[1:]):
            total += float(np.linalg.norm(np.array([b.north - a.north, b.east - a.east, b.down - a.down])))
        return total

    def get_mission_duration(self) -> Optional[float]:
        if self.mission_start_time is None or self.mission_end_time is None:
            return None
        return self.mission_end_time - self.mission_start_time

    def is_active(self) -> bool:
        return self.state == MissionState.ACTIVE

    def is_complete(self) -> bool:
        return self.state == MissionState.COMPLETE

    def is_aborted(self) -> bool:
        return self.state == MissionState.ABORTED

    def get_summary(self) -> dict:
        return {"state": self.state.value, "total_waypoints": len(self.waypoints), "waypoints_reached": self.waypoints_reached,
                "progress_percent": self.get_progress_percentage(), "current_waypoint_index": self.current_waypoint_index,
                "total_distance_m": self.get_total_mission_distance(), "mission_duration_s": self.get_mission_duration(),
                "acceptance_radius_m": self.acceptance_radius, "waypoint_arrival_times": self.waypoint_arrival_times,
                "waypoint_distances": self.waypoint_distances}

    def __repr__(self) -> str:
        return (f"MissionPlanner(state={self.state.value}, waypoint={self.current_waypoint_index + 1}/{len(self.waypoints)}, "
                f"progress={self.get_progress_percentage():.1f}%)")
