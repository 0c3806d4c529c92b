import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle is test infrastructure: build it on demand (gcc only, ~1 s)
    if not os.path.exists(os.path.join(REPO, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rel_err(a, b, angle_cols=()):
    """SURVEY §7 parity metric: |a-b| / max(|b|, 1) per component; listed columns compared modulo 2 pi."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = a - b
    for c in angle_cols:
        d[..., c] = (d[..., c] + np.pi) % (2 * np.pi) - np.pi
    return np.abs(d) / np.maximum(np.abs(b), 1.0)


STATE_ANGLE_COLS = (6, 8)   # roll and yaw are re-wrapped every step (simplified_6dof.py:266,270)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    return orc
