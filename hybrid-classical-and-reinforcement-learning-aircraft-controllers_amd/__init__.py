"""MI355X-native batched flight-dynamics / cascaded-PID / rate-control-env hot path.

The directory name is the one the build contract prescribes; it is not a Python identifier, so import it as
`import hcrl_amd` (the alias module at the repo root) or via importlib.
"""
from . import layout, flight_types, params, config, samplers  # noqa: F401

__version__ = "0.1.0"
