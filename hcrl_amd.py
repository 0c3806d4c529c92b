"""Import alias: `import hcrl_amd` == the package in
`hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd/` (whose name is not an identifier)."""
import importlib
import sys

_REAL = "hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd"
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg


class _AliasFinder:
    """Make `import hcrl_amd.x` resolve to the one real submodule object instead of a second copy."""

    @staticmethod
    def find_spec(name, path=None, target=None):
        if not name.startswith("hcrl_amd."):
            return None
        real = importlib.import_module(_REAL + name[len("hcrl_amd"):])
        # do NOT pre-register sys.modules[name] here: importlib's _find_spec would then take the registered module's own
        # spec (real name, source loader) and execute the file a second time -- two copies of every class
        return importlib.util.spec_from_loader(name, loader=_Loader(real))


class _Loader:
    def __init__(self, mod):
        self.mod = mod

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        pass


import importlib.util  # noqa: E402
sys.meta_path.insert(0, _AliasFinder)
