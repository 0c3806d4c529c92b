"""Policy weights in the Stable-Baselines3 `.zip` layout (weights only, nothing is ever unpickled).

The reference stores and loads its controllers as SB3 archives (`train_rate.py:353-355` `model.save(.../final_model)`,
`controllers/learned_rate_agent.py:87-118` `RecurrentPPO.load(path)` / `PPO.load(path)`).  An SB3 archive is a zip of

    data                           JSON of the algorithm's attributes; anything not JSON-able is a
                                   {":type:", ":serialized:" <base64 cloudpickle>} record
    policy.pth                     torch.save(policy.state_dict())
    policy.optimizer.pth           torch.save(optimizer.state_dict())
    pytorch_variables.pth          extra tensors (none for PPO)
    _stable_baselines3_version     text
    system_info.txt                text

(stable-baselines3 `common/save_util.py` save_to_zip_file / load_from_zip_file; the package itself is absent here, so
the layout and the parameter names below follow its published source and are NOT checked against it: parity unpinned.)

What this module does with that format:
  * READ: `policy.pth` through `torch.load(weights_only=True)` and the plain-JSON part of `data`; every
    ":serialized:" record is dropped unread.  The network sizes are taken from the tensor shapes.
  * WRITE: `policy.pth` with SB3's parameter names, a `data` file holding the JSON-able hyper-parameters, and the
    version / system-info text files.  No optimizer state (its parameter order is SB3's, not ours) and no pickled
    classes: on the reference side load the weights with `model.policy.load_state_dict(...)`.

Parameter names.  `RateLSTMPolicy` keeps sb3_contrib's names except for the two trunks:
    pi_net.N.*  <->  mlp_extractor.policy_net.N.*        vf_net.N.*  <->  mlp_extractor.value_net.N.*
and an SB3 policy that shares its features extractor lists those tensors three times (`features_extractor.*`,
`pi_features_extractor.*`, `vf_features_extractor.*` are one module under three attribute names).
"""
import io
import json
import platform
import zipfile
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

SB3_VERSION_WRITTEN = "2.3.2"          # the release whose layout is reproduced (requirements of the reference: >= 2.0)
_TRUNKS = (("pi_net.", "mlp_extractor.policy_net."), ("vf_net.", "mlp_extractor.value_net."))
_FE, _FE_ALIASES = "features_extractor.", ("pi_features_extractor.", "vf_features_extractor.")


def to_sb3_state_dict(policy) -> "OrderedDict[str, torch.Tensor]":
    """state_dict of a RateLSTMPolicy under SB3's names (CPU fp32 tensors)."""
    out = OrderedDict()
    for k, v in policy.state_dict().items():
        v = v.detach().to("cpu", torch.float32).contiguous()
        for ours, theirs in _TRUNKS:
            if k.startswith(ours):
                k = theirs + k[len(ours):]
        out[k] = v
        if k.startswith(_FE):
            for alias in _FE_ALIASES:
                out[alias + k[len(_FE):]] = v
    return out


def from_sb3_state_dict(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """SB3-named state_dict -> RateLSTMPolicy names.  Raises if the archive's actor and critic use different feature
    extractors (`share_features_extractor=False`): this policy has one."""
    out = OrderedDict()
    for k, v in sd.items():
        alias = next((a for a in _FE_ALIASES if k.startswith(a)), None)
        if alias is not None:
            base = _FE + k[len(alias):]
            if base in sd and not torch.equal(sd[base], v):
                raise ValueError(f"{k} differs from {base}: separate actor/critic feature extractors are not supported")
            if base in sd:
                continue
            k = base
        for ours, theirs in _TRUNKS:
            if k.startswith(theirs):
                k = ours + k[len(theirs):]
        out[k] = v
    return out


def policy_kwargs_from_state_dict(sd: Dict[str, torch.Tensor]) -> dict:
    """Constructor arguments of RateLSTMPolicy implied by the tensor shapes of a (RateLSTMPolicy-named) state_dict."""
    def trunk(prefix):
        idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith(prefix) and k.endswith(".weight")})
        return tuple(int(sd[f"{prefix}{i}.weight"].shape[0]) for i in idx)
    if not any(k.startswith("lstm_actor.") for k in sd):
        return {"use_lstm": False, "mlp_net_arch": trunk("pi_net.")}
    n_layers = len([k for k in sd if k.startswith("features_extractor.lstm.weight_ih_l")])
    return {"use_lstm": True,
            "features_dim": int(sd["features_extractor.output_proj.0.weight"].shape[0]),
            "lstm_hidden_size": int(sd["features_extractor.lstm.weight_hh_l0"].shape[1]),
            "n_lstm_layers": n_layers,
            "policy_lstm_hidden": int(sd["lstm_actor.weight_hh_l0"].shape[1]),
            "net_arch_pi": trunk("pi_net."), "net_arch_vf": trunk("vf_net.")}


def is_sb3_zip(path) -> bool:
    """True for an SB3 archive (a torch.save file is a zip too: it holds `<name>/data.pkl`, not `policy.pth`)."""
    try:
        if not zipfile.is_zipfile(path):
            return False
        with zipfile.ZipFile(path) as z:
            return "policy.pth" in z.namelist()
    except OSError:
        return False


def _plain(value):
    """The JSON-able part of one `data` entry: ":serialized:" payloads (cloudpickle) are never decoded."""
    if isinstance(value, dict):
        if ":serialized:" in value:
            keep = {k: _plain(v) for k, v in value.items() if k not in (":serialized:",)}
            keep[":dropped:"] = "serialized payload not read"
            return keep
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, list):
        return [_plain(v) for v in value]
    return value


def read_sb3_zip(path, device="cpu") -> Tuple["OrderedDict[str, torch.Tensor]", dict]:
    """-> (policy state_dict under RateLSTMPolicy names, {"data": plain JSON attributes, "version": str | None})."""
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        if "policy.pth" not in names:
            raise ValueError(f"{path}: not a Stable-Baselines3 archive (no policy.pth)")
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location=device, weights_only=True)
        data = _plain(json.loads(z.read("data").decode())) if "data" in names else {}
        version = z.read("_stable_baselines3_version").decode().strip() if "_stable_baselines3_version" in names else None
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError(f"{path}: policy.pth is not a tensor state_dict")
    return from_sb3_state_dict(sd), {"data": data, "version": version}


def save_sb3_zip(path, policy, hyper: Optional[dict] = None, num_timesteps: int = 0) -> str:
    """Write `policy` (+ JSON-able hyper-parameters) in the SB3 archive layout; returns the path written (`.zip` is
    appended when missing, as SB3's save does)."""
    path = str(path)
    if not path.endswith(".zip"):
        path += ".zip"
    use_lstm = bool(getattr(policy, "use_lstm", True))
    data = {
        "num_timesteps": int(num_timesteps), "_total_timesteps": int(num_timesteps),
        "policy_class": {":type:": "<class 'abc.ABCMeta'>",
                         "__module__": "sb3_contrib.common.recurrent.policies" if use_lstm else "stable_baselines3.common.policies",
                         "__name__": "RecurrentActorCriticPolicy" if use_lstm else "ActorCriticPolicy",
                         ":note:": "class object not pickled; pass custom_objects={'policy_class': ...} or load policy.pth directly"},
        "observation_space": {":type:": "<class 'gymnasium.spaces.box.Box'>", "dtype": "float32", "_shape": [18],
                              ":note:": "rate_env.py:97-100; not pickled"},
        "action_space": {":type:": "<class 'gymnasium.spaces.box.Box'>", "dtype": "float32", "_shape": [4],
                         "low": [-1.0, -1.0, -1.0, 0.0], "high": [1.0, 1.0, 1.0, 1.0], ":note:": "rate_env.py:103-107; not pickled"},
        "policy_kwargs_plain": {k: (list(v) if isinstance(v, tuple) else v)
                                for k, v in policy_kwargs_from_state_dict(policy.state_dict()).items()},
        "written_by": "hcrl_amd.sb3_zip (weights + plain hyper-parameters only)",
    }
    for k, v in (hyper or {}).items():
        if isinstance(v, (bool, int, float, str)) or v is None:
            data[k] = v
    buf = io.BytesIO()
    torch.save(to_sb3_state_dict(policy), buf)
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr("data", json.dumps(data, indent=4))
        z.writestr("policy.pth", buf.getvalue())
        z.writestr("_stable_baselines3_version", SB3_VERSION_WRITTEN)
        z.writestr("system_info.txt", f"- OS: {platform.platform()}\n- Python: {platform.python_version()}\n"
                                      f"- PyTorch: {torch.__version__}\n- writer: hcrl_amd.sb3_zip\n")
    return path
