"""SB3-layout checkpoint I/O (sb3_zip.py): names, round trips, and that nothing pickled is ever decoded.

stable-baselines3 / sb3_contrib are absent: the expected parameter names below restate their published module layout
(RecurrentActorCriticPolicy: features_extractor / pi_features_extractor / vf_features_extractor, lstm_actor, lstm_critic,
mlp_extractor.policy_net / value_net, action_net, value_net, log_std) -- parity unpinned for the format itself.
Reference call sites: train_rate.py:353-355 (save), controllers/learned_rate_agent.py:87-118 (load).
"""
import base64
import io
import json
import zipfile

import pytest
import torch

from hcrl_amd import sb3_zip
from hcrl_amd.policy import RateLSTMPolicy


def test_lstm_policy_names_follow_sb3():
    sd = sb3_zip.to_sb3_state_dict(RateLSTMPolicy())
    keys = set(sd)
    fe = ["embedding.0.weight", "embedding.0.bias", "output_proj.0.weight", "output_proj.0.bias"] + \
         [f"lstm.{n}_l{k}" for k in (0, 1) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    expect = {"log_std", "action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias"}
    expect |= {f"{p}.{k}" for p in ("features_extractor", "pi_features_extractor", "vf_features_extractor") for k in fe}
    expect |= {f"{l}.{n}_l0" for l in ("lstm_actor", "lstm_critic") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    expect |= {f"mlp_extractor.{net}.{i}.{w}" for net in ("policy_net", "value_net") for i in (0, 2) for w in ("weight", "bias")}
    assert keys == expect
    assert sd["mlp_extractor.policy_net.0.weight"].shape == (128, 256) and sd["lstm_actor.weight_ih_l0"].shape == (1024, 128)
    assert sd["pi_features_extractor.lstm.weight_hh_l1"].data_ptr() == sd["features_extractor.lstm.weight_hh_l1"].data_ptr()


def test_mlp_policy_names_follow_sb3():
    keys = set(sb3_zip.to_sb3_state_dict(RateLSTMPolicy(use_lstm=False)))
    expect = {"log_std", "action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias"}
    expect |= {f"mlp_extractor.{net}.{i}.{w}" for net in ("policy_net", "value_net") for i in (0, 2, 4) for w in ("weight", "bias")}
    assert keys == expect


@pytest.mark.parametrize("kw", [{}, {"use_lstm": False}, {"policy_lstm_hidden": 128, "net_arch_pi": (64,), "net_arch_vf": (64, 32)}])
def test_round_trip_and_shape_inference(tmp_path, kw):
    torch.manual_seed(3)
    pol = RateLSTMPolicy(**kw)
    with torch.no_grad():
        for p in pol.parameters():
            p.add_(0.01 * torch.randn_like(p))
    path = sb3_zip.save_sb3_zip(tmp_path / "final_model", pol, hyper={"n_steps": 64, "gamma": 0.99, "skip": (1, 2)}, num_timesteps=123)
    assert path.endswith("final_model.zip") and sb3_zip.is_sb3_zip(path)
    with zipfile.ZipFile(path) as z:
        assert {"data", "policy.pth", "_stable_baselines3_version", "system_info.txt"} <= set(z.namelist())
        data = json.loads(z.read("data"))
    assert data["num_timesteps"] == 123 and data["n_steps"] == 64 and "skip" not in data
    sd, meta = sb3_zip.read_sb3_zip(path)
    assert meta["version"] == sb3_zip.SB3_VERSION_WRITTEN and meta["data"]["gamma"] == 0.99
    again = RateLSTMPolicy(**sb3_zip.policy_kwargs_from_state_dict(sd))
    again.load_state_dict(sd)                                      # strict: every name and shape matches
    for (k, a), (_, b) in zip(pol.state_dict().items(), again.state_dict().items()):
        assert torch.equal(a, b), k


def test_torch_checkpoint_is_not_taken_for_an_archive(tmp_path):
    p = tmp_path / "ck.pt"
    torch.save({"policy": RateLSTMPolicy(use_lstm=False).state_dict()}, p)
    assert zipfile.is_zipfile(p) and not sb3_zip.is_sb3_zip(p)
    assert not sb3_zip.is_sb3_zip(tmp_path / "missing.zip")


def test_serialized_records_are_dropped_unread(tmp_path):
    """An archive written by SB3 carries cloudpickle payloads in `data`; the reader must not decode them."""
    pol = RateLSTMPolicy(use_lstm=False)
    buf = io.BytesIO(); torch.save(sb3_zip.to_sb3_state_dict(pol), buf)
    bomb = base64.b64encode(b"\x80\x04not a pickle we would ever want to run").decode()
    data = {"gamma": 0.99, "policy_class": {":type:": "<class 'abc.ABCMeta'>", ":serialized:": bomb},
            "lr_schedule": {":type:": "<class 'function'>", ":serialized:": bomb}, "nested": [{":serialized:": bomb, "x": 1}]}
    path = tmp_path / "sb3_written.zip"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", json.dumps(data)); z.writestr("policy.pth", buf.getvalue())
        z.writestr("policy.optimizer.pth", b"ignored"); z.writestr("_stable_baselines3_version", "2.1.0\n")
    sd, meta = sb3_zip.read_sb3_zip(path)
    assert meta["version"] == "2.1.0" and meta["data"]["gamma"] == 0.99
    assert ":serialized:" not in json.dumps(meta["data"]) and meta["data"]["nested"][0]["x"] == 1
    RateLSTMPolicy(use_lstm=False).load_state_dict(sd)


def test_pickled_policy_file_is_refused(tmp_path):
    """policy.pth holding anything but tensors must not load (weights_only=True refuses arbitrary objects)."""
    import pickle

    class Evil:
        def __reduce__(self):
            return (print, ("executed",))
    path = tmp_path / "evil.zip"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("policy.pth", pickle.dumps({"w": Evil()})); z.writestr("data", "{}")
    with pytest.raises(Exception):
        sb3_zip.read_sb3_zip(path)


def test_separate_feature_extractors_are_rejected():
    sd = sb3_zip.to_sb3_state_dict(RateLSTMPolicy())
    sd["vf_features_extractor.embedding.0.bias"] = sd["vf_features_extractor.embedding.0.bias"] + 1.0
    with pytest.raises(ValueError, match="separate actor/critic"):
        sb3_zip.from_sb3_state_dict(sd)


def test_load_policy_and_trainer_accept_the_archive(tmp_path):
    from hcrl_amd.eval_rate import load_policy
    torch.manual_seed(0)
    pol = RateLSTMPolicy()
    path = sb3_zip.save_sb3_zip(tmp_path / "m.zip", pol)
    got = load_policy(path, device="cpu", bf16=False)
    obs = torch.randn(5, 18)
    st = pol.initial_state(5)
    with torch.no_grad():
        a0 = pol.step(obs, st, torch.zeros(5), deterministic=True)
        a1 = got.step(obs, st, torch.zeros(5), deterministic=True)
    assert torch.equal(a0[0], a1[0]) and torch.equal(a0[1], a1[1])
