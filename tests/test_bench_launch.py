"""CPU: bench.py's self-launch decision (no GPU needed -- the child launcher is intercepted)."""
import importlib.util
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # __name__ != "__main__": nothing is launched or measured on import
    return mod


def test_gpus_n_without_launcher_env_starts_n_child_ranks(monkeypatch, capsys):
    bench = _bench_module()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    class Done:
        returncode = 0
        stdout = b'[Gloo] chatter\n{"metric": "env-steps/sec", "value": 1.0, "n_gpus": 4}\n'

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    with pytest.raises(SystemExit) as ex:
        bench._self_launch_if_needed()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    out = capsys.readouterr().out.splitlines()
    assert out == ['{"metric": "env-steps/sec", "value": 1.0, "n_gpus": 4}']      # exactly the relayed JSON line


def test_no_self_launch_for_one_gpu_or_under_a_launcher(monkeypatch):
    bench = _bench_module()
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not launch")))
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    bench._self_launch_if_needed()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench._self_launch_if_needed()
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "3")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus=8"])
    bench._self_launch_if_needed()


def test_failed_children_propagate_a_nonzero_exit(monkeypatch, capsys):
    bench = _bench_module()
    monkeypatch.delenv("WORLD_SIZE", raising=False); monkeypatch.delenv("RANK", raising=False)

    class Failed:
        returncode = 3
        stdout = b""

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Failed())
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as ex:
        bench._self_launch_if_needed()
    assert ex.value.code == 3 and capsys.readouterr().out == ""
