"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports exactly what include/fdyn.h declares,
the host mirror keeps the reference's interface, and the product never touches the oracle or falls back to a CPU path.
No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
import hcrl_amd
from hcrl_amd import _lib, layout as L, config as cfgmod, samplers
from hcrl_amd.flight_types import AircraftState, ControlSurfaces, ControllerConfig, Waypoint
from hcrl_amd.params import AircraftParams

PKG = os.path.join(REPO, "hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd")


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "fdyn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(?:int|int64_t)\s+(fdyn_\w+)\s*\(", src))
    # the env entry points are declared through FDYN_DECLARE_ENV(SUFFIX, S)
    names = {n for n in names if "##" not in n}
    for suffix in re.findall(r"FDYN_DECLARE_ENV\((\w+),", src):
        if suffix != "SUFFIX":
            names |= {f"fdyn_rate_env_reset_{suffix}", f"fdyn_rate_env_step_{suffix}"}
    return names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=REPO)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/fdyn.h but not exported"
    assert declared == set(_lib.SIGNATURES), "host binding table and header drifted apart"
    assert lib.fdyn_abi_version() == 2


def test_code_object_is_gfx950():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.LIB_PATH], capture_output=True,
                         text=True).stdout
    assert "gfx950" in out


def test_num_substeps_matches_python_int_truncation():
    lib = _lib.load()
    for dt, dtp in [(0.03, 0.001), (0.02, 0.001), (0.0005, 0.001), (0.07, 0.01), (0.02, 0.003), (0.1, 0.001), (0.029, 0.001)]:
        assert lib.fdyn_num_substeps(dt, dtp) == max(1, int(dt / dtp))     # simulation_backend.py:95


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hcrl_amd.fleet import BatchedSixDOF
    from hcrl_amd.rate_env import GpuRateVecEnv
    with pytest.raises(_lib.FdynError):
        BatchedSixDOF(4)
    with pytest.raises(_lib.FdynError):
        GpuRateVecEnv(4)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "liboracle" not in text and "flight_oracle" not in text, f
    for f in os.listdir(os.path.join(REPO, "include")):
        assert "oracle" not in open(os.path.join(REPO, "include", f)).read().lower() or f == "fdyn_layout.h"


def test_layout_module_mirrors_header():
    assert L.FD_NX == 12 and L.FD_NU == 4 and L.FD_OBS_DIM == 18 and L.FD_ACT_DIM == 4
    assert L.FD_NP_USED <= L.FD_NP and L.FD_NPID == 9 and L.FD_NPC == 8 and L.FD_NPS == 3
    assert L.FD_EV_NF == 1 + L.FD_OBS_DIM


def test_types_keep_reference_conventions():
    s = ControlSurfaces(elevator=0.1, aileron=-0.2, rudder=0.3, throttle=0.7)
    assert np.allclose(s.to_array(), [0.1, -0.2, 0.3, 0.7])                     # types.py:189-201
    assert ControlSurfaces.from_array(s.to_array()) == s
    w = Waypoint.from_altitude(10, 20, 100, speed=15)
    assert w.down == -100 and w.altitude == 100
    st = AircraftState.from_vector(np.arange(12.0))
    assert st.north == 0 and st.p == 9 and st.yaw == 8 and np.array_equal(st.to_vector(), np.arange(12.0))
    c = ControllerConfig()
    assert (c.roll_rate_gains.kp, c.roll_rate_gains.i_limit, c.max_yaw_rate) == (1.3, 25.0, 160.0)


def test_params_validation_mirrors_reference():
    for kw in ({"mass": 0}, {"inertia_yy": -1}, {"wing_area": 0}, {"air_density": 11}, {"gravity": 0}, {"max_thrust": -1}):
        with pytest.raises(ValueError):
            AircraftParams(**kw)                                                  # simplified_6dof.py:119-143


def test_cascade_tables_alias_the_configs_like_the_reference():
    fc = cfgmod.load_controller_config("cascaded_pid.yaml")
    t = cfgmod.pid_table(ControllerConfig(), fc)
    C = cfgmod.cascade_consts(ControllerConfig(), fc, guidance_type="PP")
    # attitude/rate limits come from the LEGACY config (180/180/160 deg/s, 30/30 deg), not the YAML (200/100/60, 30/20)
    assert np.isclose(C[L.FD_C_MAX_YAW_RATE], np.radians(160)) and np.isclose(C[L.FD_C_MAX_PITCH], np.radians(30))
    assert t[L.FD_PID_ATT_ROLL, L.FD_PC_OUT_MAX] == np.float32(np.radians(180))
    assert t[L.FD_PID_HEADING, L.FD_PC_INT_MAX] == 25.0 and t[L.FD_PID_HEADING, L.FD_PC_OUT_MAX] == np.float32(np.radians(25))
    assert t[L.FD_PID_ENERGY, L.FD_PC_INT_MAX] == 10.0 and t[L.FD_PID_BALANCE, L.FD_PC_INT_MAX] == 5.0
    assert np.isclose(C[L.FD_C_MAX_PITCH_CMD_RAD], np.radians(10)) and C[L.FD_C_GUIDANCE_TYPE] == L.FD_GUIDANCE_PP
    assert cfgmod.cascade_consts(guidance_type="whatever")[L.FD_C_GUIDANCE_TYPE] == L.FD_GUIDANCE_DEFAULT
    with pytest.raises(ValueError):
        cfgmod.waypoint_table([])                                                 # mission_planner.py:60-61


def test_reset_pool_follows_reference_stream_order():
    pool = samplers.presample_reset_pool([42, 43], 3, "easy", "step")
    assert pool.shape == (2, 3, L.FD_NR)
    # RateControlEnv(easy, step, rng_seed=42).reset(seed=42) -> this command and these initial conditions
    assert np.allclose(pool[0, 0, L.FD_R_CMD0:L.FD_R_CMD2 + 1], [-0.90996233, -0.79713236, -0.34282152], atol=1e-8)
    assert np.allclose(pool[0, 0, [L.FD_R_AIRSPEED, L.FD_R_ALTITUDE, L.FD_R_ROLL, L.FD_R_PITCH, L.FD_R_YAW]],
                       [20.6181, 192.6071, 0.1214717, 0.05165746, 0.980294], rtol=1e-6)
    ec = samplers.env_consts("easy", 10.0, 0.02, "step")
    assert ec[L.FD_EC_MAX_STEPS] == 500 and ec[L.FD_EC_DIFFICULTY_SCALE] == 0.3


@pytest.mark.parametrize("kx,kh", [(128, 256), (128, 0), (256, 0), (128, 128)])
def test_lstm_mfma_weight_stream_stays_inside_w(kx, kh):
    """Host-side replay of the weight-chunk address arithmetic of csrc/lstm_mfma.hip (FD_ORIGIN + FD_LOFF): every 16-byte
    vector any thread fetches must lie inside W[4H][K].  (An earlier revision read gate 3 + 2 = 5 for the zero-state layers:
    harmless values, but a memory fault whenever W sat at the end of a mapping.)"""
    H, K, threads, nslice = 256, kx + kh, 256, 32
    nchunk = (K + 191) // 192
    kc = K // nchunk
    vec_per_row = kc // 8
    nv = 2 * nslice * vec_per_row // threads
    tid = np.arange(threads)
    worst = 0
    for i in range(nv):
        v = tid + i * threads
        row = v // vec_per_row
        loff = ((row >> 5) * 2 * H + (row & 31)) * K + (v % vec_per_row) * 8
        for sl in range(H // nslice):
            for p in range(2):
                for ch in range(nchunk):
                    origin = (p * H + sl * nslice) * K + ch * kc
                    worst = max(worst, int((origin + loff).max()) + 8)
    assert worst <= 4 * H * K, (worst, 4 * H * K)


def test_alias_imports_are_the_same_module_objects():
    """`hcrl_amd.x` and the real package's `x` must be ONE module (one copy of every class / enum / library handle)."""
    import importlib
    import sys
    from hcrl_amd.flight_types import ControlMode as A
    real = "hybrid-classical-and-reinforcement-learning-aircraft-controllers_amd"
    for sub in ("flight_types", "config", "layout", "samplers", "_lib"):
        importlib.import_module(f"hcrl_amd.{sub}")
        assert sys.modules[f"hcrl_amd.{sub}"] is importlib.import_module(f"{real}.{sub}"), sub
    assert A is importlib.import_module(f"{real}.flight_types").ControlMode


def test_both_reference_yaml_schemas_normalize_to_the_trainer_schema():
    """train_rate.py-style files with omitted sections and train_overnight.py-style files (network / parallel / logging /
    checkpointing / approach / demonstrations / behavior_cloning) end up with the keys hcrl_amd.train_rate reads."""
    import yaml
    from hcrl_amd.training_utils import normalize_config
    from hcrl_amd.ppo import PPOConfig
    a = normalize_config(yaml.safe_load("""
environment: {difficulty: medium, episode_length: 10.0, dt: 0.02, command_type: step}
training: {total_timesteps: 1000000, n_envs: 8, eval_freq: 50000, save_freq: 200000, log_interval: 10}
curriculum: {enabled: true, phases: [{name: medium, difficulty: medium, timesteps: 500000, command_type: step}]}
ppo: {learning_rate: 3.0e-4, n_steps: 1024, batch_size: 256, n_epochs: 5, gamma: 0.99}
lstm: {enabled: false}
mlp: {net_arch: [128, 128]}
paths: {model_save_dir: m, tensorboard_log: t, best_model_path: b}
"""))
    assert a["lstm"] == {"enabled": False, "lstm_hidden_size": 256, "n_lstm_layers": 2, "features_dim": 128}
    assert a["mlp"]["net_arch"] == [128, 128] and a["curriculum"]["enabled"] and "imitation" not in a
    b = normalize_config(yaml.safe_load("""
approach: {use_imitation: true, use_residual: false, use_curriculum: true}
demonstrations: {n_episodes: 500, difficulty: easy, save_path: x.pkl}
behavior_cloning: {epochs: 20, batch_size: 256, learning_rate: 0.001}
curriculum:
  phases:
    - {name: easy, difficulty: easy, timesteps: 3000000, command_type: step}
    - {name: hard_mixed, difficulty: hard, timesteps: 8000000, command_type: random}
environment: {episode_length: 10.0, dt: 0.02}
ppo: {learning_rate: 0.0003, n_steps: 2048, batch_size: 256, n_epochs: 10, gamma: 0.99, gae_lambda: 0.95, clip_range: 0.2,
      ent_coef: 0.01, vf_coef: 0.5, max_grad_norm: 0.5}
network: {type: mlp, mlp: {net_arch: [256, 256, 128]}, lstm: {hidden_size: 256, n_layers: 2}}
parallel: {n_envs: 8, vec_env_type: subproc}
evaluation: {eval_freq: 100000, n_eval_episodes: 10, deterministic: true}
checkpointing: {save_freq: 500000, keep_last_n: 5}
paths: {model_dir: md, tensorboard_log: tb, best_model: best}
logging: {log_interval: 10, verbose: 1}
seed: 42
"""))
    assert b["lstm"]["enabled"] is False and b["mlp"]["net_arch"] == [256, 256, 128] and b["training"]["n_envs"] == 8
    assert b["training"]["total_timesteps"] == 11000000 and b["curriculum"]["enabled"] and b["environment"]["difficulty"] == "easy"
    assert b["training"]["save_freq"] == 500000 and b["training"]["eval_freq"] == 100000 and b["paths"]["model_save_dir"] == "md"
    assert b["imitation"] == {"n_episodes": 500, "difficulty": "easy", "save_path": "x.pkl", "epochs": 20, "batch_size": 256,
                              "learning_rate": 0.001}
    assert PPOConfig.from_dict(b["ppo"]).n_steps == 2048


def test_fe_weight_image_layout():
    """policy.pack_fe_weights lays the features extractor's weights out as csrc/policy_fe64.hip streams them: chunk order, row
    padding, 1 KB chunk granularity, the k order inside blocks of 16 (replayed here from the kernel's constants) and the LSTM
    rows scaled by their gate's exponent factor before the rounding to bf16."""
    import torch
    from hcrl_amd.policy import pack_fe_weights, _KPERM16, FE_GATE_SCALE
    torch.manual_seed(0)
    bf = torch.bfloat16
    w_emb, w1, w2, wp = (torch.randn(128, 18).to(bf).float(), torch.randn(1024, 128).to(bf).float(),
                         torch.randn(1024, 256).to(bf).float(), torch.randn(128, 256).to(bf).float())
    img = pack_fe_weights(w_emb, w1, w2, wp).float()
    rowb = lambda K: 2 * K + 16                       # noqa: E731
    pieces = lambda rows, K: (rows * rowb(K) + 1023) // 1024      # noqa: E731
    off_a = pieces(128, 32)
    off_b = off_a + 8 * (pieces(64, 128) + pieces(32, 128))
    off_c = off_b + 8 * (pieces(64, 256) + pieces(32, 256))
    assert img.numel() * 2 == (off_c + 4 * pieces(32, 256)) * 1024 == 702464

    sc = lambda w, gate: (w * FE_GATE_SCALE[gate]).to(bf).float().item()      # noqa: E731

    def at(piece_off, K, row, k):                     # element (row, k-slot) of the chunk that starts at `piece_off`
        return img[piece_off * 512 + row * (K + 8) + k].item()
    assert at(0, 32, 5, 7) == w_emb[5, 7].item() and at(0, 32, 5, 20) == 0.0
    s = 3                                             # layer 1, slice 3: (i, g) chunk then o chunk
    o1 = off_a + s * (pieces(64, 128) + pieces(32, 128))
    for slot in (0, 5, 9, 127):
        k = 16 * (slot // 16) + _KPERM16[slot % 16]
        assert at(o1, 128, 2, slot) == sc(w1[32 * s + 2, k], 0)                     # gate i
        assert at(o1, 128, 32 + 2, slot) == sc(w1[512 + 32 * s + 2, k], 2)          # gate g
        assert at(o1 + pieces(64, 128), 128, 2, slot) == sc(w1[768 + 32 * s + 2, k], 3)   # gate o
    o2 = off_b + 7 * (pieces(64, 256) + pieces(32, 256))
    assert at(o2, 256, 40, 200) == sc(w2[512 + 224 + 8, 16 * 12 + _KPERM16[8]], 2)
    assert at(off_c + 2 * pieces(32, 256), 256, 31, 37) == wp[64 + 31, 32 + _KPERM16[5]].item()
