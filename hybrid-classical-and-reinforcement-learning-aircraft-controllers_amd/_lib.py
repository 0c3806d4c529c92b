"""ctypes binding of csrc/libfdyn_hip.so (the C-ABI in include/fdyn.h).

There is NO CPU fallback: if the HIP library is missing or a GPU is not present, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FDYN_LIB", os.path.join(_HERE, "csrc", "libfdyn_hip.so"))   # FDYN_LIB: A/B experiment builds

FDYN_OK, FDYN_ERR_BAD_DT, FDYN_ERR_BAD_TYPES, FDYN_ERR_BAD_SIZE, FDYN_ERR_NULL = 0, -1, -2, -3, -4

_p = C.c_void_p
_i, _i64, _u64, _d, _f = C.c_int, C.c_int64, C.c_uint64, C.c_double, C.c_float

_SIXDOF = [_p, _p, _p, _p, _i, _i64, _d, _i, _p, _p]
_CASCADE = [_p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i64, _d, _i, _p, _p, _p]
_ENV_RESET = [_p, _p, _p, _p, _p, _p, _p, _i, _u64, _p, _i64, _p]
_ENV_STEP = [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _u64, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p,
             _i, _i64, _p]

SIGNATURES = {
    "fdyn_abi_version": (_i, []),
    "fdyn_num_substeps": (_i, [_d, _d]),
    "fdyn_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _i]),
    "fdyn_event_capacity": (_i64, [_i64]),
    "fdyn_sixdof_step_f64": (_i, _SIXDOF), "fdyn_sixdof_step_mixed": (_i, _SIXDOF), "fdyn_sixdof_step_f32": (_i, _SIXDOF),
    "fdyn_derived_f64": (_i, [_p, _i64, _p, _p]), "fdyn_derived_f32": (_i, [_p, _i64, _p, _p]),
    "fdyn_pid_compute_batch": (_i, [_p, _i, _p, _p, _p, _f, _p, _i64, _p]),
    "fdyn_cascade_step_f64": (_i, _CASCADE), "fdyn_cascade_step_mixed": (_i, _CASCADE), "fdyn_cascade_step_f32": (_i, _CASCADE),
    "fdyn_rate_env_reset_f64": (_i, _ENV_RESET), "fdyn_rate_env_reset_mixed": (_i, _ENV_RESET),
    "fdyn_rate_env_reset_f32": (_i, _ENV_RESET),
    "fdyn_rate_env_step_f64": (_i, _ENV_STEP), "fdyn_rate_env_step_mixed": (_i, _ENV_STEP),
    "fdyn_rate_env_step_f32": (_i, _ENV_STEP),
    "fdyn_lstm_cell_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "fdyn_lstm_cell_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "fdyn_lstm_seq_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _i64, _i64, _i, _p]),
    "fdyn_lstm_seq_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _i64, _i, _p]),
    "fdyn_lstm_seq_bwd_bsum": (_i, [_p, _i, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i, _p]),
    "fdyn_lstm_seq_bwd_pre": (_i, [_p, _i, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _i64, _i64, _i, _p]),
    "fdyn_lstm_cell0_fwd": (_i, [_p, _i, _p, _p, _i64, _i, _p]),
    "fdyn_lstm_cell0_bwd": (_i, [_p, _i, _p, _p, _p, _i64, _i64, _i, _p]),
    "fdyn_colsum_partials": (_i, [_p, _i64, _i, _p, _p, _p]),
    "fdyn_episode_flags": (_i, [_p, _p, _p, _p, _p, _i64, _p]),
    "fdyn_policy_trunks": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]),
    "fdyn_policy_trunks_heads": (_i, [_p] * 11 + [_u64, _p, _i, _p, _p, _p, _i64, _p]),
    "fdyn_agent_step_f64": (_i, [_i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i64, _d, _i, _p, _p]),
    "fdyn_agent_step_mixed": (_i, [_i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i64, _d, _i, _p, _p]),
    "fdyn_agent_step_f32": (_i, [_i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i64, _d, _i, _p, _p]),
    "fdyn_colsum_ws_floats": (_i64, [_i64, _i, _i64, _i]),
    "fdyn_colsum": (_i, [_p, _i, _i64, _i, _i64, _i, _p, _p, _p]),
    "fdyn_ppo_loss": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _f, _f, _i64, _p, _p, _p, _p, _p]),
    "fdyn_lstm_cell_mfma": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p]),
    "fdyn_lstm_cell_mfma_inplace_ok": (_i, [_i, _i, _i, _i64]),
    "fdyn_lstm_cell_mfma_pair": (_i, [_p, _i, _p, _i, _i64, _i] + [_p] * 13),
    "fdyn_lstm_cell_mfma_train": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _i64, _i, _p]),
    "fdyn_policy_recurrent_image_bytes": (_i, []),
    "fdyn_policy_recurrent": (_i, [_p] * 12 + [_i64, _p]),
    "fdyn_policy_features_image_bytes": (_i, []),
    "fdyn_policy_features": (_i, [_p, _p, _p, _p, _i64, _p]),
    "fdyn_policy_features_flags": (_i, [_p] * 9 + [_i64, _p]),
    "fdyn_gaussian_head": (_i, [_p, _i, _p, _u64, _p, _i, _p, _p, _i64, _p]),
    "fdyn_policy_heads": (_i, [_p, _p, _p, _p, _p, _p, _p, _u64, _p, _i, _p, _p, _p, _i64, _p]),
    "fdyn_gae": (_i, [_p, _p, _p, _p, _p, _f, _f, _i, _i64, _p, _p, _p]),
    "fdyn_rate_metrics_f64": (_i, [_p, _p, _p, _p, _p, _p, _d, _i, _i, _i64, _p, _p]),
    "fdyn_rate_reward_seq_f64": (_i, [_p, _p, _p, _p, _p, _p, _p, _d, _i, _i64, _p, _p, _p, _p, _p]),
    "fdyn_rate_reward_seq_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _i, _i64, _p, _p, _p, _p, _p]),
    "fdyn_sensor_update_f64": (_i, [_p, _p, _p, _p, _p, _u64, _p, _p, _i64, _p]),
    "fdyn_sensor_update_f32": (_i, [_p, _p, _p, _p, _p, _u64, _p, _p, _i64, _p]),
    "fdyn_sensor_observe": (_i, [_p, _p, _p, _p, _p, _u64, _p, _i64, _p]),
    "fdyn_rate_metrics_f32": (_i, [_p, _p, _p, _p, _p, _p, _d, _i, _i, _i64, _p, _p]),
}

_lib = None


class FdynError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FdynError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). The batched flight-dynamics path has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header/library drift
            fn.restype, fn.argtypes = res, args
        if lib.fdyn_abi_version() != 2:
            raise FdynError("libfdyn_hip.so ABI version mismatch")
        _lib = lib
    return _lib


_SYNC_DEBUG = bool(os.environ.get("FDYN_SYNC_DEBUG"))


def check(rc, what="fdyn call"):
    if _SYNC_DEBUG:                       # debugging aid: localise an asynchronous GPU fault to one launch
        import sys
        import torch
        print(f"[fdyn] {what} rc={rc} ...", end="", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print(" done", file=sys.stderr, flush=True)
    if rc == FDYN_OK:
        return
    if rc == FDYN_ERR_BAD_DT:
        # the reference raises ValueError for dt outside (min_timestep, max_timestep] (simplified_6dof.py:241-245)
        raise ValueError(f"{what}: invalid timestep, must be in (1e-06, 1.0]")
    names = {FDYN_ERR_BAD_TYPES: "n_types outside 1..8", FDYN_ERR_BAD_SIZE: "bad size argument",
             FDYN_ERR_NULL: "required pointer is NULL"}
    raise FdynError(f"{what}: {names.get(rc, 'HIP error ' + str(rc))}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C-ABI arrays must be contiguous"
    return t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise FdynError("no GPU visible: the batched flight-dynamics path runs only on the HIP device (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


PRECISIONS = ("f64", "mixed", "f32")


def state_dtype(precision):
    import torch
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {PRECISIONS}")
    return torch.float32 if precision == "f32" else torch.float64
