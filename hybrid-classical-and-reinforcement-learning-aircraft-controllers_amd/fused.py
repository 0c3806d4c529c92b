"""torch.autograd bindings of the fused policy kernels (csrc/policy_kernels.hip).

`lstm_cell(gates, c_prev)` = the LSTM point-wise update; on a GPU tensor it is ONE HIP launch forward and one backward,
on a CPU tensor (unit tests, gloo rehearsals) it is the same arithmetic in plain torch ops.
"""
from typing import Optional, Tuple

import torch

from . import _lib


def _lstm_cell_torch(gates, c_prev):
    i, f, g, o = gates.float().chunk(4, dim=-1)
    c = torch.sigmoid(i) * torch.tanh(g)
    if c_prev is not None:
        c = c + torch.sigmoid(f) * c_prev
    return (torch.sigmoid(o) * torch.tanh(c)).to(gates.dtype), c


class _LSTMCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gates, c_prev):
        lib = _lib.load()
        gates = gates.contiguous()
        B, H4 = gates.shape
        H = H4 // 4
        bf16 = gates.dtype == torch.bfloat16
        assert bf16 or gates.dtype == torch.float32
        if c_prev is not None:
            c_prev = c_prev.float().contiguous()
        need_grad = gates.requires_grad or (c_prev is not None and c_prev.requires_grad)
        h = torch.empty((B, H), dtype=gates.dtype, device=gates.device)
        c = torch.empty((B, H), dtype=torch.float32, device=gates.device)
        act = torch.empty_like(gates) if need_grad else None
        _lib.check(lib.fdyn_lstm_cell_fwd(gates.data_ptr(), int(bf16), _lib.ptr(c_prev), None, h.data_ptr(), c.data_ptr(),
                                          _lib.ptr(act), B, H, _lib.current_stream()), "lstm_cell_fwd")
        if need_grad:
            ctx.save_for_backward(act, c_prev, c)
            ctx.has_prev = c_prev is not None
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        lib = _lib.load()
        act, c_prev, c = ctx.saved_tensors if ctx.has_prev else (ctx.saved_tensors[0], None, ctx.saved_tensors[-1])
        B, H4 = act.shape
        H = H4 // 4
        bf16 = act.dtype == torch.bfloat16
        dh = torch.zeros((B, H), dtype=act.dtype, device=act.device) if dh is None else dh.to(act.dtype).contiguous()
        dc = None if dc is None else dc.float().contiguous()
        dgates = torch.empty_like(act)
        dc_prev = torch.empty((B, H), dtype=torch.float32, device=act.device) if ctx.has_prev else None
        _lib.check(lib.fdyn_lstm_cell_bwd(act.data_ptr(), int(bf16), _lib.ptr(c_prev), c.data_ptr(), dh.data_ptr(),
                                          _lib.ptr(dc), dgates.data_ptr(), _lib.ptr(dc_prev), B, H, _lib.current_stream()),
                   "lstm_cell_bwd")
        return dgates, dc_prev


def lstm_cell(gates: torch.Tensor, c_prev: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """gates [B, 4H] pre-activation (i, f, g, o), c_prev [B, H] fp32 or None (zero state) -> h [B, H] (gates dtype), c fp32."""
    if gates.is_cuda:
        return _LSTMCellFn.apply(gates, c_prev)
    return _lstm_cell_torch(gates, c_prev)


def gae(rewards, values, episode_starts, last_values, last_dones, gamma: float, lam: float):
    """[T, N] fp32 rollout -> advantages, returns; one HIP launch on the GPU."""
    T, N = rewards.shape
    adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
    lib = _lib.load()
    _lib.check(lib.fdyn_gae(rewards.contiguous().data_ptr(), values.contiguous().data_ptr(),
                            episode_starts.contiguous().data_ptr(), last_values.contiguous().data_ptr(),
                            last_dones.contiguous().data_ptr(), float(gamma), float(lam), T, N, adv.data_ptr(),
                            ret.data_ptr(), _lib.current_stream()), "gae")
    return adv, ret
