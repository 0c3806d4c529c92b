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


# ---- weight gradients of tall-skinny linears ---------------------------------------------------------------------------
# dW = dY^T X with M = T*B rows (10^5..10^6) and a small [N, K] result is a handful of output tiles: left to the library's
# default heuristics it runs on 16-48 of the 256 CUs (rocprofv3: 125-150 TFLOP/s, 40 % of a PPO iteration).  Splitting the
# row dimension into S batched GEMMs fills the chip; the S partial results are summed in fp32.
_BMM_F32_OUT = None


def wgrad_splitk(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dy [M, N], x [M, K] (same dtype) -> dy^T x as fp32 [N, K]."""
    global _BMM_F32_OUT
    M, N = dy.shape
    K = x.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    S = 1
    while S * tiles < 1024 and M % (2 * S) == 0 and M // (2 * S) >= 2048:
        S *= 2
    if S == 1:
        return (dy.t() @ x).float()
    a, b = dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)
    if dy.dtype != torch.float32 and _BMM_F32_OUT is not False:
        try:                                              # fp32 partials straight from the MFMA accumulators
            part = torch.bmm(a, b, out_dtype=torch.float32)
            _BMM_F32_OUT = True
            return part.sum(0)
        except (TypeError, RuntimeError):
            _BMM_F32_OUT = False
    return torch.bmm(a, b).sum(0, dtype=torch.float32)


class _SplitKLinearFn(torch.autograd.Function):
    """F.linear whose weight gradient is the split-K batched GEMM above (bias gradient: one fp32 column reduction)."""

    @staticmethod
    def forward(ctx, x, w, b):
        y = torch.nn.functional.linear(x, w, b)            # under autocast: bf16 GEMM with the bias in the epilogue
        ctx.save_for_backward(x if x.dtype == y.dtype else x.to(y.dtype), w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = (dy2 @ w.to(dy.dtype)).view_as(x) if ctx.needs_input_grad[0] else None
        dw = wgrad_splitk(dy2.contiguous(), x2.contiguous()).to(w.dtype) if ctx.needs_input_grad[1] else None
        db = dy2.sum(0, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], min_rows: int = 8192) -> torch.Tensor:
    """`F.linear` for the training path: tall inputs on the GPU take the split-K weight gradient."""
    rows = x.numel() // max(x.shape[-1], 1)
    if x.is_cuda and rows >= min_rows and torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
        return _SplitKLinearFn.apply(x, w, b)
    return torch.nn.functional.linear(x, w, b)


class DeferredWgrad:
    """Weight gradient of a linear applied once per time step of a recurrence (BPTT): every step's backward stores its
    dY in slot t and returns only dX; when the backward pass ends, ONE split-K batched GEMM over all T*B rows gives
    dW = sum_t dY_t^T X_t and one reduction gives db -- instead of T small GEMMs + T bias reductions + 2T accumulations.
    `sinks`: [(parameter, column slice or None)] receiving dW (column ranges of the fused [W_ih | W_hh]); `bias_sinks`:
    parameters receiving db."""

    def __init__(self, T: int, B: int, K: int, N: int, dtype, device, sinks, bias_sinks):
        self.x = torch.empty((T, B, K), dtype=dtype, device=device)       # the forward writes its inputs here (no copy)
        self.dy = torch.zeros((T, B, N), dtype=dtype, device=device)
        self.sinks, self.bias_sinks = sinks, bias_sinks
        self._queued = False

    def flush(self):
        self._queued = False
        T, B, K = self.x.shape
        N = self.dy.shape[-1]
        dy = self.dy.view(T * B, N)
        dw = wgrad_splitk(dy, self.x.view(T * B, K))
        db = dy.sum(0, dtype=torch.float32)
        for p, cols in self.sinks:
            g = dw if cols is None else dw[:, cols]
            p.grad = g.to(p.dtype).clone() if p.grad is None else p.grad.add_(g.to(p.dtype))
        for p in self.bias_sinks:
            p.grad = db.to(p.dtype).clone() if p.grad is None else p.grad.add_(db.to(p.dtype))


class _DeferredLinearFn(torch.autograd.Function):
    """y = [x | h] W^T + b for step t of a recurrence; the concatenation is written straight into the bucket's slot."""

    @staticmethod
    def forward(ctx, x, h, w, b, bucket, t):
        kx = x.shape[-1]
        slot = bucket.x[t]
        slot[:, :kx].copy_(x)
        slot[:, kx:].copy_(h)
        ctx.bucket, ctx.t, ctx.kx = bucket, t, kx
        ctx.save_for_backward(w)
        return torch.nn.functional.linear(slot, w, b)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        bucket = ctx.bucket
        bucket.dy[ctx.t].copy_(dy)
        if not bucket._queued:                              # runs once, after the whole backward pass
            bucket._queued = True
            torch.autograd.Variable._execution_engine.queue_callback(bucket.flush)
        dcat = dy @ w
        return dcat[:, :ctx.kx], dcat[:, ctx.kx:], None, None, None, None


def deferred_linear(x: torch.Tensor, h: torch.Tensor, w: torch.Tensor, b: torch.Tensor, bucket: DeferredWgrad, t: int) -> torch.Tensor:
    """w [N, kx + kh], b [N] in the compute dtype (constants of this backward pass); x [B, kx], h [B, kh]."""
    return _DeferredLinearFn.apply(x, h, w, b, bucket, t)


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
