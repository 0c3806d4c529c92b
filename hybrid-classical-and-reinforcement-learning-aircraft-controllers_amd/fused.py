"""torch.autograd bindings of the fused policy kernels (csrc/policy_kernels.hip).

`lstm_cell(gates, c_prev)` = the LSTM point-wise update; on a GPU tensor it is ONE HIP launch forward and one backward,
on a CPU tensor (unit tests, gloo rehearsals) it is the same arithmetic in plain torch ops.
"""
import os
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
    def forward(ctx, gates, c_prev, need_c=True):
        lib = _lib.load()
        gates = gates.contiguous()
        B, H4 = gates.shape
        H = H4 // 4
        bf16 = gates.dtype == torch.bfloat16
        assert bf16 or gates.dtype == torch.float32
        if c_prev is not None:
            c_prev = c_prev.float().contiguous()
        need_grad = gates.requires_grad or (c_prev is not None and c_prev.requires_grad)
        # an unused output (the cell state of the zero-state layers) must reach backward as None, not as a materialised
        # [B, H] fp32 zero tensor: that fill + the kernel reading it back cost ~150 us per layer per slice at 524 288 rows
        ctx.set_materialize_grads(False)
        h = torch.empty((B, H), dtype=gates.dtype, device=gates.device)
        # need_c=False (zero-state cells whose caller only wants h -- the feature extractor's layers): c is neither written
        # nor kept, the backward rebuilds c = i * g from the saved gates; 18 % of each kernel's traffic at 524 288 rows
        assert need_c or c_prev is None, "only a zero-state cell can drop its cell state"
        c = torch.empty((B, H), dtype=torch.float32, device=gates.device) if need_c else None
        act = torch.empty_like(gates) if need_grad else None
        _lib.check(lib.fdyn_lstm_cell_fwd(gates.data_ptr(), int(bf16), _lib.ptr(c_prev), None, h.data_ptr(), _lib.ptr(c),
                                          _lib.ptr(act), B, H, _lib.current_stream()), "lstm_cell_fwd")
        if need_grad:
            ctx.has_prev, ctx.has_c = c_prev is not None, need_c
            ctx.save_for_backward(*([act] + ([c_prev] if ctx.has_prev else []) + ([c] if need_c else [])))
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        if dh is None and dc is None:
            return None, None, None
        lib = _lib.load()
        saved = list(ctx.saved_tensors)
        act = saved.pop(0)
        c_prev = saved.pop(0) if ctx.has_prev else None
        c = saved.pop(0) if ctx.has_c else None
        B, H4 = act.shape
        H = H4 // 4
        bf16 = act.dtype == torch.bfloat16
        dh = torch.zeros((B, H), dtype=act.dtype, device=act.device) if dh is None else dh.to(act.dtype).contiguous()
        dc = None if dc is None else dc.float().contiguous()
        dgates = torch.empty_like(act)
        dc_prev = torch.empty((B, H), dtype=torch.float32, device=act.device) if ctx.has_prev else None
        _lib.check(lib.fdyn_lstm_cell_bwd(act.data_ptr(), int(bf16), _lib.ptr(c_prev), _lib.ptr(c), dh.data_ptr(),
                                          _lib.ptr(dc), dgates.data_ptr(), _lib.ptr(dc_prev), B, H, _lib.current_stream()),
                   "lstm_cell_bwd")
        return dgates, dc_prev, None


def lstm_cell(gates: torch.Tensor, c_prev: Optional[torch.Tensor], need_c: bool = True) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """gates [B, 4H] pre-activation (i, f, g, o), c_prev [B, H] fp32 or None (zero state) -> h [B, H] (gates dtype), c fp32.
    need_c=False (zero state only): the caller wants h alone -- c comes back as None and is never stored."""
    if gates.is_cuda:
        return _LSTMCellFn.apply(gates, c_prev, need_c)
    h, c = _lstm_cell_torch(gates, c_prev)
    return h, (c if need_c else None)


# ---- weight gradients of tall-skinny linears ---------------------------------------------------------------------------
# dW = dY^T X with M = T*B rows (10^5..10^6) and a small [N, K] result is a handful of output tiles: left to the library's
# default heuristics it runs on 16-48 of the 256 CUs (rocprofv3: 125-150 TFLOP/s, 40 % of a PPO iteration).  Splitting the
# row dimension into S batched GEMMs fills the chip; the S partial results are summed in fp32.
_BMM_F32_OUT = None


def colsum(x: torch.Tensor, group_rows: int = 0, n_groups: int = 1) -> torch.Tensor:
    """x [M, N] (bf16 / fp32, contiguous) -> fp32 [N] column sums (bias gradients): hand-written HIP reduction on the GPU
    (graph-replay safe, unlike the generic multi-block reductions on this stack), plain torch on the CPU.
    n_groups > 1: rows come in segments of `group_rows`, segment j belonging to group j % n_groups -> [n_groups, N]."""
    M, N = x.shape
    if not x.is_cuda:
        if n_groups <= 1:
            return x.sum(0, dtype=torch.float32)
        return x.view(-1, n_groups, group_rows, N).sum((0, 2), dtype=torch.float32)
    x = x.contiguous()
    lib = _lib.load()
    out = torch.empty((n_groups, N) if n_groups > 1 else (N,), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.fdyn_colsum_ws_floats(M, N, group_rows, n_groups)), dtype=torch.float32, device=x.device)
    _lib.check(lib.fdyn_colsum(x.data_ptr(), int(x.dtype == torch.bfloat16), M, N, group_rows, n_groups, out.data_ptr(),
                               ws.data_ptr(), _lib.current_stream()), "colsum")
    return out


def _sum_parts(part: torch.Tensor) -> torch.Tensor:
    """part [S, N, K] fp32 -> sum over S as a [1, S] x [S, N*K] GEMM (no generic reduction kernel)."""
    S, N, K = part.shape
    ones = torch.ones((1, S), dtype=part.dtype, device=part.device)
    return torch.mm(ones, part.view(S, N * K)).view(N, K)


def _bmm_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Batched GEMM with fp32 results straight from the MFMA accumulators where the build supports it."""
    global _BMM_F32_OUT
    if a.dtype != torch.float32 and _BMM_F32_OUT is not False:
        try:
            out = torch.bmm(a, b, out_dtype=torch.float32)
            _BMM_F32_OUT = True
            return out
        except (TypeError, RuntimeError):
            _BMM_F32_OUT = False
    return torch.bmm(a, b).float()


def wgrad_splitk(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dy [M, N], x [M, K] (same dtype) -> dy^T x as fp32 [N, K]."""
    M, N = dy.shape
    K = x.shape[1]
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    S = 1
    while S * tiles < 1024 and M % (2 * S) == 0 and M // (2 * S) >= 2048:
        S *= 2
    if S == 1:
        return (dy.t() @ x).float()
    return _sum_parts(_bmm_f32(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)))


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
        db = colsum(dy2) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], min_rows: int = 1) -> torch.Tensor:
    """`F.linear` for the training path: on the GPU the weight gradient is the split-K GEMM (one plain GEMM for short inputs) and
    the bias gradient OUR column sum.  min_rows = 1, i.e. always: the framework's own bias-gradient reduction is one of the
    reductions that return garbage from the second replay of a captured graph on this stack -- with the
    former threshold of 8192 rows a SMALL training configuration (512 envs x 8 steps in 4 slices) trained on garbage bias
    gradients under the update graph and went to NaN in its third iteration (scratch/dbg_graph_grads.py compares replayed and
    eager gradients parameter by parameter; tests/test_gpu_training.py::test_small_update_graph_replays_match_eager pins it)."""
    rows = x.numel() // max(x.shape[-1], 1)
    if x.is_cuda and rows >= min_rows and torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
        return _SplitKLinearFn.apply(x, w, b)
    return torch.nn.functional.linear(x, w, b)


# Which BPTT forwards take the fused MFMA cell (fdyn_lstm_cell_mfma_train).  Measured on MI355X at the bench's train workload
# (65 536 envs, 16 steps, 2 x 2 slices; PPO iteration, 8 timed iterations each): "0" 56.6-56.9 ms, "fe" 55.9, "seq" 58.0,
# "all" 56.9-57.1 -- the fused kernel moves its 1.5x larger output (activated gates, packed next input) at the ~2.5 TB/s its
# 64-byte row segments reach, which is what hipBLASLt's GEMM + the 5.4 TB/s point-wise kernel take together; for the recurrent
# steps, where the actor's and the critic's un-fused chains overlap in the graph, it loses.  Hence "fe".
MFMA_TRAIN_DEFAULT = "fe"


def _mfma_train_ok(t: torch.Tensor, kx: int, kh: int, H: int) -> bool:
    """The BPTT forward takes the fused MFMA cell (csrc/lstm_mfma.hip, TRAIN) for the shapes that kernel is built for."""
    mode = os.environ.get("FDYN_MFMA_TRAIN", MFMA_TRAIN_DEFAULT)        # "all", "seq" (recurrent cells), "fe" (zero-state layers), "0"
    if mode not in ("all", "seq" if kh else "fe"):
        return False
    return (t.is_cuda and t.dtype == torch.bfloat16 and H % 32 == 0 and (kx, kh) in ((128, 256), (128, 0), (256, 0), (128, 128))
            and (kh == 0 or kh == H))


def _bsum_rows_per_block(rows: int, H: int) -> int:
    """Rows a block of the bias-summing backward kernels owns (0 = this H cannot keep one column per lane)."""
    hv = H // 8
    if H % 8 or hv == 0 or 256 % hv:
        return 0
    return (256 // hv) * (16 if rows >= 262144 else 4)


def colsum_partials(partial: torch.Tensor) -> torch.Tensor:
    """partial [nb, N] fp32 (one row per block of a bias-summing backward kernel) -> [N] column sums, no atomics."""
    nb, N = partial.shape
    lib = _lib.load()
    out = torch.empty(N, dtype=torch.float32, device=partial.device)
    mid = torch.empty(((nb + 63) // 64, N), dtype=torch.float32, device=partial.device)
    _lib.check(lib.fdyn_colsum_partials(partial.data_ptr(), nb, N, out.data_ptr(), mid.data_ptr(), _lib.current_stream()),
               "colsum_partials")
    return out


class _ZeroStateLayerFn(torch.autograd.Function):
    """One LSTM layer of the features extractor, h = sigmoid(o) tanh(sigmoid(i) tanh(g)) with [i g o] = x W3^T + b3, as ONE
    autograd node in the THREE-gate layout: the reference runs these layers from h = c = 0 on every call
    (learned_controllers/networks/lstm_policy.py:75-92), so the forget gate multiplies zero -- its rows of W_ih, its
    pre-activation, its saved activation and its (exactly zero) gradient are never formed: a quarter less GEMM work (forward,
    dX and dW) and point-wise traffic.  Backward: one point-wise launch that also leaves the bias gradient as per-block partial
    sums (no second pass over [M, 3H]), dX GEMM, split-K dW GEMM; the f rows of dW_ih / db come back as zeros."""

    @staticmethod
    def forward(ctx, x, w_ih, b_ih, b_hh):
        lib = _lib.load()
        H = w_ih.shape[0] // 4
        dt = x.dtype
        bf16 = dt == torch.bfloat16
        assert bf16 or dt == torch.float32
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        M = x2.shape[0]
        need = any(ctx.needs_input_grad)
        h = torch.empty((M, H), dtype=dt, device=x.device)
        if need and _mfma_train_ok(x2, x2.shape[1], 0, H):
            # GEMM + gate non-linearities + cell update in ONE MFMA kernel: the [M, 3H] pre-activations never exist in HBM
            with torch.autocast("cuda", enabled=False):
                w4 = w_ih.to(dt).contiguous()
                w3 = torch.cat([w4[:H], w4[2 * H:]])                                          # rows of i, g, o (backward GEMMs)
                b32 = (b_ih + b_hh).float().contiguous()
            gates = torch.empty((M, 3 * H), dtype=dt, device=x.device)
            _lib.check(lib.fdyn_lstm_cell_mfma_train(x2.data_ptr(), x2.shape[1], None, 0, None, None, w4.data_ptr(), b32.data_ptr(),
                                                     h.data_ptr(), None, gates.data_ptr(), None, 0, None, M, H,
                                                     _lib.current_stream()), "lstm_cell_mfma_train")
        else:
            with torch.autocast("cuda", enabled=False):
                w3 = torch.cat([w_ih[:H], w_ih[2 * H:]]).to(dt)                               # rows of i, g, o
                b = b_ih + b_hh
                b3 = torch.cat([b[:H], b[2 * H:]]).to(dt)
                gates = torch.addmm(b3, x2, w3.t())                                           # [M, 3H], bias in the epilogue
            # the activated gates overwrite the pre-activations in place
            _lib.check(lib.fdyn_lstm_cell0_fwd(gates.data_ptr(), int(bf16), h.data_ptr(), gates.data_ptr() if need else None, M, H,
                                               _lib.current_stream()), "lstm_cell0_fwd")
        if need:
            ctx.save_for_backward(x2, w3, gates)
            ctx.H, ctx.xshape, ctx.dtypes = H, x.shape, (w_ih.dtype, b_ih.dtype, b_hh.dtype)
        return h.view(*x.shape[:-1], H)

    @staticmethod
    def backward(ctx, dh):
        lib = _lib.load()
        x2, w3, act = ctx.saved_tensors
        H, M, dt, dev = ctx.H, act.shape[0], act.dtype, act.device
        bf16 = dt == torch.bfloat16
        dh = dh.reshape(M, H).to(dt).contiguous()
        rpb = _bsum_rows_per_block(M, H)
        ws = torch.empty(((M + rpb - 1) // rpb, 3 * H), dtype=torch.float32, device=dev) if rpb else None
        _lib.check(lib.fdyn_lstm_cell0_bwd(act.data_ptr(), int(bf16), dh.data_ptr(), act.data_ptr(), _lib.ptr(ws), rpb, M, H,
                                           _lib.current_stream()), "lstm_cell0_bwd")
        dg = act                                                                              # dgates, in place
        needs = ctx.needs_input_grad
        with torch.autocast("cuda", enabled=False):
            dx = (dg @ w3).view(ctx.xshape) if needs[0] else None
            dw = db = None
            if needs[1]:
                dw3 = wgrad_splitk(dg, x2)                                                    # fp32 [3H, K]
                dw = torch.zeros((4 * H, x2.shape[1]), dtype=torch.float32, device=dev)
                dw[:H].copy_(dw3[:H]); dw[2 * H:].copy_(dw3[H:])
                dw = dw.to(ctx.dtypes[0])
            if needs[2] or needs[3]:
                db3 = colsum_partials(ws) if rpb else colsum(dg)
                db = torch.zeros(4 * H, dtype=torch.float32, device=dev)
                db[:H].copy_(db3[:H]); db[2 * H:].copy_(db3[H:])
        return dx, dw, (db.to(ctx.dtypes[1]) if needs[2] else None), (db.to(ctx.dtypes[2]) if needs[3] else None)


def zero_state_lstm_layer(x: torch.Tensor, w_ih: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor) -> torch.Tensor:
    """x [..., K] -> h [..., H] of an nn.LSTM layer stepped once from zero state (the features extractor's layers)."""
    rows = x.numel() // max(x.shape[-1], 1)
    if x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and rows >= linear.__defaults__[0]:
        return _ZeroStateLayerFn.apply(x, w_ih, b_ih, b_hh)
    return lstm_cell(linear(x, w_ih, b_ih + b_hh), None, need_c=False)[0]       # four-gate layout: CPU, small batches


class _LinearReLUFn(torch.autograd.Function):
    """relu(x W^T + b) with the bias and the ReLU in the GEMM epilogue (hipBLASLt through torch._addmm_activation), split-K
    weight gradient, ReLU mask applied to dY in one launch."""

    @staticmethod
    def forward(ctx, x, w, b, dt):
        x2 = x.reshape(-1, x.shape[-1]).to(dt).contiguous()
        with torch.autocast("cuda", enabled=False):
            wd = w.to(dt)
            y = torch._addmm_activation(b.to(dt), x2, wd.t())
        ctx.save_for_backward(x2, wd, y)
        ctx.xshape, ctx.dtypes = x.shape, (x.dtype, w.dtype, b.dtype)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wd, y = ctx.saved_tensors
        needs = ctx.needs_input_grad
        with torch.autocast("cuda", enabled=False):
            dy2 = torch.ops.aten.threshold_backward(dy.reshape(y.shape).to(y.dtype).contiguous(), y, 0)
            dx = (dy2 @ wd).view(ctx.xshape).to(ctx.dtypes[0]) if needs[0] else None
            dw = wgrad_splitk(dy2, x2).to(ctx.dtypes[1]) if needs[1] else None
            db = colsum(dy2).to(ctx.dtypes[2]) if needs[2] else None
        return dx, dw, db, None


def linear_relu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, min_rows: Optional[int] = None) -> torch.Tensor:
    """relu(F.linear(x, w, b)) for the training path; tall GPU inputs take the fused epilogue + split-K weight gradient
    (min_rows: as fused.linear, whose default it follows)."""
    rows = x.numel() // max(x.shape[-1], 1)
    min_rows = linear.__defaults__[0] if min_rows is None else min_rows
    if x.is_cuda and b is not None and rows >= min_rows and torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if dt in (torch.bfloat16, torch.float32) and (dt != torch.float32 or (w.dtype == dt and x.dtype == dt)):
            return _LinearReLUFn.apply(x, w, b, dt)
    return torch.relu(linear(x, w, b, min_rows))


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
        db = colsum(dy)
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


class _LSTMSequenceFn(torch.autograd.Function):
    """G recurrent LSTM cells that read the same input sequence (the policy's actor and critic cells), over T steps, as ONE
    autograd node (BPTT inside).  The cells ride a leading group dimension: per step and direction ONE batched GEMM and ONE
    fused point-wise launch over G*B rows (fdyn_lstm_seq_fwd / _bwd: masks, recurrent add, next-input packing folded in);
    the weight gradients are one batched split-K GEMM over all (t, g) blocks, the bias gradients one grouped column sum."""

    @staticmethod
    def forward(ctx, feats, keep, h0, c0, *params):
        lib = _lib.load()
        G = len(params) // 4
        T, B, kx = feats.shape
        H = params[1].shape[1]
        K, dt, dev = kx + H, feats.dtype, feats.device
        bf16 = dt == torch.bfloat16
        assert bf16 or dt == torch.float32
        assert h0.shape == (G, B, H) and c0.shape == (G, B, H) and keep.shape == (T, B)
        for g in range(G):
            assert params[4 * g].shape == (4 * H, kx) and params[4 * g + 1].shape == (4 * H, H)
        w = torch.stack([torch.cat([params[4 * g], params[4 * g + 1]], 1) for g in range(G)]).to(dt)          # [G, 4H, K]
        b = torch.stack([params[4 * g + 2] + params[4 * g + 3] for g in range(G)]).to(dt).contiguous()      # [G, 4H]
        wT = w.transpose(1, 2)
        keep = keep.float().contiguous()
        keep_rows = keep.repeat(1, G).contiguous()                           # [T, G*B]: row g*B + b -> keep[t, b]
        need = ctx.needs_input_grad[0] or any(ctx.needs_input_grad[4:])
        x_all = torch.empty((T, G, B, K), dtype=dt, device=dev)
        x_all[..., :kx].copy_(feats.unsqueeze(1))
        x_all[0, :, :, kx:].copy_(h0.to(dt) * keep[0].view(1, B, 1).to(dt))
        act = torch.empty((T, G, B, 4 * H), dtype=dt, device=dev) if need else None
        c_all = torch.empty((T + 1, G, B, H), dtype=torch.float32, device=dev)
        c_all[0].copy_(c0)
        h_seq = torch.empty((T, G, B, H), dtype=dt, device=dev)
        st, R, esz = _lib.current_stream(), G * B, x_all.element_size()
        if need and _mfma_train_ok(feats, kx, H, H):
            # every step of every cell is ONE MFMA kernel (GEMM + gate non-linearities + cell update + the activated gates for
            # the backward pass + the next step's recurrent input columns): no [B, 4H] pre-activations in HBM
            wc, b32 = w.contiguous(), torch.stack([params[4 * g + 2] + params[4 * g + 3] for g in range(G)]).float().contiguous()
            h0c = h0.to(dt).contiguous()
            for t in range(T):
                last = t == T - 1
                for g in range(G):
                    hp = h0c[g] if t == 0 else h_seq[t - 1, g]
                    _lib.check(lib.fdyn_lstm_cell_mfma_train(
                        feats[t].data_ptr(), kx, hp.data_ptr(), H, c_all[t, g].data_ptr(), keep[t].data_ptr(), wc[g].data_ptr(),
                        b32[g].data_ptr(), h_seq[t, g].data_ptr(), c_all[t + 1, g].data_ptr(), act[t, g].data_ptr(),
                        None if last else x_all[t + 1, g].data_ptr() + kx * esz, K, None if last else keep[t + 1].data_ptr(),
                        B, H, st), "lstm_cell_mfma_train")
            T_done = T
        else:
            T_done = 0
        # un-fused steps: the GEMM writes its pre-activations straight into the slot the backward pass reads (act[t]); the
        # point-wise kernel does not re-write them as activations (8 of its 28 bytes per hidden unit) -- the backward kernel
        # re-evaluates the non-linearities from the same rounded values (fdyn_lstm_seq_bwd_pre)
        pre = need and T_done == 0 and not os.environ.get("FDYN_NO_PRE")
        for t in range(T_done, T):
            gates = torch.bmm(x_all[t], wT, out=act[t]) if pre else torch.bmm(x_all[t], wT)      # [G, B, 4H]; bias: in the cell kernel
            last = t == T - 1
            _lib.check(lib.fdyn_lstm_seq_fwd(gates.data_ptr(), int(bf16), c_all[t].data_ptr(), keep_rows[t].data_ptr(),
                                             h_seq[t].data_ptr(), c_all[t + 1].data_ptr(), act[t].data_ptr() if (need and not pre) else None,
                                             None if last else x_all[t + 1].data_ptr() + kx * esz, K,
                                             None if last else keep_rows[t + 1].data_ptr(), b.data_ptr(), B, R, H, st),
                       "lstm_seq_fwd")
        if need:
            ctx.save_for_backward(x_all, act, c_all, keep_rows, w, b)
            ctx.kx, ctx.G, ctx.param_dtypes, ctx.pre = kx, G, tuple(p.dtype for p in params), pre
        ctx.mark_non_differentiable(c_all)
        ctx.set_materialize_grads(False)          # no [T+1, G, B, H] zero tensor for the state output nobody differentiates
        return h_seq, c_all

    @staticmethod
    def backward(ctx, dh_seq, _dc_all):
        if dh_seq is None:
            return (None,) * (4 + 4 * ctx.G)
        lib = _lib.load()
        x_all, act, c_all, keep_rows, w, b = ctx.saved_tensors
        T, G, B, K = x_all.shape
        H, kx, dt, dev = c_all.shape[-1], ctx.kx, x_all.dtype, x_all.device
        bf16 = dt == torch.bfloat16
        dh_seq = dh_seq.to(dt).contiguous()
        dcat = torch.empty((T, G, B, K), dtype=dt, device=dev)
        dc = [torch.empty((G, B, H), dtype=torch.float32, device=dev) for _ in range(2)]
        st, R, esz = _lib.current_stream(), G * B, x_all.element_size()
        # one cell per node: the point-wise kernel also leaves the bias gradient as per-block partial sums (no second pass
        # over the [T*B, 4H] dgates); several cells in one node keep the grouped column sum below
        rpb = _bsum_rows_per_block(R, H) if (G == 1 and any(ctx.needs_input_grad[4:])) else 0
        nblk = (R + rpb - 1) // rpb if rpb else 0
        bws = torch.empty((T, nblk, 4 * H), dtype=torch.float32, device=dev) if rpb else None
        for t in range(T - 1, -1, -1):
            last = t == T - 1
            # dgates overwrite the saved activations in place: act becomes dY for the weight gradient below
            args = (act[t].data_ptr(), int(bf16), c_all[t].data_ptr(), keep_rows[t].data_ptr(),
                    c_all[t + 1].data_ptr(), dh_seq[t].data_ptr(),
                    None if last else dcat[t + 1].data_ptr() + kx * esz, K,
                    None if last else keep_rows[t + 1].data_ptr(),
                    None if last else dc[(t + 1) & 1].data_ptr(), act[t].data_ptr(), dc[t & 1].data_ptr())
            if ctx.pre:
                _lib.check(lib.fdyn_lstm_seq_bwd_pre(args[0], args[1], b.data_ptr(), B, *args[2:], bws[t].data_ptr() if rpb else None,
                                                     rpb, R, H, st), "lstm_seq_bwd_pre")
            elif rpb:
                _lib.check(lib.fdyn_lstm_seq_bwd_bsum(*args, bws[t].data_ptr(), rpb, R, H, st), "lstm_seq_bwd_bsum")
            else:
                _lib.check(lib.fdyn_lstm_seq_bwd(*args, R, H, st), "lstm_seq_bwd")
            torch.bmm(act[t], w, out=dcat[t])
        needs = ctx.needs_input_grad
        grads = [None] * (4 * G)
        if any(needs[4:]):
            # dW_g = sum over (t, b) of dY^T X: one batched GEMM over the T*G row blocks (fp32), then a [1, T] GEMM over t
            part = _bmm_f32(act.view(T * G, B, 4 * H).transpose(1, 2), x_all.view(T * G, B, K))     # [T*G, 4H, K]
            ones = torch.ones((1, T), dtype=part.dtype, device=dev)
            dw = torch.mm(ones, part.view(T, G * 4 * H * K)).view(G, 4 * H, K)
            db = colsum_partials(bws.view(T * nblk, 4 * H)).view(1, 4 * H) if rpb else colsum(act.view(T * G * B, 4 * H), B, G).view(G, 4 * H)
            d = ctx.param_dtypes
            for g in range(G):
                if needs[4 + 4 * g]:
                    grads[4 * g] = dw[g, :, :kx].to(d[4 * g])
                if needs[5 + 4 * g]:
                    grads[4 * g + 1] = dw[g, :, kx:].to(d[4 * g + 1])
                if needs[6 + 4 * g]:
                    grads[4 * g + 2] = db[g].to(d[4 * g + 2])
                if needs[7 + 4 * g]:
                    grads[4 * g + 3] = db[g].to(d[4 * g + 3])
        dfeats = None
        if needs[0]:
            dfeats = dcat[:, 0, :, :kx] if G == 1 else dcat[..., :kx].float().sum(1).to(dt) if G > 2 else \
                (dcat[:, 0, :, :kx] + dcat[:, 1, :, :kx])
        return (dfeats, None, None, None, *grads)


def lstm_sequence(feats, cells, h0, c0, keep):
    """feats [T,B,kx] (compute dtype); cells = [(w_ih, w_hh, b_ih, b_hh), ...] parameters of G nn.LSTM layers that all read
    `feats`; h0 / c0 [G,B,H] (states before step 0, constants); keep [T,B] (0 where an episode starts at that step)
    -> h_seq [T,G,B,H] (compute dtype), c_all [T+1,G,B,H] fp32."""
    flat = [p for cell in cells for p in cell]
    return _LSTMSequenceFn.apply(feats.contiguous(), keep, h0, c0, *flat)


class _PPOLossFn(torch.autograd.Function):
    """Clipped-surrogate PPO loss of one slice, forward and gradient in one fused launch (fdyn_ppo_loss)."""

    @staticmethod
    def forward(ctx, mean, values, log_std, actions, old_logp, adv, ret, old_values, normalize_adv, clip_range, clip_range_vf,
                vf_coef, ent_coef):
        lib = _lib.load()
        f = lambda t: t.detach().float().contiguous()                     # noqa: E731
        mean_, values_, actions_, old_logp_, adv_, ret_ = f(mean), f(values), f(actions), f(old_logp), f(adv), f(ret)
        M = values_.numel()
        assert mean_.numel() == 4 * M and actions_.numel() == 4 * M and old_logp_.numel() == M and adv_.numel() == M and ret_.numel() == M
        ov = f(old_values) if clip_range_vf is not None else None
        dev = mean_.device
        dmean, dvalues = torch.empty_like(mean_), torch.empty_like(values_)
        stats = torch.empty(9, dtype=torch.float32, device=dev)
        ws = torch.empty(2, dtype=torch.float32, device=dev)
        ls = f(log_std)
        _lib.check(lib.fdyn_ppo_loss(mean_.data_ptr(), actions_.data_ptr(), ls.data_ptr(), values_.data_ptr(), old_logp_.data_ptr(),
                                     adv_.data_ptr(), ret_.data_ptr(), _lib.ptr(ov), int(bool(normalize_adv)), float(clip_range),
                                     float(clip_range_vf) if clip_range_vf is not None else -1.0, float(vf_coef), M,
                                     dmean.data_ptr(), dvalues.data_ptr(), stats.data_ptr(), ws.data_ptr(), _lib.current_stream()),
                   "ppo_loss")
        # entropy of the diagonal Gaussian: sum_k (0.5 + 0.5 log 2 pi + log_std_k); loss += ent_coef * (-entropy)
        entropy = 4 * 1.4189385332046727 + ((ls[0] + ls[1]) + (ls[2] + ls[3]))
        loss = stats[4] - ent_coef * entropy
        ctx.save_for_backward(dmean, dvalues, stats[5:9] - ent_coef)
        ctx.shapes = (mean.shape, values.shape, mean.dtype, values.dtype, log_std.dtype)
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, g, _gs):
        dmean, dvalues, dls = ctx.saved_tensors
        ms, vs, md, vd, ld = ctx.shapes
        return ((dmean * g).view(ms).to(md), (dvalues * g).view(vs).to(vd), (dls * g).to(ld)) + (None,) * 10


def ppo_loss(mean, values, log_std, actions, old_logp, adv, ret, old_values, normalize_adv, clip_range, clip_range_vf, vf_coef,
             ent_coef):
    """-> (loss, stats[4] = policy loss, value loss, approx KL, clip fraction).  Fused HIP kernel on the GPU; the CPU path
    (unit tests, gloo rehearsals) is the same arithmetic in torch ops."""
    if mean.is_cuda:
        loss, st = _PPOLossFn.apply(mean, values, log_std, actions, old_logp, adv, ret, old_values, normalize_adv, clip_range,
                                    clip_range_vf, vf_coef, ent_coef)
        return loss, st[:4]
    import math
    var = (2 * log_std).exp()
    logp = (-((actions - mean) ** 2) / (2 * var) - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
    if normalize_adv:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    ratio = torch.exp(logp - old_logp)
    pl = -torch.min(adv * ratio, adv * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)).mean()
    if clip_range_vf is not None:
        values = old_values + torch.clamp(values - old_values, -clip_range_vf, clip_range_vf)
    vl = torch.nn.functional.mse_loss(ret, values)
    entropy = (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum()
    loss = pl + ent_coef * (-entropy) + vf_coef * vl
    with torch.no_grad():
        kl = ((ratio - 1) - (logp - old_logp)).mean()
        cf = ((ratio - 1).abs() > clip_range).float().mean()
    return loss, torch.stack([pl.detach(), vl.detach(), kl, cf])


def episode_flags(terminated, truncated, episode_start, keep, counter=None):
    """episode_start <- (terminated | truncated) as fp32, keep <- 1 - episode_start, counter += 1: the glue between an env step
    and the next policy step in ONE launch (GPU); plain tensor ops elsewhere."""
    if terminated.is_cuda and terminated.dtype == torch.uint8 and truncated.dtype == torch.uint8:
        lib = _lib.load()
        _lib.check(lib.fdyn_episode_flags(_lib.ptr(terminated), _lib.ptr(truncated), _lib.ptr(episode_start), _lib.ptr(keep),
                                          _lib.ptr(counter), terminated.numel(), _lib.current_stream()), "episode_flags")
        return
    done = ((terminated != 0) | (truncated != 0)).float()
    episode_start.copy_(done)
    if keep is not None:
        keep.copy_(1.0 - done)
    if counter is not None:
        counter.add_(1)


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
