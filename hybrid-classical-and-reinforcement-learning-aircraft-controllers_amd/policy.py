"""PPO + LSTM rate-controller policy in PyTorch-ROCm.

Architecture = what the reference's `train_rate.py:115-147` builds: sb3_contrib `RecurrentPPO("MlpLstmPolicy")` with
`LSTMPolicy.get_policy_kwargs()` (learned_controllers/networks/lstm_policy.py:107-136):

  features extractor (lstm_policy.py:13-97): Linear(18,128)+ReLU -> nn.LSTM(128,256,num_layers=2) -> Linear(256,128)+ReLU
      NOTE the reference calls `self.lstm(embedded)` on a length-1 sequence WITHOUT carried state (:75-92), so this
      2-layer LSTM always starts from h=c=0: the W_hh products and the forget gate vanish and each layer reduces to
      h = sigmoid(o) * tanh(sigmoid(i) * tanh(g)) with gates = x W_ih^T + b_ih + b_hh.  We keep nn.LSTM's parameter
      tensors (checkpoint-compatible) but evaluate that closed form: one GEMM per layer instead of two.
  actor LSTM / critic LSTM (sb3_contrib defaults): nn.LSTM(128, 256, 1) each, state carried across env steps and
      zeroed at episode starts -- THE recurrence, and the only dense contraction with a sequential dependency;
  mlp_extractor: pi [128, 64], vf [128, 64], ReLU;  action_net Linear(64,4), value_net Linear(64,1), log_std (4).
SB3 / sb3_contrib are third-party and absent here: numerics parity with them is unpinned; shapes,
parameter names and hyper-parameters follow the reference's config (learned_controllers/config/ppo_lstm.yaml:38-58).

Every matmul is a `[B, K] x [K, 4H]` GEMM that rocBLAS/hipBLASLt runs on MFMA; with `compute_dtype=torch.bfloat16`
the GEMMs run in bf16 with fp32 accumulate (autocast), the cell state and the PPO loss stay fp32.
"""
import math
import os
from typing import NamedTuple, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused import DeferredWgrad, deferred_linear, linear, linear_relu, lstm_cell, lstm_sequence, zero_state_lstm_layer

OBS_DIM, ACT_DIM = 18, 4


class RNNStates(NamedTuple):
    pi_h: torch.Tensor   # [B, H]
    pi_c: torch.Tensor
    vf_h: torch.Tensor
    vf_c: torch.Tensor

    def detach(self):
        return RNNStates(*(t.detach() for t in self))

    def masked(self, keep):
        """keep = 1 - episode_start, [B] -> zero the state of envs that start a new episode."""
        k = keep.unsqueeze(-1)
        return RNNStates(*(t * k.to(t.dtype) for t in self))

    def index(self, idx):
        return RNNStates(*(t[idx] for t in self))


def _lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM step: ONE GEMM over the concatenated [x, h] (K = in + H, MFMA via hipBLASLt; PyTorch gate order
    i, f, g, o) followed by ONE fused point-wise launch (fused.lstm_cell)."""
    gates = F.linear(torch.cat([x, h.to(x.dtype)], dim=-1), torch.cat([w_ih, w_hh], dim=1), b_ih + b_hh)
    return lstm_cell(gates, c)


def _lstm_zero_state_layer(x, w_ih, b_ih, b_hh):
    return zero_state_lstm_layer(x, w_ih, b_ih, b_hh)


def _run_seq(seq, x):
    """nn.Sequential of Linear / ReLU, with the Linears routed through fused.linear (split-K weight gradients) and a
    Linear + ReLU pair through fused.linear_relu (bias + ReLU in the GEMM epilogue)."""
    mods, i = list(seq), 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and m.bias is not None:
            x = linear_relu(x, m.weight, m.bias)
            i += 2
            continue
        x = linear(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
        i += 1
    return x


# k order inside every block of 16 of a layer whose input arrives as the previous layer's MFMA output tile (csrc/policy_fe64.hip)
_KPERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)


FE_GATE_SCALE = (-1.4426950408889634, 1.0, -2.0 * 1.4426950408889634, -1.4426950408889634)     # gates i, f (unused), g, o


def pack_fe_weights(w_emb, w1, w2, w_proj):
    """The features extractor's four weight matrices as the byte image csrc/policy_fe64.hip streams through LDS.

    Chunks in the order the kernel uses them -- embedding [128 rows, K 18 -> 32]; per 32-unit slice of LSTM layer 1 and 2 the
    rows of gates (i, g) [64 rows] then of gate o [32 rows] (gate f multiplies the zero cell state and is not needed); per
    32-feature slice of the projection [32 rows] -- each row K bf16 followed by 16 bytes of padding, each chunk padded to a
    multiple of 1 KB.  Layers 1, 2 and the projection read their input straight out of the previous layer's MFMA output
    registers, which fixes the order of k inside every block of 16 (_KPERM16)."""
    bf = torch.bfloat16
    H = w1.shape[0] // 4

    def perm(K):
        return _kperm(K, w1.device)

    def chunk(rows):                      # rows [R, K] -> bytes of the padded LDS image, a whole number of 1 KB pieces
        R = rows.shape[0]
        img = torch.cat([rows.to(bf), torch.zeros(R, 8, dtype=bf, device=rows.device)], 1).reshape(-1)
        pad = (-img.numel()) % 512
        return torch.cat([img, torch.zeros(pad, dtype=bf, device=rows.device)])

    parts = [chunk(F.pad(w_emb.detach().float(), (0, 32 - w_emb.shape[1])))]
    # rows of the LSTM layers pre-scaled (in fp32, before the one rounding to bf16) by the factor their gate's exponent takes:
    # sigmoid(z) = 1 / (1 + 2^(-z log2 e)) for i and o, tanh(z) = (1 - 2^(-2 z log2 e)) / (1 + ...) for g -- the kernel's
    # accumulators (which start from the equally scaled biases) then ARE the exponents: FE_GATE_SCALE
    gate_scale = torch.tensor(FE_GATE_SCALE, dtype=torch.float32, device=w1.device).repeat_interleave(H)[:, None]
    for w in (w1, w2):
        wp = (w.detach().float() * gate_scale)[:, perm(w.shape[1])]
        for s in range(H // 32):
            parts.append(chunk(torch.cat([wp[32 * s:32 * s + 32], wp[2 * H + 32 * s:2 * H + 32 * s + 32]], 0)))
            parts.append(chunk(wp[3 * H + 32 * s:3 * H + 32 * s + 32]))
    wq = w_proj.detach().float()[:, perm(w_proj.shape[1])]
    for s in range(w_proj.shape[0] // 32):
        parts.append(chunk(wq[32 * s:32 * s + 32]))
    return torch.cat(parts).contiguous()


# ---- csrc/policy_rc64.hip: both recurrent cells in one launch, lane = batch row ---------------------------------------------
# The kernel keeps activations and state in its own layouts (whole 1 KB wave accesses, a lane touches only its own row).  Row
# 64 wb + 32 t + r lives in lane r + 32 hf of wave-block wb, tile t; hidden unit n = 32 sl + 16 hq + 8 p2 + 4 hf + p01.
_KPERM_CACHE = {}


def _kperm(K, device):
    """Column order of a K-wide weight matrix whose input arrives as MFMA output fragments; one device tensor per (K, device),
    built once (prepare_inference runs before every rollout: a torch.tensor(list, device=cuda) there is a synchronous H2D copy)."""
    key = (int(K), str(device))
    if key not in _KPERM_CACHE:
        _KPERM_CACHE[key] = torch.tensor([16 * (k // 16) + _KPERM16[k % 16] for k in range(K)], device=device)
    return _KPERM_CACHE[key]


def rc_pack_x(x):
    """[B][16 S] row-major activations -> B fragments [B/64][2][S][64][8] bf16 (S k-steps; features 128: S = 8, h 256: S = 16),
    returned with the row-major shape."""
    B, K = x.shape
    return (x.to(torch.bfloat16).reshape(B // 64, 2, 32, K // 32, 2, 2, 2, 4).permute(0, 1, 3, 4, 6, 2, 5, 7)
            .contiguous().view(B, K))


def rc_unpack_x(xi):
    B, K = xi.shape
    return xi.reshape(B // 64, 2, K // 32, 2, 2, 32, 2, 4).permute(0, 1, 5, 2, 3, 6, 4, 7).contiguous().view(B, K)


rc_pack_h, rc_unpack_h = rc_pack_x, rc_unpack_x


def rc_pack_c(c):
    """[B][256] fp32 cell state -> [B/64][slice 8][tile 2][group 4][lane 64][4] fp32."""
    B, Hh = c.shape
    return c.float().reshape(B // 64, 2, 32, Hh // 32, 4, 2, 4).permute(0, 3, 1, 4, 5, 2, 6).contiguous().view(B, Hh)


def rc_unpack_c(ci):
    B, Hh = ci.shape
    return ci.reshape(B // 64, Hh // 32, 2, 4, 2, 32, 4).permute(0, 2, 5, 1, 3, 4, 6).contiguous().view(B, Hh)


def pack_rc_weights(cells):
    """[(W_ih [4H][128], W_hh [4H][256])] for the actor and the critic -> the byte image csrc/policy_rc64.hip streams through
    LDS: per cell and 32-unit slice the rows of gates i, g, f, o (the order the kernel multiplies them in), each row
    [W_ih | W_hh] with the k order of the activation fragments (_KPERM16) + 16 bytes of padding, each 32-row chunk padded to
    28 KB (seven 1 KB pieces per wave)."""
    bf = torch.bfloat16
    parts = []
    for w_ih, w_hh in cells:
        dev = w_ih.device
        w = torch.cat([w_ih.detach().float()[:, _kperm(w_ih.shape[1], dev)], w_hh.detach().float()[:, _kperm(w_hh.shape[1], dev)]], 1)
        Hh = w.shape[0] // 4
        for sl in range(Hh // 32):
            for gate in (0, 2, 1, 3):
                rows = w[gate * Hh + 32 * sl:gate * Hh + 32 * sl + 32].to(bf)
                img = torch.cat([rows, torch.zeros(32, 8, dtype=bf, device=dev)], 1).reshape(-1)
                parts.append(torch.cat([img, torch.zeros(28 * 512 - img.numel(), dtype=bf, device=dev)]))
    return torch.cat(parts).contiguous()


class LSTMFeaturesExtractor(nn.Module):
    def __init__(self, obs_dim: int = OBS_DIM, features_dim: int = 128, lstm_hidden_size: int = 256, n_lstm_layers: int = 2):
        super().__init__()
        self.features_dim = features_dim
        self.embedding = nn.Sequential(nn.Linear(obs_dim, 128), nn.ReLU())
        self.lstm = nn.LSTM(input_size=128, hidden_size=lstm_hidden_size, num_layers=n_lstm_layers, batch_first=True)
        self.output_proj = nn.Sequential(nn.Linear(lstm_hidden_size, features_dim), nn.ReLU())

    def forward(self, obs):
        x = _run_seq(self.embedding, obs)
        for layer in range(self.lstm.num_layers):
            x = _lstm_zero_state_layer(x, getattr(self.lstm, f"weight_ih_l{layer}"), getattr(self.lstm, f"bias_ih_l{layer}"),
                                       getattr(self.lstm, f"bias_hh_l{layer}")).to(x.dtype)
        return _run_seq(self.output_proj, x)


def _mlp(sizes):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers += [nn.Linear(a, b), nn.ReLU()]
    return nn.Sequential(*layers)


class RateLSTMPolicy(nn.Module):
    """Actor-critic with separate actor/critic LSTMs (sb3_contrib RecurrentActorCriticPolicy defaults)."""

    def __init__(self, features_dim: int = 128, lstm_hidden_size: int = 256, n_lstm_layers: int = 2,
                 policy_lstm_hidden: int = 256, net_arch_pi=(128, 64), net_arch_vf=(128, 64), use_lstm: bool = True,
                 mlp_net_arch=(256, 128, 64), compute_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.use_lstm, self.hidden, self.compute_dtype = use_lstm, policy_lstm_hidden, compute_dtype
        self.deferred_wgrad = True       # BPTT: one split-K weight-gradient GEMM per recurrent cell per backward pass
        self.sequence_bptt = True        # BPTT: each recurrent cell over T steps is one autograd node (fused.lstm_sequence)
        self.group_cells_max_batch = 8192    # slices up to this many envs run actor + critic cells as one batched node
        if use_lstm:
            self.features_extractor = LSTMFeaturesExtractor(OBS_DIM, features_dim, lstm_hidden_size, n_lstm_layers)
            self.lstm_actor = nn.LSTM(features_dim, policy_lstm_hidden, 1)
            self.lstm_critic = nn.LSTM(features_dim, policy_lstm_hidden, 1)
            self.pi_net = _mlp([policy_lstm_hidden, *net_arch_pi])
            self.vf_net = _mlp([policy_lstm_hidden, *net_arch_vf])
            last_pi, last_vf = net_arch_pi[-1], net_arch_vf[-1]
        else:   # SimpleMLPPolicy (lstm_policy.py:139-164): PPO("MlpPolicy"), pi = vf = mlp.net_arch
            self.features_extractor = nn.Identity()
            self.pi_net = _mlp([OBS_DIM, *mlp_net_arch])
            self.vf_net = _mlp([OBS_DIM, *mlp_net_arch])
            last_pi = last_vf = mlp_net_arch[-1]
        self.action_net = nn.Linear(last_pi, ACT_DIM)
        self.value_net = nn.Linear(last_vf, 1)
        self.log_std = nn.Parameter(torch.zeros(ACT_DIM))
        self._init_weights()

    def _init_weights(self):
        # SB3 ortho_init: sqrt(2) for the trunks, 0.01 for the action head, 1 for the value head
        for mod, gain in ((self.features_extractor, math.sqrt(2)), (self.pi_net, math.sqrt(2)), (self.vf_net, math.sqrt(2)),
                          (self.action_net, 0.01), (self.value_net, 1.0)):
            for m in mod.modules():
                if isinstance(m, nn.Linear):
                    nn.init.orthogonal_(m.weight, gain=gain)
                    nn.init.zeros_(m.bias)

    def initial_state(self, batch: int, device=None) -> RNNStates:
        z = lambda: torch.zeros(batch, self.hidden, device=device or self.log_std.device)  # noqa: E731
        return RNNStates(z(), z(), z(), z())

    # ---- single env step (rollout) --------------------------------------------------------------------------
    def _core(self, obs, states: RNNStates):
        feats = self.features_extractor(obs)
        if not self.use_lstm:
            return self.pi_net(feats), self.vf_net(feats), states
        la, lc = self.lstm_actor, self.lstm_critic
        pi_h, pi_c = _lstm_cell(feats, states.pi_h, states.pi_c, la.weight_ih_l0, la.weight_hh_l0, la.bias_ih_l0, la.bias_hh_l0)
        vf_h, vf_c = _lstm_cell(feats, states.vf_h, states.vf_c, lc.weight_ih_l0, lc.weight_hh_l0, lc.bias_ih_l0, lc.bias_hh_l0)
        return self.pi_net(pi_h), self.vf_net(vf_h), RNNStates(pi_h, pi_c, vf_h, vf_c)

    # ---- fused inference path: every LSTM cell is ONE hand-written MFMA kernel (csrc/lstm_mfma.hip) ---------------
    def prepare_inference(self):
        """Snapshot the weights as bf16 [4H, in+H] / fp32 bias tensors for the fused rollout path.  Call after every
        optimizer phase (RecurrentPPO.collect_rollout does); the training path never reads this cache."""
        if not self.use_lstm:
            self._inf = None
            return
        fe, bf = self.features_extractor, torch.bfloat16

        def cat(l, k):
            return torch.cat([getattr(l, f"weight_ih_l{k}"), getattr(l, f"weight_hh_l{k}")], 1).detach().to(bf).contiguous()

        def bias(l, k):
            return (getattr(l, f"bias_ih_l{k}") + getattr(l, f"bias_hh_l{k}")).detach().float().contiguous()
        def lin(seq):
            """[(w_bf16, b_bf16), ...] of the Linear layers of a Sequential (ReLU follows each one)."""
            return [(m.weight.detach().to(bf).contiguous(), m.bias.detach().to(bf).contiguous())
                    for m in seq if isinstance(m, nn.Linear)]
        new = {
            "fe_w": [getattr(fe.lstm, f"weight_ih_l{k}").detach().to(bf).contiguous() for k in range(fe.lstm.num_layers)],
            "fe_b": [bias(fe.lstm, k) for k in range(fe.lstm.num_layers)],
            "pi_w": cat(self.lstm_actor, 0), "pi_b": bias(self.lstm_actor, 0),
            "vf_w": cat(self.lstm_critic, 0), "vf_b": bias(self.lstm_critic, 0),
            # the small Linear layers, pre-cast once per rollout: under autocast every call re-casts weight and bias
            # (17 tiny cast launches = 75 us of a 65 536-env policy step, rocprofv3)
            "emb": lin(fe.embedding), "proj": lin(fe.output_proj), "pi": lin(self.pi_net), "vf": lin(self.vf_net),
            "act": lin([self.action_net])[0], "val": lin([self.value_net])[0],
        }
        # the whole features extractor as one kernel (csrc/policy_fe64.hip) when it has the reference's shape
        emb_l = [m for m in fe.embedding if isinstance(m, nn.Linear)]
        proj_l = [m for m in fe.output_proj if isinstance(m, nn.Linear)]
        if (fe.lstm.num_layers == 2 and len(emb_l) == 1 and len(proj_l) == 1 and emb_l[0].weight.shape == (128, OBS_DIM)
                and fe.lstm.weight_ih_l0.shape == (1024, 128) and fe.lstm.weight_ih_l1.shape == (1024, 256)
                and proj_l[0].weight.shape == (128, 256)):
            new["fe_img"] = pack_fe_weights(emb_l[0].weight, fe.lstm.weight_ih_l0, fe.lstm.weight_ih_l1, proj_l[0].weight)
            new["fe_bias"] = torch.cat([emb_l[0].bias.detach().float(), bias(fe.lstm, 0), bias(fe.lstm, 1),
                                        proj_l[0].bias.detach().float()]).contiguous()
        # the two trunks as one kernel (csrc/policy_trunk.hip) when they have the reference's shape [256 -> 128 -> 64]
        pi_l, vf_l = [m for m in self.pi_net if isinstance(m, nn.Linear)], [m for m in self.vf_net if isinstance(m, nn.Linear)]
        if (len(pi_l) == 2 and len(vf_l) == 2 and all(l[0].weight.shape == (128, 256) and l[1].weight.shape == (64, 128) for l in (pi_l, vf_l))):
            perm = _kperm(128, pi_l[0].weight.device)
            new["trunk_w1"] = torch.stack([pi_l[0].weight.detach(), vf_l[0].weight.detach()]).to(bf).contiguous()
            new["trunk_b1"] = torch.stack([pi_l[0].bias.detach(), vf_l[0].bias.detach()]).float().contiguous()
            new["trunk_w2p"] = torch.stack([pi_l[1].weight.detach()[:, perm], vf_l[1].weight.detach()[:, perm]]).to(bf).contiguous()
            new["trunk_b2"] = torch.stack([pi_l[1].bias.detach(), vf_l[1].bias.detach()]).float().contiguous()
        old = getattr(self, "_inf", None)
        if old is None:
            self._inf = new
        else:       # refresh IN PLACE: a captured rollout graph holds these pointers

            def refresh(dst, src):
                if isinstance(dst, torch.Tensor):
                    dst.copy_(src)
                else:
                    for d_, s_ in zip(dst, src):
                        refresh(d_, s_)
            for k in new:
                refresh(old[k], new[k])

    @staticmethod
    def _mlp_bf16(x, layers):
        for w, b in layers:
            x = torch._addmm_activation(b, x, w.t())        # relu(x W^T + b): bias + ReLU in the GEMM epilogue (hipBLASLt)
        return x

    def _fused_ok(self, obs):
        inf = None if os.environ.get("FDYN_NO_MFMA") else getattr(self, "_inf", None)
        return (inf is not None and obs.is_cuda and self.compute_dtype == torch.bfloat16 and not torch.is_grad_enabled()
                and self.hidden == 256 and all(w.shape[1] in (128, 256) for w in inf["fe_w"]))

    def _core_fused(self, obs, states: RNNStates, keep, out_states: Optional[RNNStates] = None, heads: Optional[dict] = None,
                    flags: Optional[tuple] = None):
        """Explicit bf16 inference path (no autocast): pre-cast Linear weights + one MFMA kernel per LSTM cell.
        heads = {"deterministic": bool}: where the trunk kernel serves the shape, the output heads and the sampling run behind
        the trunks in the same launch and the return value is (actions, value, logp, new_states).
        flags = (terminated, truncated, episode_start, keep, counter): the glue between the previous env step and this step, done
        by the features kernel for its rows (else by fused.episode_flags in a launch of its own, before anything reads keep)."""
        from . import _lib
        lib, inf, B, H = _lib.load(), self._inf, obs.shape[0], self.hidden
        bf, dev = torch.bfloat16, obs.device
        st = _lib.current_stream()
        def shapes_ok(x_, w_, b_, kh_):      # the kernel trusts its sizes: check them on the host before every launch
            return (x_.is_contiguous() and x_.dtype == bf and x_.shape == (B, w_.shape[1] - kh_) and w_.is_contiguous()
                    and w_.dtype == bf and b_.dtype == torch.float32 and b_.numel() == w_.shape[0] and w_.shape[0] % 128 == 0)
        if "fe_img" in inf and B % 256 == 0 and B >= 128 * 256 and not os.environ.get("FDYN_NO_FE64"):
            # one kernel from the observation to the features: activations stay in registers between the four layers
            o32 = obs.float().contiguous()
            assert o32.shape == (B, OBS_DIM) and inf["fe_img"].numel() * 2 == lib.fdyn_policy_features_image_bytes() \
                and inf["fe_bias"].numel() == 128 + 1024 + 1024 + 128, "policy_features operand shapes"
            feats = torch.empty((B, 128), dtype=bf, device=dev)
            if flags is not None and flags[0].dtype == torch.uint8 and flags[1].dtype == torch.uint8:
                term, trunc, es, kp, ctr = flags
                assert term.shape == (B,) and trunc.shape == (B,) and es.shape == (B,) and kp.shape == (B,) and es.dtype == torch.float32 \
                    and kp.dtype == torch.float32 and all(t.is_contiguous() for t in (term, trunc, es, kp)), "policy_features_flags operands"
                _lib.check(lib.fdyn_policy_features_flags(o32.data_ptr(), inf["fe_img"].data_ptr(), inf["fe_bias"].data_ptr(), feats.data_ptr(),
                                                          term.data_ptr(), trunc.data_ptr(), es.data_ptr(), kp.data_ptr(), _lib.ptr(ctr), B, st),
                           "policy_features_flags")
                flags = None
            else:
                if flags is not None:
                    from .fused import episode_flags
                    episode_flags(*flags)
                    flags = None
                _lib.check(lib.fdyn_policy_features(o32.data_ptr(), inf["fe_img"].data_ptr(), inf["fe_bias"].data_ptr(),
                                                    feats.data_ptr(), B, st), "policy_features")
        else:
            if flags is not None:
                from .fused import episode_flags
                episode_flags(*flags)
                flags = None
            x = self._mlp_bf16(obs.to(bf), inf["emb"])
            for w, b in zip(inf["fe_w"], inf["fe_b"]):                   # zero-state layers: no h/c input at all
                assert shapes_ok(x, w, b, 0), "lstm_cell_mfma operand shapes"
                h = torch.empty((B, w.shape[0] // 4), dtype=bf, device=dev)
                _lib.check(lib.fdyn_lstm_cell_mfma(x.data_ptr(), x.shape[1], None, 0, None, None, w.data_ptr(), b.data_ptr(),
                                                   h.data_ptr(), None, None, B, w.shape[0] // 4, st), "lstm_cell_mfma")
                x = h
            feats = self._mlp_bf16(x, inf["proj"])
        out = []
        outs = (None, None, None, None) if out_states is None else tuple(out_states)
        cells = []
        for (w, b, hp, cp), (ho, co) in zip(((inf["pi_w"], inf["pi_b"], states.pi_h, states.pi_c),
                                             (inf["vf_w"], inf["vf_b"], states.vf_h, states.vf_c)), (outs[0:2], outs[2:4])):
            hp, cp = hp.to(bf).contiguous(), cp.float().contiguous()
            assert shapes_ok(feats, w, b, H) and hp.shape == (B, H) and cp.shape == (B, H) and keep.shape == (B,) \
                and w.shape[0] == 4 * H, "lstm_cell_mfma operand shapes"
            # the kernel writes the new state straight into the caller's ping-pong buffers when given (no copies)
            ok = ho is not None and ho.dtype == bf and ho.shape == (B, H) and ho.is_contiguous() and co.dtype == torch.float32 \
                and co.shape == (B, H) and co.is_contiguous() \
                and ((ho.data_ptr() != hp.data_ptr() and co.data_ptr() != cp.data_ptr()) or lib.fdyn_lstm_cell_mfma_inplace_ok(128, H, H, B))
            h = ho if ok else torch.empty((B, H), dtype=bf, device=dev)
            c = co if ok else torch.empty((B, H), dtype=torch.float32, device=dev)
            cells.append((hp, cp, w, b, h, c))
            out += [h, c]
        # actor and critic cell in one launch (fdyn_lstm_cell_mfma_pair falls back to two where the paired kernel does not apply)
        if os.environ.get("FDYN_NO_CELL_PAIR"):             # A/B knob: the two launches of rounds 1-2
            for hp, cp, w, b, h, c in cells:
                _lib.check(lib.fdyn_lstm_cell_mfma(feats.data_ptr(), feats.shape[1], hp.data_ptr(), H, cp.data_ptr(), keep.data_ptr(),
                                                   w.data_ptr(), b.data_ptr(), h.data_ptr(), c.data_ptr(), None, B, H, st), "lstm_cell_mfma")
        else:
            _lib.check(lib.fdyn_lstm_cell_mfma_pair(feats.data_ptr(), feats.shape[1], keep.data_ptr(), H, B, H,
                                                    *[t.data_ptr() for cell in cells for t in cell], st), "lstm_cell_mfma_pair")
        # the two trunks; the 64 -> 4 and 64 -> 1 output layers are fused with the sampling (fdyn_policy_heads)
        if "trunk_w1" in inf and H == 256 and not os.environ.get("FDYN_NO_TRUNK"):
            h_pi, h_vf = out[0], out[2]
            assert h_pi.dtype == bf and h_vf.dtype == bf and h_pi.is_contiguous() and h_vf.is_contiguous() \
                and h_pi.shape == (B, 256) and h_vf.shape == (B, 256) and inf["trunk_w1"].shape == (2, 128, 256) \
                and inf["trunk_w2p"].shape == (2, 64, 128), "policy_trunks operand shapes"
            if heads is not None and inf["act"][0].shape == (ACT_DIM, 64) and inf["val"][0].shape == (1, 64) \
                    and not os.environ.get("FDYN_NO_TRUNK_HEADS"):
                actions = torch.empty((B, ACT_DIM), dtype=torch.float32, device=dev)
                logp = torch.empty(B, dtype=torch.float32, device=dev)
                value = torch.empty(B, dtype=torch.float32, device=dev)
                _lib.check(lib.fdyn_policy_trunks_heads(
                    h_pi.data_ptr(), h_vf.data_ptr(), inf["trunk_w1"].data_ptr(), inf["trunk_b1"].data_ptr(), inf["trunk_w2p"].data_ptr(),
                    inf["trunk_b2"].data_ptr(), inf["act"][0].data_ptr(), inf["act"][1].data_ptr(), inf["val"][0].data_ptr(),
                    inf["val"][1].data_ptr(), self.log_std.detach().float().contiguous().data_ptr(), self._noise_seed,
                    self._noise_step.data_ptr(), int(heads["deterministic"]), actions.data_ptr(), logp.data_ptr(), value.data_ptr(), B, st),
                    "policy_trunks_heads")
                return actions, value, logp, RNNStates(*out)
            lat_pi = torch.empty((B, 64), dtype=bf, device=dev)
            lat_vf = torch.empty((B, 64), dtype=bf, device=dev)
            _lib.check(lib.fdyn_policy_trunks(h_pi.data_ptr(), h_vf.data_ptr(), inf["trunk_w1"].data_ptr(), inf["trunk_b1"].data_ptr(),
                                              inf["trunk_w2p"].data_ptr(), inf["trunk_b2"].data_ptr(), lat_pi.data_ptr(),
                                              lat_vf.data_ptr(), B, st), "policy_trunks")
            return lat_pi, lat_vf, RNNStates(*out)
        return self._mlp_bf16(out[0], inf["pi"]), self._mlp_bf16(out[2], inf["vf"]), RNNStates(*out)

    def recurrent_inplace_ok(self, batch: int, device) -> bool:
        """True if the fused rollout path may update the recurrent state of `batch` envs IN PLACE (one state set instead of a
        ping-pong pair: the footprint that decides whether the state survives in the Infinity Cache between steps)."""
        from . import _lib
        if not (self.use_lstm and self.compute_dtype == torch.bfloat16 and torch.device(device).type == "cuda" and self.hidden == 256
                and os.environ.get("FDYN_INPLACE_STATE", "1") != "0" and not os.environ.get("FDYN_NO_MFMA")):
            return False
        return bool(_lib.load().fdyn_lstm_cell_mfma_inplace_ok(128, self.hidden, self.hidden, int(batch)))

    def noise_counter(self, device):
        """The device-side step counter of the fused path's action noise (created on first use)."""
        if not hasattr(self, "_noise_seed"):
            self._noise_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            self._noise_step = torch.zeros(1, dtype=torch.int32, device=device)
        return self._noise_step

    def step(self, obs, states: RNNStates, episode_start, deterministic: bool = False,
             out_states: Optional[RNNStates] = None, keep: Optional[torch.Tensor] = None, bump_noise: bool = True,
             done_flags: Optional[tuple] = None):
        """obs [B,18], episode_start [B] (1 where the env was just reset) -> actions, values, log_probs, new states.
        keep: 1 - episode_start if the caller already has it (fused.episode_flags writes both); bump_noise=False: the caller
        moves the noise counter on itself (the same launch) -- a rollout loop then has no framework glue launches left.
        done_flags = (terminated, truncated) of the PREVIOUS env step (fused path only, with `keep` given): episode_start and keep
        are REWRITTEN from them and the noise counter moves on inside the step's first kernel."""
        if self._fused_ok(obs):
            if keep is None:
                assert done_flags is None, "done_flags needs the caller's keep buffer"
                keep = (1.0 - episode_start.float()).contiguous()        # the mask is applied inside the kernel
            # output heads + sampling + log-prob ride behind the trunks (or in ONE launch of their own): in-kernel Philox keyed
            # by a per-policy seed, the env index and a step counter that lives on the device, so a captured graph draws fresh
            # noise on every replay
            self.noise_counter(obs.device)
            fl = None
            if done_flags is not None:
                fl = (done_flags[0], done_flags[1], episode_start, keep, self._noise_step)
            elif bump_noise:
                self._noise_step.add_(1)
            res = self._core_fused(obs, states, keep, out_states, heads={"deterministic": deterministic}, flags=fl)
            if len(res) == 4:
                return res
            lat_pi, lat_vf, new_states = res
            from . import _lib
            inf, B, dev = self._inf, obs.shape[0], obs.device
            assert lat_pi.shape == (B, 64) and lat_vf.shape == (B, 64) and lat_pi.is_contiguous() and lat_vf.is_contiguous() \
                and inf["act"][0].shape == (ACT_DIM, 64) and inf["val"][0].shape == (1, 64), "policy_heads operand shapes"
            actions = torch.empty((B, ACT_DIM), dtype=torch.float32, device=dev)
            logp = torch.empty(B, dtype=torch.float32, device=dev)
            value = torch.empty(B, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().fdyn_policy_heads(lat_pi.data_ptr(), lat_vf.data_ptr(), inf["act"][0].data_ptr(),
                                                     inf["act"][1].data_ptr(), inf["val"][0].data_ptr(), inf["val"][1].data_ptr(),
                                                     self.log_std.detach().float().contiguous().data_ptr(), self._noise_seed,
                                                     self._noise_step.data_ptr(), int(deterministic), actions.data_ptr(),
                                                     logp.data_ptr(), value.data_ptr(), B, _lib.current_stream()), "policy_heads")
            return actions, value, logp, new_states
        states = states.masked(1.0 - episode_start.float())
        with torch.autocast(obs.device.type, dtype=self.compute_dtype, enabled=self.compute_dtype is not None):
            lat_pi, lat_vf, new_states = self._core(obs, states)
            mean = self.action_net(lat_pi).float()
            value = self.value_net(lat_vf).float().squeeze(-1)
        std = self.log_std.exp()
        actions = mean if deterministic else mean + std * torch.randn_like(mean)
        return actions, value, self._log_prob(actions, mean), new_states

    def _log_prob(self, actions, mean):
        var = (2 * self.log_std).exp()
        return (-((actions - mean) ** 2) / (2 * var) - self.log_std - 0.5 * math.log(2 * math.pi)).sum(-1)

    def entropy(self):
        return (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std).sum()

    # ---- sequence evaluation (PPO update, BPTT over T) ----------------------------------------------------------
    def evaluate_sequence(self, obs, actions, episode_starts, states: RNNStates) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """obs [T,B,18], actions [T,B,4], episode_starts [T,B], states at t=0 -> values [T,B], log_probs [T,B], entropy."""
        values, mean = self.sequence_heads(obs, episode_starts, states)
        return values, self._log_prob(actions, mean), self.entropy()

    def sequence_heads(self, obs, episode_starts, states: RNNStates) -> Tuple[torch.Tensor, torch.Tensor]:
        """BPTT forward over T steps: obs [T,B,18], episode_starts [T,B], states at t=0 -> values [T,B], action means [T,B,4]."""
        T = obs.shape[0]
        with torch.autocast(obs.device.type, dtype=self.compute_dtype, enabled=self.compute_dtype is not None):
            if not self.use_lstm:
                flat = obs.reshape(-1, OBS_DIM)
                lat_pi, lat_vf = _run_seq(self.pi_net, flat), _run_seq(self.vf_net, flat)
                mean = linear(lat_pi, self.action_net.weight, self.action_net.bias).float().view(T, -1, ACT_DIM)
                values = linear(lat_vf, self.value_net.weight, self.value_net.bias).float().view(T, -1)
                return values, mean
            # the zero-state feature extractor has no time dependence: run it for all T*B rows in one set of GEMMs
            feats = self.features_extractor(obs.reshape(-1, OBS_DIM)).view(T, -1, self.features_extractor.features_dim)
            la, lc = self.lstm_actor, self.lstm_critic
            wa, ba = torch.cat([la.weight_ih_l0, la.weight_hh_l0], 1), la.bias_ih_l0 + la.bias_hh_l0
            wc, bc = torch.cat([lc.weight_ih_l0, lc.weight_hh_l0], 1), lc.bias_ih_l0 + lc.bias_hh_l0
            pi_h, pi_c, vf_h, vf_c = states
            if self.sequence_bptt and feats.is_cuda:
                # the actor and critic cells over all T steps are ONE autograd node (fused.lstm_sequence): per step and
                # direction one batched GEMM + one point-wise launch for both cells, weight gradients from one batched GEMM
                keep_all = 1.0 - episode_starts.float()
                cells = [(l.weight_ih_l0, l.weight_hh_l0, l.bias_ih_l0, l.bias_hh_l0) for l in (la, lc)]
                dt = feats.dtype
                if feats.shape[1] <= self.group_cells_max_batch:
                    # small slices: both cells in one node (halves the launches; measured 244 -> 214 ms at 2048-env slices)
                    h_seq, _ = lstm_sequence(feats, cells, torch.stack([pi_h.to(dt), vf_h.to(dt)]), torch.stack([pi_c, vf_c]), keep_all)
                    pi_seq, vf_seq = h_seq.unbind(1)       # backward = one stack copy (two selects: two zero fills + an add)
                else:
                    # large slices fill the chip per cell; the batched GEMM is then slower than two plain ones (74 vs 68 ms)
                    pi_seq = lstm_sequence(feats, cells[:1], pi_h.to(dt).unsqueeze(0), pi_c.unsqueeze(0), keep_all)[0].squeeze(1)
                    vf_seq = lstm_sequence(feats, cells[1:], vf_h.to(dt).unsqueeze(0), vf_c.unsqueeze(0), keep_all)[0].squeeze(1)
                mean = linear(_run_seq(self.pi_net, pi_seq), self.action_net.weight, self.action_net.bias).float()
                values = linear(_run_seq(self.vf_net, vf_seq), self.value_net.weight, self.value_net.bias).float().squeeze(-1)
                return values, mean
            pi_hs, vf_hs = [], []
            # per-step autograd with deferred weight gradients (fused.DeferredWgrad): each step's backward only produces dX;
            # dW / db of the two recurrent cells come from one split-K GEMM over all T*B rows when the backward pass ends
            defer = self.deferred_wgrad and feats.is_cuda and torch.is_grad_enabled()
            if defer:
                Bn, kx, dt = feats.shape[1], feats.shape[2], feats.dtype
                wa, ba, wc, bc = wa.detach().to(dt), ba.detach().to(dt), wc.detach().to(dt), bc.detach().to(dt)
                mk = lambda l, w: DeferredWgrad(T, Bn, w.shape[1], w.shape[0], dt, feats.device,   # noqa: E731
                                                [(l.weight_ih_l0, slice(0, kx)), (l.weight_hh_l0, slice(kx, w.shape[1]))],
                                                [l.bias_ih_l0, l.bias_hh_l0])
                bk_a, bk_c = mk(la, wa), mk(lc, wc)
            for t in range(T):
                keep = (1.0 - episode_starts[t].float()).unsqueeze(-1)
                kh = keep.to(pi_h.dtype)
                pi_h, pi_c, vf_h, vf_c = pi_h * kh, pi_c * keep, vf_h * kh, vf_c * keep
                x = feats[t]
                if defer:
                    pi_h, pi_c = lstm_cell(deferred_linear(x, pi_h.to(x.dtype), wa, ba, bk_a, t), pi_c)
                    vf_h, vf_c = lstm_cell(deferred_linear(x, vf_h.to(x.dtype), wc, bc, bk_c, t), vf_c)
                else:
                    pi_h, pi_c = lstm_cell(F.linear(torch.cat([x, pi_h.to(x.dtype)], -1), wa, ba), pi_c)
                    vf_h, vf_c = lstm_cell(F.linear(torch.cat([x, vf_h.to(x.dtype)], -1), wc, bc), vf_c)
                pi_hs.append(pi_h); vf_hs.append(vf_h)
            pi_seq, vf_seq = torch.stack(pi_hs), torch.stack(vf_hs)
            mean = linear(_run_seq(self.pi_net, pi_seq), self.action_net.weight, self.action_net.bias).float()
            values = linear(_run_seq(self.vf_net, vf_seq), self.value_net.weight, self.value_net.bias).float().squeeze(-1)
        return values, mean

    def predict_values(self, obs, states: RNNStates, episode_start):
        return self.step(obs, states, episode_start, deterministic=True)[1]

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())

    @staticmethod
    def flops_per_env_step(use_lstm: bool = True) -> float:
        """Forward multiply-add flops per env step (2*K*N per row): embedding + 2 zero-state layers + proj + 2 LSTMs + heads."""
        if not use_lstm:
            return 2.0 * 2 * (18 * 256 + 256 * 128 + 128 * 64) + 2 * (64 * 4 + 64)
        return 2.0 * (18 * 128 + 128 * 1024 + 256 * 1024 + 256 * 128 + 2 * (384 * 1024) + 2 * (256 * 128 + 128 * 64) + 64 * 4 + 64)
