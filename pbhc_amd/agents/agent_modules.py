"""Networks of the general-tracking (KungfuBot2) agent: Actor (motion ConvEncoder + history ConvEncoder | priv MLP -> MLP) and
ActorCritic, with the reference's `state_dict()` key names so checkpoints load both ways with `strict=True`
(reference: humanoidverse/agents/modules/agent_modules.py:11-166, encoder_modules.py:22-107, modules.py:5-66).

The two Conv1d layers of an encoder see 20 (or 10) time steps and leave 3: each output position is a strided GEMM on a window VIEW
of the `[B, T, C]` activations the per-step Linear already produces (`_WindowConv1d`: no unfolded copy, no MIOpen convolution);
the parameters keep nn.Conv1d's shapes and names.
"""
from __future__ import annotations

import os as _os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Normal

from .modules import BaseModule

class _TallSkinnyLinear(torch.autograd.Function):
    """y = x W^T + b for x [K, M] with K (rows x time steps, ~5e5) >> M, N (<= 360 x 60).  The weight gradient dy^T x is a reduction over K
    with a tiny output: as one GEMM rocBLAS runs it on a handful of workgroups (measured 690 us per call at K = 491520 on MI355X); split
    over P row chunks it is a batched GEMM that fills the chip, followed by a [P, N, M] sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w if ctx.needs_input_grad[0] else None
        K = x.shape[0]
        P = 1
        while P < 512 and K % (2 * P) == 0 and K // (2 * P) >= 256:
            P *= 2
        dw = torch.bmm(dy.view(P, K // P, -1).transpose(1, 2), x.view(P, K // P, -1)).sum(0)
        return dx, dw, dy.sum(0)


def _linear(x, lin):
    if x.requires_grad or lin.weight.requires_grad:
        return _TallSkinnyLinear.apply(x.contiguous(), lin.weight, lin.bias)
    return F.linear(x, lin.weight, lin.bias)


class _TallSkinnyLinearReLU(torch.autograd.Function):
    """relu(x W^T + b) for the encoders' per-time-step Linear (x: [rows x time steps, d], ~5e5 rows): the forward is ONE fp32-MFMA GEMM with bias
    + ReLU in its epilogue (`pbhc_linear_act_fwd`; the library ran GEMM, then a clamp pass over the [5e5, 60] output), the backward masks the
    incoming gradient and sums the bias gradient in one pass (`pbhc_act_bwd_bias`; autograd ran threshold_backward + a column sum), the weight
    gradient keeps the split-K batched form of _TallSkinnyLinear."""

    @staticmethod
    def forward(ctx, x, w, b):
        from .. import _lib

        K, n = x.shape[0], w.shape[0]
        y = torch.empty(K, n, device=x.device)
        _lib.check(_lib.lib().pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, K, n, x.shape[1], 3, _lib.current_stream()),
                   "pbhc_linear_act_fwd")
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib

        x, w, y = ctx.saved_tensors
        K, n = y.shape
        dy = dy.contiguous()
        dz = torch.empty_like(dy)                                          # out of place: the incoming gradient stays the caller's
        gb = torch.empty(n, device=y.device)
        scratch = torch.empty(_lib.K["PBHC_ACT_MAX_BLOCKS"] * n, device=y.device)
        _lib.check(_lib.lib().pbhc_act_bwd_bias(dy.data_ptr(), y.data_ptr(), K, n, 3, dz.data_ptr(), gb.data_ptr(), scratch.data_ptr(), _lib.current_stream()),
                   "pbhc_act_bwd_bias")
        dx = dz @ w if ctx.needs_input_grad[0] else None
        P = 1
        while P < 512 and K % (2 * P) == 0 and K // (2 * P) >= 256:
            P *= 2
        dw = torch.bmm(dz.view(P, K // P, -1).transpose(1, 2), x.view(P, K // P, -1)).sum(0)
        return dx, dw, gb


def _linear_relu(x, lin):
    """F.relu(lin(x)) for a tall input."""
    from . import fused_mlp

    if fused_mlp.FUSED_GEMM and x.is_cuda and x.dtype == torch.float32 and lin.bias is not None and lin.weight.is_contiguous():
        if x.requires_grad or lin.weight.requires_grad:
            return _TallSkinnyLinearReLU.apply(x.contiguous(), lin.weight, lin.bias)
    return F.relu(_linear(x, lin))


def _conv_window_grads(x, dz, wp, k, s, need_dx):
    """dx [B, T, C] (or None) and dw [O, k*C] of the L window GEMMs of a Conv1d backward: dz [B, L, O] (contiguous), x [B, T, C] (contiguous),
    wp [O, k*C].  One window's product (24 576 rows x 40 x 360: 0.7 GFLOP) fills a fraction of the chip — 18-20 us each at 35 TFLOP/s, sixteen
    per optimiser step for the motion encoder's first layer — but queueing the independent ones on side streams (the L weight-gradient
    products; the input-gradient products of windows ceil(k / s) apart, which touch disjoint time steps) measured 85.6 against 75.3 ms per
    update on MI355X: concurrent small GEMMs slow each other by more than the overlap saves, as every two-stream form of the update did."""
    B, T, Cin = x.shape
    L, O = dz.shape[1], dz.shape[2]
    kC = k * Cin
    dx = None
    if need_dx:
        dx = torch.zeros_like(x)
        for l in range(L):                                               # overlapping windows: sequential accumulation
            dx[:, l * s:l * s + k, :].reshape(B, kC).addmm_(dz[:, l, :], wp)
    P = 1
    while P < 64 and B % (2 * P) == 0 and B // (2 * P) >= 256:
        P *= 2
    part = None
    for l in range(L):
        a = x[:, l * s:l * s + k, :].reshape(B, kC).view(P, B // P, kC)
        g = dz[:, l, :].view(P, B // P, O).transpose(1, 2)
        part = torch.bmm(g, a) if part is None else torch.baddbmm(part, g, a, out=part)
    return dx, part.sum(0)


# 1: a Conv1d layer's input gradient as one strided-batched GEMM per stride phase (_conv_dgrad_phases) instead of the per-window accumulation
# loop.  Measured in the general-tracking update (tools/probes/conv_bwd_probe.py, bench.py): 2 launches of 75 us (K = 120: five rounds of short
# tiles) + a 15 us padded copy against 8 x 21 us + a 16 us fill — 70.4 against 68.9 ms per update — so the loop stays the default.
CONV_DGRAD_PHASES = _os.environ.get("PBHC_CONV_DGRAD_PHASES", "0") != "0"
_DZ_PAD_CACHE = {}


def _padded_dz(B, L, O, J, T, s, device):
    """[B, Lp, O] with the L real windows behind J - 1 zero windows (and enough zero windows after them): allocated once per shape — the
    interior is rewritten by every backward, the pad windows stay zero; consumers are stream-ordered inside that backward"""
    pad = J - 1
    Lp = max(L + 2 * pad, (T - 1) // s + J)
    key = (B, L, O, J, Lp, str(device))
    buf = _DZ_PAD_CACHE.get(key)
    if buf is None:
        if len(_DZ_PAD_CACHE) > 16:
            _DZ_PAD_CACHE.clear()
        buf = _DZ_PAD_CACHE[key] = torch.zeros(B, Lp, O, device=device)
    return buf, pad, Lp


def _conv_dgrad_phases(dzp, pad, w, k, s, T):
    """Input gradient of nn.Conv1d(C -> O, k, s | k) on time-major activations as ONE strided-batched GEMM per stride phase: time step
    t = m s + p receives dz[:, l, :] . W[:, :, t - l s] from the J = k / s windows l = m - J + 1 .. m, which are J * O CONTIGUOUS floats of
    the zero-padded gradient `dzp` [B, Lp, O] (window l at index l + pad) — so dx[:, m s + p, :] = dzp[:, m : m + J, :] . B_p with
    B_p [J O, C] the phase's taps stacked, batched over m (A advances by O floats, the output by s C): every element of dx is written once
    (the accumulation loop: L launches reading and re-writing overlapping slices of a zero-filled dx, 168 us for the motion encoder's first
    conv layer at 24 576 rows against 2 launches)"""
    from .. import _lib

    B, Lp, O = dzp.shape
    Cin = w.shape[1]
    J = k // s
    lib, st = _lib.lib(), _lib.current_stream()
    dx = torch.empty(B, T, Cin, device=dzp.device)
    wt = w.permute(1, 2, 0)                                             # [C, k, O]
    for p in range(s):
        taps = [(J - 1 - jj) * s + p for jj in range(J)]                  # window l = m - (J - 1) + jj reads tap t - l s
        bpt = wt[:, taps, :].reshape(Cin, J * O)                          # [N = C, K = (jj, o)]: the "weight" of an NT GEMM
        nm = (T - 1 - p) // s + 1
        _lib.check(lib.pbhc_linear_act_fwd_strided(dzp.data_ptr(), Lp * O, O, bpt.data_ptr(), None, dx.data_ptr() + 4 * p * Cin, None, T * Cin, s * Cin, nm,
                                                   B, Cin, J * O, 0, st), "pbhc_linear_act_fwd_strided")
    return dx


class _WindowConv1d(torch.autograd.Function):
    """nn.Conv1d(C -> O, kernel k, stride s, no padding) on x [B, T, C] (time-major rows, the layout the per-step Linear produces) -> [B, L, O].

    Output position l reads the k consecutive time steps l*s .. l*s+k-1, which are ONE contiguous run of k*C floats per row of x: the
    window matrix A_l = x[:, l*s : l*s+k, :] is a [B, k*C] view with row stride T*C, so each output position is a plain strided GEMM
    with the weight reordered to (k, C) — no unfolded copy of the activations (283 MB per conv layer and minibatch at 24 576 rows), no
    `unfold_backward` scatter: the input gradient accumulates through L overlapping strided GEMM outputs (beta = 1), the weight gradient
    through L split-K batched GEMMs (see _TallSkinnyLinear)."""

    @staticmethod
    def forward(ctx, x, w, b, k, s):
        B, T, Cin = x.shape
        O = w.shape[0]
        L = (T - k) // s + 1
        wp = w.permute(0, 2, 1).reshape(O, k * Cin)                    # [O, (k, C)]
        # every window's GEMM writes a CONTIGUOUS [B, O] slab (an `out=` row-slice of [B, L, O] makes addmm broadcast-copy the bias into it
        # first: one extra launch per window, 36 per optimiser step in the teacher); the bias is added once, in the pass that lays the
        # slabs out time-major
        buf = torch.empty(L, B, O, device=x.device, dtype=x.dtype)
        wt = wp.t()
        for l in range(L):
            torch.mm(x[:, l * s:l * s + k, :].reshape(B, k * Cin), wt, out=buf[l])
        out = torch.add(buf.permute(1, 0, 2), b)                        # [B, L, O] contiguous
        ctx.save_for_backward(x, w)
        ctx.k, ctx.s = k, s
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        k, s = ctx.k, ctx.s
        B, T, Cin = x.shape
        O = w.shape[0]
        L = dy.shape[1]
        dy = dy.contiguous()
        wp = w.permute(0, 2, 1).reshape(O, k * Cin)
        dx, dwp = _conv_window_grads(x, dy, wp, k, s, ctx.needs_input_grad[0])
        dw = dwp.view(O, k, Cin).permute(0, 2, 1)
        return dx, dw, dy.sum((0, 1)), None, None


class _WindowConv1dAct(torch.autograd.Function):
    """act(nn.Conv1d(C -> O, k, s)(x)) on time-major x [B, T, C] -> [B, L, O] with the L window GEMMs of _WindowConv1d in ONE launch of the fused
    fp32-MFMA kernel (`pbhc_linear_act_fwd_strided`: window l is the strided matrix x + l*s*C with row stride T*C, written straight into
    out[:, l, :]; bias + ReLU / SiLU in the epilogue) — instead of L library GEMMs, a bias / layout pass and an activation pass.  Backward: the
    activation derivative and the bias gradient in one pass (`pbhc_act_bwd_bias` over the [B*L, O] rows), then _WindowConv1d's window loops."""

    @staticmethod
    def forward(ctx, x, w, b, k, s, act):
        from .. import _lib

        B, T, Cin = x.shape
        O = w.shape[0]
        L = (T - k) // s + 1
        wp = w.permute(0, 2, 1).reshape(O, k * Cin).contiguous()        # [O, (k, C)]
        out = torch.empty(B, L, O, device=x.device)
        pre = torch.empty_like(out) if act == 2 else None               # SiLU' needs the pre-activation
        _lib.check(_lib.lib().pbhc_linear_act_fwd_strided(x.data_ptr(), T * Cin, s * Cin, wp.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                          pre.data_ptr() if pre is not None else None, L * O, O, L, B, O, k * Cin, act,
                                                          _lib.current_stream()), "pbhc_linear_act_fwd_strided")
        ctx.save_for_backward(x, w, pre if pre is not None else out)
        ctx.k, ctx.s, ctx.act = k, s, act
        return out

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib

        x, w, saved = ctx.saved_tensors
        k, s = ctx.k, ctx.s
        B, T, Cin = x.shape
        O = w.shape[0]
        L = dy.shape[1]
        dy = dy.contiguous()
        wp = w.permute(0, 2, 1).reshape(O, k * Cin)
        dz = torch.empty_like(dy)
        gb = torch.empty(O, device=dy.device)
        scratch = torch.empty(_lib.K["PBHC_ACT_MAX_BLOCKS"] * O, device=dy.device)
        _lib.check(_lib.lib().pbhc_act_bwd_bias(dy.data_ptr(), saved.data_ptr(), B * L, O, ctx.act, dz.data_ptr(), gb.data_ptr(), scratch.data_ptr(),
                                                _lib.current_stream()), "pbhc_act_bwd_bias")
        # (opt-in, see CONV_DGRAD_PHASES: the input gradient as one GEMM launch per stride phase on a zero-padded copy of dz; the weight gradient
        # keeps reading the CONTIGUOUS dz — its batched-GEMM shapes are the ones the shipped GEMM selections were made for)
        phases = CONV_DGRAD_PHASES and ctx.needs_input_grad[0] and k % s == 0 and (k // s) * O >= 4 and L >= 6 and T * Cin * B < (1 << 30)
        if phases:
            dzp, pad, Lp = _padded_dz(B, L, O, k // s, T, s, dy.device)
            dzp[:, pad:pad + L, :].copy_(dz)
            dx = _conv_dgrad_phases(dzp, pad, w, k, s, T)
        _, dwp = _conv_window_grads(x, dz, wp, k, s, False) if phases else (None, None)
        if not phases:
            dx, dwp = _conv_window_grads(x, dz, wp, k, s, ctx.needs_input_grad[0])
        dw = dwp.view(O, k, Cin).permute(0, 2, 1)
        return dx, dw, gb, None, None, None


ONE_LAUNCH_ENCODER = _os.environ.get("PBHC_ENCODER_ONE_LAUNCH", "1") != "0"      # (0: the four per-layer launches; measurement aid)
_CONV_TABLE = {5: ([20, 10], [2, 2], [1, 1]), 10: ([20, 10], [4, 2], [2, 1]), 20: ([40, 20], [6, 4], [2, 2])}     # encoder_modules.py:60-77


class ConvEncoder(nn.Module):
    def __init__(self, obs_dim_dict, module_config_dict, time_steps):
        super().__init__()
        self.obs_dim_dict = obs_dim_dict
        self.module_config_dict = module_config_dict
        self.time_steps = time_steps
        input_dim = 0
        for each in module_config_dict["input_dim"]:
            if each in obs_dim_dict:
                input_dim += obs_dim_dict[each]
            elif isinstance(each, (int, float)):
                input_dim += each
            else:
                raise ValueError(f"_calculate_dim - Unknown input type: {each}")
        self.input_dim = input_dim
        self.output_dim = module_config_dict["output_dim"]
        self.hidden_dim = module_config_dict["hidden_dim"]
        lc = module_config_dict["layer_config"]
        if lc["type"] != "Conv1d":
            raise NotImplementedError(f"Unsupported layer type: {lc['type']}")
        if time_steps not in _CONV_TABLE:
            raise ValueError(f"Unsupported time_steps for now: {time_steps}")
        out_channels, kernel_sizes, strides = _CONV_TABLE[time_steps]
        self.encoder = nn.Sequential(nn.Linear(self.input_dim, self.hidden_dim), nn.ReLU())
        act = getattr(nn, lc["activation"])()
        layers, in_ch = [], self.hidden_dim
        for oc, k, s in zip(out_channels, kernel_sizes, strides):
            layers += [nn.Conv1d(in_ch, oc, k, s), act]
            in_ch = oc
        self.conv_module = nn.Sequential(*layers)
        self.output_layer = nn.Linear(out_channels[-1] * 3, self.output_dim)
        self._act = act
        self._strides = strides
        self.unfold_gemm = None        # None: unfolded-window GEMMs on the GPU, nn.Conv1d on the CPU (what the deploy exporters trace)

    # ---- no-grad forward on the GPU (the rollout: 4 096 rows per control step, inside a captured graph) ------------------------------------
    def _infer_ok(self, x):
        from . import fused_mlp

        act_id = fused_mlp._ACT_ID.get(type(self._act), 0)
        return (fused_mlp.FUSED_GEMM and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and self.input_dim >= 4
                and act_id in (1, 2, 3) and not (act_id == 1 and self._act.alpha != 1.0)
                and all(self.conv_module[2 * i].bias is not None for i in range(len(self._strides))) and self.encoder[0].bias is not None)

    def _layout_weights(self):
        """the conv weights as [O, (k, C)] (a window of the time-major activations is k*C contiguous floats) and the output layer's columns in
        (l, c) order (it then reads the last conv layer's [B, L, O] rows as they lie, no transpose pass)"""
        ws = [self.conv_module[2 * i].weight.permute(0, 2, 1).reshape(self.conv_module[2 * i].out_channels, -1) for i in range(len(self._strides))]
        O = self.conv_module[2 * (len(self._strides) - 1)].out_channels
        wo = self.output_layer.weight
        ws.append(wo.view(wo.shape[0], O, wo.shape[1] // O).permute(0, 2, 1).reshape(wo.shape[0], -1))
        return ws

    def prepare_inference(self):
        """refresh the re-laid-out weight copies IN PLACE (fixed addresses: the rollout's captured graph reads them) — the weights are constant
        over a rollout, so the agent calls this once before its loop and `release_inference()` after it; without it every forward re-lays them out"""
        with torch.no_grad():
            ws = self._layout_weights()
            cache = self.__dict__.get("_infer_w")
            if cache is None or any(c.shape != w.shape or c.device != w.device for c, w in zip(cache, ws)):
                self._infer_w = [w.contiguous().clone() for w in ws]
            else:
                for c, w in zip(cache, ws):
                    c.copy_(w)
            self._pack_encoder()
        self._infer_ready = True

    def _pack_encoder(self):
        """the four weight matrices in the operand layout of the one-launch encoder (`pbhc_conv_encoder_fwd`), refreshed in place"""
        import ctypes as C

        from .. import _lib
        from . import fused_mlp

        self._enc_c = None
        lin, out = self.encoder[0], self.output_layer
        convs = [self.conv_module[2 * i] for i in range(len(self._strides))]
        if not (len(convs) == 2 and self.input_dim <= 128 and lin.weight.is_cuda and lin.weight.is_contiguous() and fused_mlp.FUSED_GEMM and fused_mlp.FUSED_STACK):
            return
        lib, st = _lib.lib(), _lib.current_stream()
        mats = [lin.weight] + list(self._infer_w)
        sizes = [int(lib.pbhc_mlp_packed_floats(m.shape[0], m.shape[1])) for m in mats]
        bufs = self.__dict__.get("_enc_bufs")
        if bufs is None or [b.numel() for b in bufs] != sizes or bufs[0].device != lin.weight.device:
            bufs = self._enc_bufs = [torch.empty(n, device=lin.weight.device) for n in sizes]
        for m, b in zip(mats, bufs):
            _lib.check(lib.pbhc_mlp_pack(m.data_ptr(), m.shape[0], m.shape[1], b.data_ptr(), st), "pbhc_mlp_pack")
        e = _lib.PbhcConvEncoder()
        e.w1, e.wc1, e.wc2, e.wo = (b.data_ptr() for b in bufs)
        e.b1, e.bc1, e.bc2, e.bo = lin.bias.data_ptr(), convs[0].bias.data_ptr(), convs[1].bias.data_ptr(), out.bias.data_ptr()
        e.T, e.d, e.H, e.E = self.time_steps, self.input_dim, self.hidden_dim, out.out_features
        e.O1, e.k1, e.s1 = convs[0].out_channels, convs[0].kernel_size[0], self._strides[0]
        e.O2, e.k2, e.s2 = convs[1].out_channels, convs[1].kernel_size[0], self._strides[1]
        e.act = fused_mlp._ACT_ID[type(self._act)]
        if (e.s1 * e.H) % 4 or (e.s2 * e.O1) % 4 or int(lib.pbhc_conv_encoder_lds_bytes(C.byref(e))) > 160 * 1024:
            return
        self._enc_c = e

    def release_inference(self):
        self._infer_ready = False

    def _forward_infer(self, x, B, out=None):
        """per-step Linear + ReLU as ONE strided-batched launch over the T time steps straight from the (row-padded) observation slab, each
        Conv1d layer as one strided-batched launch with bias + activation in its epilogue, the output Linear on the [B, L*O] rows as they lie:
        4 launches (the generic no-grad path: slab copy, GEMM, unfold copy, GEMM, activation, ... 12)"""
        from .. import _lib
        from . import fused_mlp

        lib, st = _lib.lib(), _lib.current_stream()
        T, d, H = self.time_steps, self.input_dim, self.hidden_dim
        lin = self.encoder[0]
        ready = self.__dict__.get("_infer_ready", False)
        e = self.__dict__.get("_enc_c") if ready else None
        if (e is not None and ONE_LAUNCH_ENCODER and B <= 16384 and x.stride(0) >= (T - 1) * d + (d + 15) // 16 * 16
                and (out is None or (out.dim() == 2 and out.stride(1) == 1))):
            # the whole encoder as ONE launch (pbhc_conv_encoder_fwd: 16 rows per workgroup through the four layers, activations in LDS)
            import ctypes as C

            y = out if out is not None else torch.empty(B, e.E, device=x.device)
            _lib.check(lib.pbhc_conv_encoder_fwd(x.data_ptr(), x.stride(0), C.byref(e), y.data_ptr(), y.stride(0), B, st), "pbhc_conv_encoder_fwd")
            return y
        ws = self._infer_w if ready else [w.contiguous() for w in self._layout_weights()]
        h = torch.empty(B, T, H, device=x.device)
        _lib.check(lib.pbhc_linear_act_fwd_strided(x.data_ptr(), x.stride(0), d, lin.weight.data_ptr(), lin.bias.data_ptr(), h.data_ptr(), None, T * H, H, T,
                                                   B, H, d, 3, st), "pbhc_linear_act_fwd_strided")
        act_id = fused_mlp._ACT_ID[type(self._act)]
        for i, s in enumerate(self._strides):
            conv = self.conv_module[2 * i]
            k, Tc, Cin, O = conv.kernel_size[0], h.shape[1], h.shape[2], conv.out_channels
            L = (Tc - k) // s + 1
            nxt = torch.empty(B, L, O, device=x.device)
            _lib.check(lib.pbhc_linear_act_fwd_strided(h.data_ptr(), Tc * Cin, s * Cin, ws[i].data_ptr(), conv.bias.data_ptr(), nxt.data_ptr(), None, L * O, O, L,
                                                       B, O, k * Cin, act_id, st), "pbhc_linear_act_fwd_strided")
            h = nxt
        if out is not None:
            return torch.addmm(self.output_layer.bias, h.view(B, -1), ws[-1].t(), out=out)
        return F.linear(h.view(B, -1), ws[-1], self.output_layer.bias)

    def forward(self, x, out=None):
        """out (no-grad callers only): a [B, output_dim] tensor the result is written to (the rollout keeps every step's motion embedding)"""
        if out is not None:
            assert not torch.is_grad_enabled()
            if not (x.is_cuda and self._infer_ok(x)):
                return out.copy_(self.forward(x))
            return self._forward_infer(x, x.shape[0], out)
        B = x.shape[0] if x.dim() == 2 else x.numel() // (self.input_dim * self.time_steps)
        if not (x.is_cuda if self.unfold_gemm is None else self.unfold_gemm):
            h = self.encoder(x.reshape(-1, self.input_dim)).view(B, self.time_steps, self.hidden_dim).permute(0, 2, 1)
            return self.output_layer(self.conv_module(h).flatten(start_dim=1))
        if not torch.is_grad_enabled() and self._infer_ok(x):
            return self._forward_infer(x, B)
        x = _linear_relu(x.reshape(-1, self.input_dim), self.encoder[0]).view(B, self.time_steps, self.hidden_dim)   # [B, T, H]: x.view(-1, input_dim) chunks, sic
        for i, s in enumerate(self._strides):
            conv = self.conv_module[2 * i]
            k = conv.kernel_size[0]
            if torch.is_grad_enabled():
                from . import fused_mlp

                act_id = fused_mlp._ACT_ID.get(type(self._act), 0)
                if fused_mlp.FUSED_GEMM and x.is_cuda and act_id in (1, 2, 3) and x.dtype == torch.float32 and conv.bias is not None and k * x.shape[2] >= 4 \
                        and not (act_id == 1 and self._act.alpha != 1.0):
                    x = _WindowConv1dAct.apply(x.contiguous(), conv.weight, conv.bias, k, s, act_id)   # [B, L, O], activation included
                    continue
                x = self._act(_WindowConv1d.apply(x.contiguous(), conv.weight, conv.bias, k, s))       # [B, L, O]
            else:
                # rollout (4096 rows, inside a captured graph): one GEMM over the unfolded copy beats L small ones
                w = x.unfold(1, k, s)                                      # [B, L, C, k]
                x = self._act(F.linear(w.reshape(B * w.shape[1], -1), conv.weight.view(conv.out_channels, -1), conv.bias)).view(B, w.shape[1], conv.out_channels)
        return self.output_layer(x.transpose(1, 2).reshape(B, -1))       # flatten(start_dim=1) of [B, C, 3]


class Actor(nn.Module):
    def __init__(self, obs_dim_dict, module_config_dict, num_actions):
        super().__init__()
        for idx, od in enumerate(module_config_dict["output_dim"]):
            if od == "robot_action_dim":
                module_config_dict["output_dim"][idx] = num_actions
        if module_config_dict.get("type", "MLP") != "MLP":
            raise NotImplementedError
        self.actor_module = BaseModule(obs_dim_dict, module_config_dict)
        new_obs_dim_dict = dict(obs_dim_dict)
        me = module_config_dict["motion_encoder"]
        self.motion_encoder = ConvEncoder(obs_dim_dict, me, me["tsteps"])
        he = module_config_dict.get("history_encoder", None)
        if he is not None:
            new_obs_dim_dict["prop_history"] //= he["tsteps"]
            self.history_encoder = ConvEncoder(new_obs_dim_dict, he, he["tsteps"])
        else:
            self.history_encoder = None
        pe = module_config_dict.get("priv_encoder", None)
        self.priv_encoder = BaseModule(obs_dim_dict, pe) if pe is not None else None

    def motion_encoding(self, motion_obs):
        return self.motion_encoder(motion_obs)

    def history_encoding(self, history_obs):
        return self.history_encoder(history_obs)

    def priv_encoding(self, priv_obs):
        return self.priv_encoder(priv_obs)

    def forward(self, obs_dict, hist_encoding: bool, obs_key="actor_obs", target_key="future_motion_targets"):
        motion_embedding = self.motion_encoding(obs_dict[target_key])
        latent = self.history_encoding(obs_dict["prop_history"]) if hist_encoding else self.priv_encoding(obs_dict["priv_obs"])
        return self.actor_module(torch.cat([obs_dict[obs_key], motion_embedding, latent], dim=-1))


class ActorCritic(nn.Module):
    def __init__(self, obs_dim_dict, module_config_dict, num_actions, init_noise_std):
        super().__init__()
        self.actor_module = Actor(obs_dim_dict, module_config_dict["actor"], num_actions)
        cc = module_config_dict["critic"]
        if cc.get("type", "MLP") != "MLP":
            raise NotImplementedError
        self.critic_module = BaseModule(obs_dim_dict, cc)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        a = module_config_dict["actor"]
        self.fix_sigma = a.get("fix_sigma", False)
        self.max_sigma = a.get("max_sigma", 1.0)
        self.min_sigma = a.get("min_sigma", 0.1)
        if self.fix_sigma:
            self.std.requires_grad = False
        self.distribution = None
        Normal.set_default_validate_args = False

    @property
    def actor(self):
        return self.actor_module

    @property
    def critic(self):
        return self.critic_module

    def reset(self, dones=None):                 # agent_modules.py:130-134 of the reference: stateless, nothing to reset
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def sigma(self):
        return self.std.clamp(min=self.min_sigma, max=self.max_sigma)

    def update_distribution(self, obs, hist_encoding, obs_key):
        mean = self.actor(obs, hist_encoding, obs_key)
        self.distribution = Normal(mean, (mean * 0.0 + self.std).clamp(min=self.min_sigma, max=self.max_sigma))

    def act(self, obs, hist_encoding=False, obs_key="actor_obs", **kwargs):
        self.update_distribution(obs, hist_encoding, obs_key)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, obs, hist_encoding=True, **kwargs):
        return self.actor(obs, hist_encoding)

    def evaluate(self, obs, obs_key="actor_obs", **kwargs):
        motion_embedding = self.actor.motion_encoding(obs["future_motion_targets"])
        return self.critic(torch.cat([obs[obs_key], obs["priv_obs"], motion_embedding], dim=-1))
