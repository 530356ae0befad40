"""Training-time forward/backward of a Linear/activation stack (`BaseModule`, reference agents/modules/modules.py:47-63) arranged for
MI355X:

* hidden layers run on the fp32 matrix cores in `pbhc_linear_act_fwd` — bias + ELU / ReLU applied to the accumulators, no separate
  activation pass — and their input gradients in `pbhc_linear_dgrad_act`, which folds in the activation derivative of the layer below and
  the row-block column sums of its bias gradient (csrc/pbhc_gemm.hip: LDS-DMA staged, `v_mfma_f32_32x32x2_f32`).  `PBHC_FUSED_GEMM=0`
  returns to library GEMMs (hipBLASLt / rocBLAS through torch) + the `pbhc_act_bwd_bias` pass;
* weight gradients and the narrow output layer stay library GEMMs (split-K batched form below); a split-rows MFMA weight-gradient kernel
  (`pbhc_linear_wgrad`) exists and is pinned by tests, but measured 104-107 TFLOP/s against the library's 115-123 and is opt-in
  (`PBHC_FUSED_WGRAD=1`);
* weight / bias gradients are written by the GEMM (`out=`) and the fused kernel straight into the parameter's `.grad` — a view of
  the agent's flat gradient buffer — so autograd's per-parameter `grad += tmp` launches and temporaries disappear.  That store
  OVERWRITES, so it is opt-in: only a stack whose owner declared `grad_direct(seq)` — "I zero the gradient buffer before every backward"
  (the agents do) — takes it, and only while the stack has ONE live application; a stack applied twice in outstanding graphs, and
  every stack of user code, hands its gradients back to autograd, which accumulates as usual;
* ELU runs in place and its derivative is taken from the output (`y > 0 ? 1 : y + 1`), so pre-activations are not kept.
Same arithmetic as autograd's (the column sums are two-stage fp32 in a fixed order); pinned by the update-parity tests.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

_ACT_ID = {nn.ELU: 1, nn.SiLU: 2, nn.ReLU: 3}
FUSED_GEMM = os.environ.get("PBHC_FUSED_GEMM", "1") != "0"
_WGRAD_P = {tuple(int(v) for v in kv.split(":")[0].split("x")): int(kv.split(":")[1]) for kv in os.environ.get("PBHC_WGRAD_P", "").split(",") if kv}
FUSED_WGRAD = os.environ.get("PBHC_FUSED_WGRAD", "0") == "1"        # measured slower than the library's split-K form: opt-in


class _Live:
    """one per forward application of a stack, owned by its autograd context: alive while that graph can still run backward"""
    __slots__ = ("__weakref__",)


def grad_direct(seq, on=True):
    """Owner's declaration for `seq` (an nn.Sequential of Linear / activation): every parameter's `.grad` is a preallocated contiguous
    tensor that the owner zeroes before each backward, so the backward may store into it instead of returning gradients to autograd."""
    seq._grad_direct = bool(on)
    if not hasattr(seq, "_fused_live"):
        seq._fused_live = weakref.WeakSet()
        seq._fused_shared = False
        seq._zero_epoch, seq._stored_epoch = 0, -1


def grads_zeroed(seq):
    """The owner has just zeroed the gradient buffer behind `seq`'s `.grad`s: the NEXT backward through the stack may store (overwrite);
    any further backward before the next zeroing accumulates through autograd — two sequential backward() calls between two zeroings
    (an auxiliary loss, gradient accumulation) therefore add up, as they do for any other module."""
    seq._zero_epoch = getattr(seq, "_zero_epoch", 0) + 1


def supported(module_seq):
    layers = list(module_seq)
    lin = [m for m in layers if isinstance(m, nn.Linear)]
    acts = [m for m in layers if not isinstance(m, nn.Linear)]
    if not lin or len(layers) != 2 * len(lin) - 1 or not isinstance(layers[-1], nn.Linear):
        return False
    if any(type(a) not in _ACT_ID for a in acts) or len({type(a) for a in acts}) > 1:
        return False
    return not any(isinstance(a, nn.ELU) and a.alpha != 1.0 for a in acts)


def _wgrad(d, x, out, defer=None):
    """out[n, k] = d^T[n, B] x[B, k]: the split-rows MFMA kernel (`pbhc_linear_wgrad`) where its shape rules allow, else the library form.
    `defer`: a list; a split-K result's [P, n, k] partials are appended to it as (partials, out) instead of being summed here — the caller
    sums every deferred pair in its ONE finishing launch (pbhc_colsum_final)."""
    n, k = out.shape
    B = d.shape[0]
    if FUSED_GEMM and FUSED_WGRAD and d.is_cuda and d.dtype == torch.float32 and out.is_contiguous() and d.is_contiguous() and x.is_contiguous():
        lib = _lib.lib()
        P = lib.pbhc_linear_wgrad_parts(B, n, k)
        if P > 0:
            scratch = torch.empty(P * n * k, device=d.device)
            _lib.check(lib.pbhc_linear_wgrad(d.data_ptr(), x.data_ptr(), out.data_ptr(), scratch.data_ptr(), B, n, k, _lib.current_stream()), "pbhc_linear_wgrad")
            return out
    return _wgrad_library(d, x, out, defer)


def _wgrad_library(d, x, out, defer=None):
    """out[n, k] = d^T[n, B] x[B, k].  With B (minibatch rows, 24 576) >> n, k the single GEMM has few output tiles and a very long K loop:
    rocBLAS / hipBLASLt run it on part of the chip (128 x 256: 53 us = 30 TFLOP/s on MI355X).  Split over P row chunks it is a batched GEMM that
    fills the 256 CUs, followed by a [P, n, k] sum (23 us for the same shape); measured with tools/wgrad_splitk_probe.py."""
    n, k = out.shape
    B = d.shape[0]
    nk = n * k
    # (chosen in the update itself — bench.py with PBHC_WGRAD_P — not in a back-to-back probe, whose operands are warm in L2: 512 x 380 8 -> 4 parts
    # -0.35 ms per update, 768 x 630 one GEMM -> 2 parts -0.30, 256 x 512 16 -> 32 and 128 x 512 8 -> 16 another -0.05.  Round 3, with the
    # partial images summed by the backward's one finishing launch instead of a torch.sum each: 512 x 768 one GEMM -> 4 parts and 768 x 630
    # 2 -> 8 parts, 29.3 -> 28.85 ms per update on one box (16 parts: slower again) — tools/probes/wgrad_parts_sweep.sh)
    P = 0 if n < 64 else 32 if nk <= 128 * 256 else 16 if nk <= 128 * 512 else 32 if nk <= 256 * 512 else 4 if nk <= 512 * 768 else 8
    if _WGRAD_P:                                       # measurement aid: PBHC_WGRAD_P="512x380:4,256x512:8"
        P = _WGRAD_P.get((n, k), P)
    if P == 0 or B % P or B // P < 256:
        return torch.mm(d.t(), x, out=out)
    part = torch.bmm(d.view(P, B // P, n).transpose(1, 2), x.view(P, B // P, k))
    if defer is not None and out.is_contiguous():
        defer.append((part, out))                          # summed by the caller's finishing launch (one launch for the whole backward
        return out                                         # instead of one torch.sum per layer: 5 x 6 us per optimiser step)
    return torch.sum(part, 0, out=out)


# ---- the backward's finishing launch, shared ----------------------------------------------------------------------------------------------
# Every fused backward ends in ONE `pbhc_colsum_final` (bias gradients, the output layer's weight gradient, split-K partials): a latency-bound
# launch of ~16 us.  An owner that runs several stacks' backwards back to back on one stream (MHPPO: critic, then actor) brackets them with
# begin_deferred_finish() / finish_deferred(): stacks whose gradients are STORED (grad_direct) then hand their jobs — and the scratch tensors
# the jobs read — to a pending list, and one launch per PBHC_MAX_COLSUM_JOBS jobs finishes them all.  Stacks that return gradients to autograd
# finish at once as before (autograd may consume their outputs right after backward returns).
_DEFER = {"on": False, "jobs": [], "keep": []}


def begin_deferred_finish():
    _DEFER["on"] = True


def _launch_jobs(job_list, st):
    MAXJ = _lib.K["PBHC_MAX_COLSUM_JOBS"]
    lib = _lib.lib()
    for k in range(0, len(job_list), MAXJ):
        chunk = job_list[k:k + MAXJ]
        jobs = (_lib._S["PbhcColsumJob"] * MAXJ)()
        for j, (part, out, nrb, n) in zip(jobs, chunk):
            j.part, j.out, j.num_row_blocks, j.n = part, out, nrb, n
        _lib.check(lib.pbhc_colsum_final(jobs, len(chunk), st), "pbhc_colsum_final")


def finish_deferred():
    """launch what the backwards since begin_deferred_finish() left pending (current stream = the stream those backwards ran on)"""
    _DEFER["on"] = False
    job_list, keep = _DEFER["jobs"], _DEFER["keep"]
    _DEFER["jobs"], _DEFER["keep"] = [], []
    if job_list:
        _launch_jobs(job_list, _lib.current_stream())
    del keep


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seq, *params):
        ctx.grad_cols = None
        return _FusedMLP._fwd(ctx, x, seq)

    @staticmethod
    def _fwd(ctx, x, seq):
        lin = [m for m in seq if isinstance(m, nn.Linear)]
        act = _ACT_ID[type(seq[1])] if len(lin) > 1 else 0
        saved_in, saved_act = [], []
        h = x if x.is_contiguous() else x.contiguous()
        fused = FUSED_GEMM and act in (1, 2, 3) and h.is_cuda and h.dtype == torch.float32
        if fused:
            lib, st = _lib.lib(), _lib.current_stream()
        out = None
        for i, l in enumerate(lin):
            saved_in.append(h)
            if out is not None:                        # the output layer already ran in the epilogue of the layer below
                continue
            if fused and i < len(lin) - 1 and l.weight.is_contiguous():
                B = h.shape[0]
                z = torch.empty(B, l.out_features, device=h.device)
                pre = torch.empty_like(z) if act == 2 else None          # SiLU' needs the pre-activation; ELU' / ReLU' come from the output
                lo = lin[-1]
                if (FWD_OUT and i == len(lin) - 2 and l.out_features == 128 and lo.out_features <= 32 and l.in_features >= 4 and lo.weight.is_contiguous()
                        and 8192 <= B <= 32 * _lib.K["PBHC_ACT_MAX_BLOCKS"]):
                    # 32-row tiles hold whole 128-wide rows: the narrow output layer is applied to them while they are in LDS
                    out = torch.empty(B, lo.out_features, device=h.device)
                    _lib.check(lib.pbhc_linear_act_fwd_out(h.data_ptr(), l.weight.data_ptr(), l.bias.data_ptr() if l.bias is not None else None, z.data_ptr(),
                                                           pre.data_ptr() if pre is not None else None, B, 128, l.in_features, act, lo.weight.data_ptr(),
                                                           lo.bias.data_ptr() if lo.bias is not None else None, lo.out_features, out.data_ptr(), st),
                               "pbhc_linear_act_fwd_out")
                    h = z
                    saved_act.append(pre if pre is not None else h)
                    continue
                _lib.check(lib.pbhc_linear_act_fwd(h.data_ptr(), l.weight.data_ptr(), l.bias.data_ptr() if l.bias is not None else None, z.data_ptr(),
                                                   pre.data_ptr() if pre is not None else None, B, l.out_features, l.in_features, act, st), "pbhc_linear_act_fwd")
                h = z
                saved_act.append(pre if pre is not None else h)
                continue
            z = torch.addmm(l.bias, h, l.weight.t())
            if i == len(lin) - 1:
                out = z
            elif act == 1:
                h = F.elu_(z)
                saved_act.append(h)                    # ELU' from the output
            elif act == 3:
                h = F.relu_(z)
                saved_act.append(h)
            else:
                saved_act.append(z)                    # SiLU' needs the pre-activation
                h = F.silu(z)
        ctx.seq, ctx.act, ctx.n, ctx.fused = seq, act, len(lin), fused
        ctx.live = None
        if getattr(seq, "_grad_direct", False):
            ctx.live = _Live()
            seq._fused_live.add(ctx.live)
        ctx.save_for_backward(*saved_in, *saved_act)
        return out

    @staticmethod
    def backward(ctx, dout):
        dx, flat = _FusedMLP._bwd(ctx, dout, ctx.needs_input_grad[0])
        return (dx, None, *flat)

    @staticmethod
    def _bwd(ctx, dout, need_dx):
        """-> (input gradient or None, [gw, gb] per layer in forward order).  `ctx.grad_cols = c0`: only the input columns [c0, in) carry a
        gradient (_FusedMLPCat) — the first layer's input-gradient GEMM runs on that column slice of its weight alone."""
        lib, st = _lib.lib(), _lib.current_stream()
        lin = [m for m in ctx.seq if isinstance(m, nn.Linear)]
        L = ctx.n
        saved = ctx.saved_tensors
        ins, acts = saved[:L], saved[L:]
        d = dout.contiguous()
        B = d.shape[0]
        widths = [l.out_features for l in lin]
        MAXB = _lib.K["PBHC_ACT_MAX_BLOCKS"]
        scratch = torch.empty(MAXB * sum(widths), device=d.device)       # per-layer row-block column sums, finished by ONE launch below
        jobs = []                                                   # (part, out, row blocks, n): one per layer's bias gradient, + the output layer's
        deferred = []                                               # weight gradient, + the split-K partials of the hidden layers' weight gradients
        keep = [scratch]
        nb = C.c_int(0)
        ret_w = []
        off = 0
        # direct stores only for a declared stack with one live application; once two applications were outstanding together, every
        # backward of that batch accumulates through autograd (the first one's direct store would otherwise be overwritten by the second)
        seq = ctx.seq
        may_direct = False
        if ctx.live is not None:
            if len(seq._fused_live) > 1:
                seq._fused_shared = True
            may_direct = not seq._fused_shared and seq._stored_epoch != seq._zero_epoch      # the first backward since the owner's zeroing
            if may_direct:
                seq._stored_epoch = seq._zero_epoch
        offs = []
        for l in lin:
            offs.append(off)
            off += MAXB * l.out_features
        have_partials = False                     # the fused input-gradient GEMM of the layer above already left d = dz and its column sums
        for i in reversed(range(L)):
            l = lin[i]
            n = l.out_features
            direct = (may_direct and l.weight.requires_grad and l.bias.requires_grad and l.weight.grad is not None and l.weight.grad.is_contiguous()
                      and l.bias.grad is not None and l.bias.grad.is_contiguous())
            gb = l.bias.grad if direct else torch.empty(n, device=d.device)
            part = scratch[offs[i]:offs[i] + MAXB * n]
            k_in = l.in_features
            if (i == L - 1 and i > 0 and ctx.fused and OUT_BWD and l.weight.is_contiguous() and ins[i].is_contiguous()
                    and ((k_in in (128, 256) and 1 <= n <= 32) or (ctx.act in (1, 3) and 8 < n <= 32 and k_in in (64, 192)))):
                # the narrow output layer: weight / bias / input gradient and the activation backward of the layer below in ONE pass over the rows.
                # 128 / 256 inputs (every shipped stack): on the matrix cores, any activation, 1..32 outputs.  Other widths: the streaming form, ELU /
                # ReLU stacks with 9..32 outputs only (it measured slower than the four launches on SiLU stacks, whose derivative needs a second
                # row stream and an exp per element — 58 / 45 us for 29 / 1 outputs)
                # (`pbhc_linear_out_bwd`; autograd: two library launches of split-K for 0.14 GFLOP, a column sum, a GEMM and an activation pass)
                gw = l.weight.grad if direct else torch.empty(n, k_in, device=d.device)
                part_dw = torch.empty(MAXB * n * k_in, device=d.device)    # (a local: alive until the finishing launch below has been queued)
                dn = torch.empty(B, k_in, device=d.device)
                saved = acts[i - 1]
                _lib.check(lib.pbhc_linear_out_bwd(d.data_ptr(), ins[i].data_ptr(), None if saved.data_ptr() == ins[i].data_ptr() else saved.data_ptr(),
                                                   l.weight.data_ptr(), B, n, k_in, ctx.act, dn.data_ptr(), part_dw.data_ptr(), part.data_ptr(),
                                                   scratch[offs[i - 1]:].data_ptr(), C.byref(nb), st), "pbhc_linear_out_bwd")
                jobs.append((part.data_ptr(), gb.data_ptr(), nb.value, n))
                jobs.append((part_dw.data_ptr(), gw.data_ptr(), nb.value, n * k_in))
                keep += [part_dw, gw, gb]
                ret_w.append((None, None) if direct else (gw, gb))
                d = dn
                have_partials = True
                continue
            if not have_partials:
                _lib.check(lib.pbhc_act_bwd_partials(d.data_ptr(), acts[i].data_ptr() if i < L - 1 else None, B, n, ctx.act if i < L - 1 else 0, d.data_ptr(),
                                                     part.data_ptr(), C.byref(nb), st), "pbhc_act_bwd_partials")
            jobs.append((part.data_ptr(), gb.data_ptr(), nb.value, n))
            keep.append(gb)
            room = deferred
            if direct:
                _wgrad(d, ins[i], l.weight.grad, room)
                ret_w.append((None, None))
            else:
                ret_w.append((_wgrad(d, ins[i], torch.empty(n, l.in_features, device=d.device), room), gb))
            have_partials = False
            if i > 0 and ctx.fused and l.weight.is_contiguous() and B <= 128 * MAXB:      # (row blocks of 128 at most: the column-sum scratch holds MAXB of them)
                # d_below = (d W) * act'(output of the layer below) + its row-block column sums, in the GEMM's epilogue
                k = l.in_features
                dn = torch.empty(B, k, device=d.device)
                _lib.check(lib.pbhc_linear_dgrad_act(d.data_ptr(), l.weight.data_ptr(), acts[i - 1].data_ptr(), dn.data_ptr(),
                                                     scratch[offs[i - 1]:].data_ptr(), C.byref(nb), B, k, n, ctx.act, st), "pbhc_linear_dgrad_act")
                d = dn
                have_partials = True
            elif i > 0 or need_dx:
                d = d @ (l.weight if (i > 0 or ctx.grad_cols is None) else l.weight[:, ctx.grad_cols:])
        for part_w, out_w in deferred:                              # [P, n, k] partials -> the weight gradient: the same fixed-order column sum
            jobs.append((part_w.data_ptr(), out_w.data_ptr(), part_w.shape[0], out_w.numel()))
            keep += [part_w, out_w]
        if _DEFER["on"] and may_direct and all(gw is None for gw, _ in ret_w):
            _DEFER["jobs"] += jobs                                  # every gradient is stored in place: the owner's finish_deferred() launches
            _DEFER["keep"] += keep
        else:
            _launch_jobs(jobs, st)
        if ctx.live is not None:
            seq._fused_live.discard(ctx.live)
            ctx.live = None
            if len(seq._fused_live) == 0:
                seq._fused_shared = False
        dx = d if need_dx else None
        flat = []
        for gw, gbias in reversed(ret_w):
            flat += [gw, gbias]
        return dx, flat


class _FusedMLPCat(torch.autograd.Function):
    """The same stack applied to cat([x_const, x_grad], -1) where only `x_grad` carries a gradient — ppo_mimic's actor / critic stacks read
    [observations | encoder outputs] (ppo_mimic.py:596-630 via agent_modules.py:118-128): autograd's form computes the first layer's input
    gradient for EVERY input column (24 576 x 512 x ~700: 0.14 ms per stack and optimiser step) and CatBackward then throws the observation
    columns away.  Here the input-gradient GEMM runs on the weight's trailing columns alone and is returned for `x_grad` directly."""

    @staticmethod
    def forward(ctx, x_const, x_grad, seq, *params):
        ctx.grad_cols = x_const.shape[-1]
        return _FusedMLP._fwd(ctx, torch.cat([x_const, x_grad], dim=-1), seq)

    @staticmethod
    def backward(ctx, dout):
        dx, flat = _FusedMLP._bwd(ctx, dout, ctx.needs_input_grad[1])
        return (None, dx, None, *flat)


class _FusedMLPInto(torch.autograd.Function):
    """_FusedMLPCat without the concatenation: `xfull` [B, in] already holds the constant (observation) columns [0, c0) — written once per
    update, when the minibatch shuffle is done — and the gradient-carrying parts (encoder outputs) are copied into its columns [c0, in) here:
    19 + 13 MB of copies per optimiser step of the general-tracking update where the two `torch.cat` moved 46 + 46 MB.  `xfull` is the first
    layer's saved input: it must stay untouched until this application's backward has run (the agent rewrites a slice only when its minibatch
    comes round again, an epoch later)."""

    @staticmethod
    def forward(ctx, xfull, c0, nparts, seq, *rest):
        parts = rest[:nparts]
        o = c0
        for t in parts:
            xfull[:, o:o + t.shape[1]].copy_(t)
            o += t.shape[1]
        assert o == xfull.shape[1]
        ctx.grad_cols = c0
        ctx.part_widths = [t.shape[1] for t in parts]
        return _FusedMLP._fwd(ctx, xfull, seq)

    @staticmethod
    def backward(ctx, dout):
        need = any(ctx.needs_input_grad[4:4 + len(ctx.part_widths)])
        dx, flat = _FusedMLP._bwd(ctx, dout, need)
        dparts = [None] * len(ctx.part_widths) if dx is None else list(torch.split(dx, ctx.part_widths, dim=1))
        return (None, None, None, None, *dparts, *flat)


OUT_BWD = os.environ.get("PBHC_FUSED_OUT_BWD", "1") != "0"
FWD_OUT = os.environ.get("PBHC_FUSED_FWD_OUT", "1") != "0"
FUSED_STACK = os.environ.get("PBHC_FUSED_STACK", "1") != "0"
_STACK_MAX_ROWS = int(os.environ.get("PBHC_FUSED_STACK_MAX_ROWS", "16384"))


def pack_stack(seq):
    """Owner's declaration "the weights of `seq` stay as they are until release_stack()" (a rollout, an evaluation run): repacks them into
    the operand layout of `pbhc_mlp_fwd` (one launch per layer, ~1.5 / 3.8 MB) so that `forward_inference` can run the whole stack as ONE
    launch.  Returns False — and changes nothing — where the stack kernel does not apply."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    lib = _lib.lib()
    n = len(lin)
    if not (FUSED_GEMM and FUSED_STACK and 1 <= n <= _lib.K["PBHC_MLP_MAX_LAYERS"] and lin[0].weight.is_cuda and lin[0].weight.dtype == torch.float32
            and all(l.weight.is_contiguous() for l in lin) and supported(seq)):
        return False
    c = getattr(seq, "_pbhc_stack", None)
    sizes = [int(lib.pbhc_mlp_packed_floats(l.out_features, l.in_features)) for l in lin]
    if c is None or c["sizes"] != sizes or c["bufs"][0].device != lin[0].weight.device:
        dims = (C.c_int * (n + 1))(lin[0].in_features, *[l.out_features for l in lin])
        lds = int(lib.pbhc_mlp_fwd_lds_bytes(dims, n))
        if lds > 160 * 1024:
            return False
        bufs = [torch.empty(sz, device=lin[0].weight.device) for sz in sizes]       # allocated once: captured graphs keep reading these addresses
        c = dict(sizes=sizes, bufs=bufs, dims=dims, w=(C.c_void_p * n)(*[b.data_ptr() for b in bufs]), valid=False)
        seq._pbhc_stack = c
    st = _lib.current_stream()
    for l, buf in zip(lin, c["bufs"]):
        _lib.check(lib.pbhc_mlp_pack(l.weight.data_ptr(), l.out_features, l.in_features, buf.data_ptr(), st), "pbhc_mlp_pack")
    c["b"] = (C.c_void_p * n)(*[None if l.bias is None else l.bias.data_ptr() for l in lin])
    c["valid"] = True
    return True


def release_stack(seq):
    """the weights of `seq` may change again: `forward_inference` goes back to the layer-by-layer kernels"""
    c = getattr(seq, "_pbhc_stack", None)
    if c is not None:
        c["valid"] = False


def forward_inference(seq, x):
    """No-grad forward of the same stack (the rollout's policy / critic evaluation).  Between pack_stack() and release_stack() — the owner's
    promise that the weights are constant — the WHOLE stack is one launch (`pbhc_mlp_fwd`: a workgroup carries 16 rows through every layer,
    activations stay in LDS — at the rollout's 4 096 rows the layer-by-layer chain is launch- and tail-bound); otherwise hidden layers go
    through `pbhc_linear_act_fwd` and the narrow output layer through the library.  Plain launches on the current stream, so either form can be captured in a hipGraph."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    act = _ACT_ID[type(seq[1])] if len(lin) > 1 else 0
    lib, st = _lib.lib(), _lib.current_stream()
    # a rollout slab is a [N, C] view of 128-byte-padded rows: read in place through the row pitch (no .contiguous() copy of the slab)
    pitched = (not x.is_contiguous() and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.stride(0) % 4 == 0
               and x.data_ptr() % 16 == 0 and x.shape[1] >= 4 and len(lin) > 1)
    h = x if (x.is_contiguous() or pitched) else x.contiguous()
    B = h.shape[0]
    c = getattr(seq, "_pbhc_stack", None)
    if c is not None and c["valid"] and B <= _STACK_MAX_ROWS:
        out = torch.empty(B, lin[-1].out_features, device=h.device)
        _lib.check(lib.pbhc_mlp_fwd(h.data_ptr(), h.stride(0), c["w"], c["b"], c["dims"], len(lin), act, out.data_ptr(), out.stride(0), B, st), "pbhc_mlp_fwd")
        return out
    for l in lin[:-1]:
        z = torch.empty(B, l.out_features, device=h.device)
        bias = l.bias.data_ptr() if l.bias is not None else None
        if h.is_contiguous():
            _lib.check(lib.pbhc_linear_act_fwd(h.data_ptr(), l.weight.data_ptr(), bias, z.data_ptr(), None, B, l.out_features, l.in_features, act, st),
                       "pbhc_linear_act_fwd")
        else:
            _lib.check(lib.pbhc_linear_act_fwd_strided(h.data_ptr(), h.stride(0), 0, l.weight.data_ptr(), bias, z.data_ptr(), None, l.out_features, 0, 1,
                                                       B, l.out_features, l.in_features, act, st), "pbhc_linear_act_fwd_strided")
        h = z
    return torch.addmm(lin[-1].bias, h, lin[-1].weight.t())


def forward_sample(seq, x, std, seed, counter, counter_offset, actions, action_mean, action_sigma, logp):
    """The policy stack AND the rollout's sampling (mh_ppo.py:286-296: act(), log-prob, the buffer writes) as one launch
    (`pbhc_mlp_fwd_sample`); needs pack_stack(seq).  Returns False — nothing launched — where that does not apply (the caller then runs
    forward_inference + pbhc_policy_sample)."""
    c = getattr(seq, "_pbhc_stack", None)
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    B = x.shape[0]
    if not (c is not None and c["valid"] and B <= _STACK_MAX_ROWS and x.dim() == 2 and x.stride(1) == 1 and lin[-1].out_features <= 32
            and actions.is_contiguous() and action_mean.is_contiguous() and action_sigma.is_contiguous() and logp.is_contiguous()):
        return False
    smp = _lib.PbhcMlpSample()
    smp.std, smp.counter, smp.seed, smp.counter_offset = std.data_ptr(), counter, int(seed), int(counter_offset)
    smp.actions, smp.action_mean, smp.action_sigma, smp.logp = actions.data_ptr(), action_mean.data_ptr(), action_sigma.data_ptr(), logp.data_ptr()
    act = _ACT_ID[type(seq[1])] if len(lin) > 1 else 0
    _lib.check(_lib.lib().pbhc_mlp_fwd_sample(x.data_ptr(), x.stride(0), c["w"], c["b"], c["dims"], len(lin), act, B, C.byref(smp), _lib.current_stream()),
               "pbhc_mlp_fwd_sample")
    return True


def forward_cat_inference(seq, xs, sample=None):
    """No-grad forward of a packed stack on `torch.cat(xs, -1)` WITHOUT the concatenation (`pbhc_mlp_fwd_cat`: the stack kernel stages its 16
    input rows from up to three column segments — observation slab | encoder outputs), optionally with the rollout's sampling in the last
    layer's epilogue (`sample`: dict(std, seed, counter, counter_offset, actions, action_mean, action_sigma, logp), as forward_sample).
    Returns the output [B, out] (None with `sample`: the mean lands in action_mean), or False — nothing launched — where it does not apply."""
    c = getattr(seq, "_pbhc_stack", None)
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    B = xs[0].shape[0]
    if not (c is not None and c["valid"] and B <= _STACK_MAX_ROWS and 1 <= len(xs) <= _lib.K["PBHC_MLP_MAX_SEGS"]
            and all(x.dim() == 2 and x.stride(1) == 1 and x.shape[0] == B and x.dtype == torch.float32 and x.is_cuda for x in xs)
            and sum(x.shape[1] for x in xs) == lin[0].in_features):
        return False
    if sample is not None and not (lin[-1].out_features <= 32 and all(sample[k].is_contiguous() for k in ("actions", "action_mean", "action_sigma", "logp"))):
        return False
    inp = _lib.PbhcMlpInput()
    for i, x in enumerate(xs):
        inp.x[i], inp.ld[i], inp.width[i] = x.data_ptr(), x.stride(0), x.shape[1]
    inp.nseg = len(xs)
    act = _ACT_ID[type(seq[1])] if len(lin) > 1 else 0
    smp_ref, out = None, None
    if sample is not None:
        smp = _lib.PbhcMlpSample()
        smp.std, smp.counter, smp.seed, smp.counter_offset = sample["std"].data_ptr(), sample["counter"], int(sample["seed"]), int(sample["counter_offset"])
        smp.actions, smp.action_mean, smp.action_sigma, smp.logp = (sample[k].data_ptr() for k in ("actions", "action_mean", "action_sigma", "logp"))
        smp_ref = C.byref(smp)
    else:
        out = torch.empty(B, lin[-1].out_features, device=xs[0].device)
    _lib.check(_lib.lib().pbhc_mlp_fwd_cat(C.byref(inp), c["w"], c["b"], c["dims"], len(lin), act, None if out is None else out.data_ptr(),
                                           0 if out is None else out.stride(0), B, smp_ref, _lib.current_stream()), "pbhc_mlp_fwd_cat")
    return out if sample is None else None


def forward_into(seq, xfull, c0, parts):
    """seq(xfull) after copying `parts` (tensors [B, w_i], the gradient-carrying inputs) into xfull[:, c0:] — see _FusedMLPInto"""
    params = []
    for m in seq:
        if isinstance(m, nn.Linear):
            params += [m.weight, m.bias]
    return _FusedMLPInto.apply(xfull, c0, len(parts), seq, *parts, *params)


def forward(seq, x):
    """seq: nn.Sequential of Linear / activation; x [B, in]."""
    params = []
    for m in seq:
        if isinstance(m, nn.Linear):
            params += [m.weight, m.bias]
    return _FusedMLP.apply(x, seq, *params)


def forward_cat(seq, x_const, x_grad):
    """seq(cat([x_const, x_grad], -1)) with the input gradient formed for `x_grad` only (see _FusedMLPCat)"""
    params = []
    for m in seq:
        if isinstance(m, nn.Linear):
            params += [m.weight, m.bias]
    return _FusedMLPCat.apply(x_const, x_grad, seq, *params)
