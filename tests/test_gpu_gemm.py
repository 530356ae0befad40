"""Fused fp32-MFMA Linear kernels (`pbhc_linear_act_fwd`, `pbhc_linear_dgrad_act`, csrc/pbhc_gemm.hip) through the C ABI against an fp64
PyTorch reference of the same op (agents/modules/modules.py:47-63: nn.Linear + ELU / SiLU / ReLU and its autograd backward).

Tolerance: the kernels are f32-in / f32-accumulate k-ordered fmaf chains; against fp64 the error of an O(1) output at K <= 1024 is a few
1e-6 (same as the library GEMM they replace) — the tests allow 3e-5 absolute on outputs of magnitude O(1)."""
import ctypes as C

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 3e-5


def _act_ref(act, z):
    return {0: z, 1: F.elu(z), 2: F.silu(z), 3: F.relu(z)}[act]


def _act_grad_ref(act, z):
    z = z.detach().clone().requires_grad_(True)
    _act_ref(act, z).sum().backward()
    return z.grad


# (M, N, K): whole tiles; ragged rows / columns; K with every remainder mod 4 and below one stage; the update's and the rollout's shapes
FWD_SHAPES = [(256, 128, 64), (200, 130, 37), (96, 128, 23), (97, 23, 630), (1, 1, 5), (130, 257, 3), (64, 128, 1), (4096, 512, 380), (24576, 128, 256)]


@pytest.mark.parametrize("shape", FWD_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_linear_act_fwd_matches_fp64(shape, act):
    from pbhc_amd import _lib

    lib = _lib.lib()
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(1000 * act + M + N + K)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    for forced in (-1, 0, 1, 2, 0x10000 | 0xff, 0x20000 | 2):          # automatic; each tile shape; register-staged version; BK 16 x 3 stages
        lib.pbhc_gemm_debug_force_shape(forced if forced >= 0 else -1)
        y = torch.full((M, N), float("nan"), device="cuda")
        pre = torch.full((M, N), float("nan"), device="cuda")
        _lib.check(lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), pre.data_ptr(), M, N, K, act, _lib.current_stream()), "fwd")
        z = x.double() @ w.double().t() + b.double()
        assert (pre.double() - z).abs().max().item() < TOL, forced
        assert (y.double() - _act_ref(act, z)).abs().max().item() < TOL, forced
    lib.pbhc_gemm_debug_force_shape(-1)
    # no bias, no pre-activation output
    y = torch.empty(M, N, device="cuda")
    _lib.check(lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, M, N, K, act, _lib.current_stream()), "fwd")
    assert (y.double() - _act_ref(act, x.double() @ w.double().t())).abs().max().item() < TOL


# (M, N = in_features, K = out_features)
DGRAD_SHAPES = [(256, 128, 64), (200, 132, 37), (96, 128, 23), (97, 128, 1), (130, 256, 3), (77, 130, 40), (4096, 256, 128), (24576, 128, 23)]


@pytest.mark.parametrize("shape", DGRAD_SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_linear_dgrad_act_matches_fp64(shape, act):
    from pbhc_amd import _lib

    lib = _lib.lib()
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(7000 * act + M + N + K)
    dy = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(K, N, device="cuda", generator=g) / K ** 0.5
    zpre = torch.randn(M, N, device="cuda", generator=g)
    saved = zpre if act == 2 else _act_ref(act, zpre)                     # SiLU: pre-activation; ELU / ReLU: the activation output
    ref = (dy.double() @ w.double()) * (_act_grad_ref(act, zpre.double()) if act else 1.0)
    MAXB = _lib.K["PBHC_ACT_MAX_BLOCKS"]
    for forced in (-1, 0, 1, 2, 0x10000 | 0xff, 0x20000 | 2):
        lib.pbhc_gemm_debug_force_shape(forced if forced >= 0 else -1)
        dx = torch.full((M, N), float("nan"), device="cuda")
        part = torch.full((MAXB * N,), float("nan"), device="cuda")
        nb = C.c_int(0)
        _lib.check(lib.pbhc_linear_dgrad_act(dy.data_ptr(), w.data_ptr(), saved.data_ptr() if act else None, dx.data_ptr(), part.data_ptr(), C.byref(nb),
                                             M, N, K, act, _lib.current_stream()), "dgrad")
        assert (dx.double() - ref).abs().max().item() < TOL, forced
        cs = part[:nb.value * N].view(nb.value, N).double().sum(0)
        assert 1 <= nb.value <= MAXB and (cs - ref.sum(0)).abs().max().item() < TOL * max(1.0, M ** 0.5), forced
    lib.pbhc_gemm_debug_force_shape(-1)
    dx = torch.empty(M, N, device="cuda")                                  # no column sums requested
    _lib.check(lib.pbhc_linear_dgrad_act(dy.data_ptr(), w.data_ptr(), saved.data_ptr() if act else None, dx.data_ptr(), None, None, M, N, K, act,
                                         _lib.current_stream()), "dgrad")
    assert (dx.double() - ref).abs().max().item() < TOL


@pytest.mark.parametrize("act_cls", [nn.ELU, nn.SiLU, nn.ReLU])
def test_fused_mlp_with_and_without_the_fused_gemms(act_cls, monkeypatch):
    """fused_mlp.forward (training path of BaseModule): outputs, input gradient and every parameter gradient agree between the MFMA kernels,
    the library-GEMM path they replace and plain autograd over the nn.Sequential."""
    from pbhc_amd.agents import fused_mlp

    torch.manual_seed(3)
    seq = nn.Sequential(nn.Linear(77, 96), act_cls(), nn.Linear(96, 64), act_cls(), nn.Linear(64, 128), act_cls(), nn.Linear(128, 5)).cuda()
    x = torch.randn(300, 77, device="cuda", requires_grad=True)
    dout = torch.randn(300, 5, device="cuda")

    def run(fn):
        for p in seq.parameters():
            p.grad = None
        x.grad = None
        y = fn(x)
        y.backward(dout)
        return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in seq.parameters()]

    ref = run(seq)
    monkeypatch.setattr(fused_mlp, "FUSED_GEMM", True)
    new = run(lambda t: fused_mlp.forward(seq, t))
    monkeypatch.setattr(fused_mlp, "FUSED_GEMM", False)
    old = run(lambda t: fused_mlp.forward(seq, t))
    for a, b, c in zip(ref, new, old):
        scale = max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() < 2e-5 * scale and (a - c).abs().max().item() < 2e-5 * scale


# (M, N = out_features, K = in_features): whole tiles; K with every remainder mod 4 (the rotated straddling chunk); N not a multiple of 64;
# the update's shapes
WGRAD_SHAPES = [(256, 64, 128), (512, 128, 37), (384, 72, 630), (1024, 128, 23), (640, 4, 5), (24576, 512, 380), (24576, 128, 256), (4096, 256, 130)]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_linear_wgrad_matches_fp64(shape):
    from pbhc_amd import _lib

    lib = _lib.lib()
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dy = torch.randn(M, N, device="cuda", generator=g)
    x = torch.randn(M, K, device="cuda", generator=g)
    P = lib.pbhc_linear_wgrad_parts(M, N, K)
    assert 1 <= P <= max(1, (M // 32) // 4)
    dw = torch.full((N, K), float("nan"), device="cuda")
    scratch = torch.full((P * N * K,), float("nan"), device="cuda")
    _lib.check(lib.pbhc_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), scratch.data_ptr(), M, N, K, _lib.current_stream()), "wgrad")
    ref = dy.double().t() @ x.double()
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6 * max(1.0, (M / 1024) ** 0.5)


def test_linear_wgrad_parts_rejects_what_the_kernel_cannot_take():
    from pbhc_amd import _lib

    lib = _lib.lib()
    assert lib.pbhc_linear_wgrad_parts(24576, 23, 128) == 0          # out_features % 4
    assert lib.pbhc_linear_wgrad_parts(1000, 64, 64) == 0            # rows % 32
    assert lib.pbhc_linear_wgrad_parts(24576, 64, 3) == 0            # in_features < 4
    assert lib.pbhc_linear_wgrad_parts(24576, 768, 630) == 16 and lib.pbhc_linear_wgrad_parts(24576, 512, 768) == 21     # 512 resident slots / tiles; whole groups per XCD when cheap


@pytest.mark.parametrize("cfg", [(257, 20, 60, 40, 6, 2), (300, 10, 30, 20, 4, 2), (64, 8, 40, 20, 4, 2), (130, 6, 20, 10, 2, 1), (70, 21, 12, 8, 6, 2)])
@pytest.mark.parametrize("act_cls", [nn.ReLU, nn.SiLU, nn.ELU])
@pytest.mark.parametrize("phases", [False, True])
def test_window_conv_strided_batch_matches_conv1d(monkeypatch, cfg, act_cls, phases):
    """agents/agent_modules._WindowConv1dAct (one `pbhc_linear_act_fwd_strided` launch for the L output positions of an encoder Conv1d,
    encoder_modules.py:60-107) against act(nn.Conv1d) under autograd: output, input gradient, weight and bias gradients — with the input
    gradient as the per-window accumulation loop (default) and as one strided-batched GEMM per stride phase (PBHC_CONV_DGRAD_PHASES=1)."""
    from pbhc_amd.agents import agent_modules as am
    from pbhc_amd.agents import fused_mlp

    monkeypatch.setattr(am, "CONV_DGRAD_PHASES", phases)
    B, T, C, O, k, s = cfg
    torch.manual_seed(B + T)
    conv = nn.Conv1d(C, O, k, s).cuda()
    act = act_cls()
    x = torch.randn(B, T, C, device="cuda", requires_grad=True)
    L = (T - k) // s + 1
    dout = torch.randn(B, L, O, device="cuda")
    y_ref = act(conv(x.permute(0, 2, 1))).permute(0, 2, 1)
    y_ref.backward(dout)
    ref = [y_ref.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()]
    x.grad = None; conv.weight.grad = None; conv.bias.grad = None
    y = am._WindowConv1dAct.apply(x.contiguous(), conv.weight, conv.bias, k, s, fused_mlp._ACT_ID[act_cls])
    y.backward(dout)
    new = [y.detach(), x.grad, conv.weight.grad, conv.bias.grad]
    for a, b in zip(ref, new):
        assert a.shape == b.shape and (a - b).abs().max().item() < 3e-5 * max(1.0, a.abs().max().item())


@pytest.mark.parametrize("cfg", [(4096, 20, 58, 60, 128, "SiLU"), (257, 10, 46, 30, 64, "SiLU"), (100, 5, 13, 20, 16, "ReLU"), (33, 20, 7, 60, 32, "ELU")])
def test_conv_encoder_no_grad_path_matches_conv1d(cfg):
    """agents/agent_modules.ConvEncoder under no_grad on the GPU (the rollout: per-step Linear as one strided-batched launch straight from the
    row-padded observation slab, each Conv1d as one launch with its activation, the output Linear on the rows as they lie; with and without
    the re-laid-out weights prepared) against the module's nn.Conv1d form (encoder_modules.py:22-107) in float64."""
    from pbhc_amd.agents import agent_modules as am

    B, T, d, H, E, act = cfg
    torch.manual_seed(B + T + d)
    enc = am.ConvEncoder({"x": d}, {"input_dim": ["x"], "hidden_dim": H, "output_dim": E, "layer_config": {"type": "Conv1d", "activation": act}}, T).cuda()
    w = T * d
    slab = torch.randn(B, (w + 31) // 32 * 32 + 32, device="cuda")
    x = slab[:, :w]
    ref = am.ConvEncoder({"x": d}, {"input_dim": ["x"], "hidden_dim": H, "output_dim": E, "layer_config": {"type": "Conv1d", "activation": act}}, T).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in enc.state_dict().items()})
    ref.unfold_gemm = False
    want = ref(x.double().cpu())
    with torch.no_grad():
        assert enc._infer_ok(x)
        got = enc(x)
        enc.prepare_inference()
        got2 = enc(x)                                   # (prepared: the one-launch encoder, pbhc_conv_encoder_fwd)
        assert enc._enc_c is not None
        am.ONE_LAUNCH_ENCODER = False
        try:
            got3 = enc(x)                               # prepared weights, per-layer launches
        finally:
            am.ONE_LAUNCH_ENCODER = True
        assert torch.equal(got, got3)
        into2 = torch.zeros(3, B, E, device="cuda")
        enc(x, out=into2[1])
        assert torch.equal(into2[1], got2) and not into2[0].any() and not into2[2].any()
        enc.release_inference()
        into = torch.zeros(3, B, E, device="cuda")
        assert enc(x, out=into[1]) is not None and torch.equal(into[1], got) and not into[0].any() and not into[2].any()
    scale = max(1.0, want.abs().max().item())
    assert (got.double().cpu() - want).abs().max().item() < 2e-5 * scale
    assert (got2.double().cpu() - want).abs().max().item() < 2e-5 * scale
    with torch.enable_grad():                          # ... and the training path on the same input
        tr = enc(x.contiguous())
    assert (tr.detach().double().cpu() - want).abs().max().item() < 2e-5 * scale


# whole-stack forward (`pbhc_mlp_fwd`, csrc/pbhc_mlp.hip): (rows, layer widths, x row pitch or 0 for contiguous).  The rollout's actor and
# critic on padded slabs, ragged rows (tail workgroup), widths that are not multiples of 16 / 4 (k tails, ragged output tiles), one layer
STACKS = [(4096, [380, 512, 256, 128, 23], 384), (4096, [630, 768, 512, 128, 21], 640), (1000, [380, 512, 256, 128, 23], 0), (37, [630, 768, 512, 128, 21], 0),
          (16, [5, 7], 0), (130, [37, 50, 19, 3], 40), (64, [64, 64], 0), (33, [23, 1000, 30, 60, 17, 4], 0)]


@pytest.mark.parametrize("stack", STACKS)
@pytest.mark.parametrize("act", [1, 2, 3])
def test_mlp_stack_forward_matches_fp64(stack, act):
    from pbhc_amd import _lib

    lib = _lib.lib()
    M, dims, pitch = stack
    g = torch.Generator(device="cuda").manual_seed(77 * act + M + sum(dims))
    xs = torch.randn(M, pitch or dims[0], device="cuda", generator=g)
    x = xs[:, :dims[0]]
    Ws = [torch.randn(dims[i + 1], dims[i], device="cuda", generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [0.1 * torch.randn(dims[i + 1], device="cuda", generator=g) for i in range(len(dims) - 1)]
    n = len(Ws)
    packed = [torch.empty(lib.pbhc_mlp_packed_floats(t.shape[0], t.shape[1]), device="cuda") for t in Ws]
    for t, pk in zip(Ws, packed):
        _lib.check(lib.pbhc_mlp_pack(t.data_ptr(), t.shape[0], t.shape[1], pk.data_ptr(), _lib.current_stream()), "pbhc_mlp_pack")
    w = (C.c_void_p * n)(*[t.data_ptr() for t in packed])
    b = (C.c_void_p * n)(*[t.data_ptr() for t in bs])
    d = (C.c_int * (n + 1))(*dims)
    assert lib.pbhc_mlp_fwd_lds_bytes(d, n) <= 160 * 1024
    y = torch.full((M, dims[-1] + 3), 7.0, device="cuda")                       # wider rows: the columns beyond the output must stay untouched
    _lib.check(lib.pbhc_mlp_fwd(x.data_ptr(), x.stride(0), w, b, d, n, act, y.data_ptr(), y.stride(0), M, _lib.current_stream()), "pbhc_mlp_fwd")
    h = x.double()
    for i in range(n):
        h = h @ Ws[i].double().t() + bs[i].double()
        if i < n - 1:
            h = _act_ref(act, h)
    torch.cuda.synchronize()
    err = (y[:, :dims[-1]].double() - h).abs().max().item()
    assert err < TOL, err
    assert torch.all(y[:, dims[-1]:] == 7.0)


@pytest.mark.parametrize("case", [(4096, [272, 128, 64], [288, 128, 64], [464, 768, 512, 256, 29], 2), (100, [13, 7], [13, 9], [20, 33, 5], 1),
                                  (37, [8, 5, 6], [8, 8, 6], [19, 64, 3], 3), (64, [40], [48], [40, 16, 4], 2)])
def test_mlp_stack_forward_on_column_segments_equals_the_concatenated_input(case):
    """`pbhc_mlp_fwd_cat` (the general-tracking actor on [actor_obs | motion embedding | latent], agent_modules.py:75-84 of the reference, without
    the torch.cat copy): bit-identical to `pbhc_mlp_fwd` on the concatenated rows — on the 16-byte path (aligned segments) and the element path
    (odd widths / pitches) — and with the sampling epilogue to `pbhc_mlp_fwd_sample`."""
    from pbhc_amd import _lib
    from pbhc_amd.agents import fused_mlp

    M, widths, pitches, dims, act = case
    torch.manual_seed(M + sum(widths))
    act_mod = {1: nn.ELU, 2: nn.SiLU, 3: nn.ReLU}[act]
    layers = []
    for i in range(len(dims) - 1):
        layers += [nn.Linear(dims[i], dims[i + 1])] + ([act_mod()] if i < len(dims) - 2 else [])
    seq = nn.Sequential(*layers).cuda()
    xs = [torch.randn(M, p, device="cuda")[:, :w] for w, p in zip(widths, pitches)]
    xcat = torch.cat(xs, dim=-1)
    with torch.no_grad():
        assert fused_mlp.pack_stack(seq)
        want = fused_mlp.forward_inference(seq, xcat)
        got = fused_mlp.forward_cat_inference(seq, xs)
        assert got is not False and torch.equal(got, want)
        A = dims[-1]
        std = torch.rand(A, device="cuda") + 0.1
        ctr = torch.tensor([41.0], dtype=torch.float64, device="cuda")
        outs = []
        for fn in (lambda kw: fused_mlp.forward_sample(seq, xcat, std, 1234567, ctr.data_ptr(), 3, kw["actions"], kw["action_mean"], kw["action_sigma"], kw["logp"]),
                   lambda kw: fused_mlp.forward_cat_inference(seq, xs, sample=dict(std=std, seed=1234567, counter=ctr.data_ptr(), counter_offset=3, **kw)) is None):
            kw = dict(actions=torch.zeros(M, A, device="cuda"), action_mean=torch.zeros(M, A, device="cuda"), action_sigma=torch.zeros(M, A, device="cuda"),
                      logp=torch.zeros(M, 1, device="cuda"))
            assert fn(kw)
            outs.append(kw)
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), k
        assert torch.equal(outs[0]["action_mean"], want)
        fused_mlp.release_stack(seq)
        assert fused_mlp.forward_cat_inference(seq, xs) is False          # released: nothing launched


def test_module_inference_forward_uses_the_stack_kernel_and_matches_layers():
    """BaseModule.forward under no_grad (the rollout's path): whole-stack kernel == the layer-by-layer fused path == nn.Sequential"""
    from pbhc_amd.agents import fused_mlp

    torch.manual_seed(3)
    seq = nn.Sequential(nn.Linear(380, 512), nn.ELU(), nn.Linear(512, 256), nn.ELU(), nn.Linear(256, 128), nn.ELU(), nn.Linear(128, 23)).cuda()
    slab = torch.randn(4096, 384, device="cuda")
    x = slab[:, :380]
    with torch.no_grad():
        ref = seq(x)
        b = fused_mlp.forward_inference(seq, x)                  # layer by layer
        assert fused_mlp.pack_stack(seq)
        a = fused_mlp.forward_inference(seq, x)                  # whole stack
        seq[0].weight.mul_(2.0)                                  # the owner changes the weights: release, then the layer path sees them
        fused_mlp.release_stack(seq)
        c = fused_mlp.forward_inference(seq, x)
        ref2 = seq(x)
    torch.cuda.synchronize()
    assert (a - ref).abs().max().item() < TOL and (b - ref).abs().max().item() < TOL and (c - ref2).abs().max().item() < 2 * TOL
    assert not torch.equal(a, b)                                 # (different summation order: the two paths really are different kernels)


@pytest.mark.parametrize("shape", [(24576, 23, 128), (24576, 21, 128), (1000, 23, 128), (37, 1, 64), (4096, 32, 256), (513, 12, 192), (70001, 20, 128), (5, 9, 128), (3000, 1, 128), (24576, 29, 128), (24576, 29, 256), (24577, 1, 256), (100, 32, 256)])
@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("mfma", [1, 0])
def test_linear_out_bwd_matches_fp64(shape, act, mfma):
    """`pbhc_linear_out_bwd`: the narrow output layer's weight / bias / input gradient + the activation backward of the layer below in one pass
    (what autograd runs as mm, a column sum, mm and elu_backward / silu_backward on agents/modules/modules.py:47-63)."""
    from pbhc_amd import _lib

    lib = _lib.lib()
    M, A, K = shape
    g = torch.Generator(device="cuda").manual_seed(31 * act + M + A + K)
    z = torch.randn(M, K, device="cuda", generator=g)                   # pre-activation of the layer below
    h = _act_ref(act, z)
    dy = torch.randn(M, A, device="cuda", generator=g)
    w = torch.randn(A, K, device="cuda", generator=g) / K ** 0.5
    saved = z if act == 2 else None                                     # ELU / ReLU: derivative from the output (= h)
    MAXB = _lib.K["PBHC_ACT_MAX_BLOCKS"]
    dh = torch.empty(M, K, device="cuda")
    pdw, pdb, pcs = torch.empty(MAXB * A * K, device="cuda"), torch.empty(MAXB * A, device="cuda"), torch.empty(MAXB * K, device="cuda")
    nb = C.c_int(0)
    lib.pbhc_debug_out_bwd_variant(mfma)        # 1 (default): K = 128 ELU / ReLU on the matrix cores; 0: the streaming form for every shape
    try:
        _lib.check(lib.pbhc_linear_out_bwd(dy.data_ptr(), h.data_ptr(), None if saved is None else saved.data_ptr(), w.data_ptr(), M, A, K, act, dh.data_ptr(),
                                           pdw.data_ptr(), pdb.data_ptr(), pcs.data_ptr(), C.byref(nb), _lib.current_stream()), "pbhc_linear_out_bwd")
    finally:
        lib.pbhc_debug_out_bwd_variant(1)
    assert 1 <= nb.value <= MAXB
    jobs = (_lib._S["PbhcColsumJob"] * 3)()
    dw, db, cs = torch.empty(A, K, device="cuda"), torch.empty(A, device="cuda"), torch.empty(K, device="cuda")
    for j, (part, out, n) in enumerate([(pdw, dw, A * K), (pdb, db, A), (pcs, cs, K)]):
        jobs[j].part, jobs[j].out, jobs[j].num_row_blocks, jobs[j].n = part.data_ptr(), out.data_ptr(), nb.value, n
    _lib.check(lib.pbhc_colsum_final(jobs, 3, _lib.current_stream()), "pbhc_colsum_final")
    torch.cuda.synchronize()
    ref_dh = (dy.double() @ w.double()) * _act_grad_ref(act, z.double())
    scale = max(1.0, M ** 0.5)
    assert (dh.double() - ref_dh).abs().max().item() < TOL
    assert (dw.double() - dy.double().t() @ h.double()).abs().max().item() < TOL * scale
    assert (db.double() - dy.double().sum(0)).abs().max().item() < TOL * scale
    assert (cs.double() - ref_dh.sum(0)).abs().max().item() < TOL * scale


# (M, K = in_features of the 128-wide hidden layer, NO = outputs of the narrow layer behind it): the update's two tails, ragged rows, a K tail,
# every output-count class (a whole 32, one, more than 24)
FWD_OUT_SHAPES = [(24576, 256, 23), (24576, 512, 20), (100, 64, 23), (33, 37, 1), (4096, 256, 32), (1, 4, 29), (8192, 630, 7)]


@pytest.mark.parametrize("shape", FWD_OUT_SHAPES)
@pytest.mark.parametrize("act", [1, 2, 3])
def test_linear_act_fwd_out_matches_fp64(shape, act):
    """the last hidden layer + the narrow output layer in one launch (`pbhc_linear_act_fwd_out`): hidden activations, pre-activations and
    the output layer's result against fp64, with and without biases"""
    from pbhc_amd import _lib

    lib = _lib.lib()
    M, K, NO = shape
    g = torch.Generator(device="cuda").manual_seed(31 * act + M + K + NO)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(128, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(128, device="cuda", generator=g)
    wo = torch.randn(NO, 128, device="cuda", generator=g) / 128 ** 0.5
    bo = torch.randn(NO, device="cuda", generator=g)
    for with_bias in (True, False):
        y = torch.full((M, 128), float("nan"), device="cuda")
        pre = torch.full((M, 128), float("nan"), device="cuda")
        out = torch.full((M, NO), float("nan"), device="cuda")
        _lib.check(lib.pbhc_linear_act_fwd_out(x.data_ptr(), w.data_ptr(), b.data_ptr() if with_bias else None, y.data_ptr(), pre.data_ptr(), M, 128, K, act,
                                               wo.data_ptr(), bo.data_ptr() if with_bias else None, NO, out.data_ptr(), _lib.current_stream()), "fwd_out")
        z = x.double() @ w.double().t() + (b.double() if with_bias else 0.0)
        h = _act_ref(act, z)
        o = h @ wo.double().t() + (bo.double() if with_bias else 0.0)
        assert (pre.double() - z).abs().max().item() < TOL
        assert (y.double() - h).abs().max().item() < TOL
        assert (out.double() - o).abs().max().item() < TOL
    # the hidden layer is the same launch as pbhc_linear_act_fwd's 32 x 128 tiles: bit-identical activations
    lib.pbhc_gemm_debug_force_shape(4)
    y2 = torch.empty(M, 128, device="cuda")
    _lib.check(lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), None, y2.data_ptr(), None, M, 128, K, act, _lib.current_stream()), "fwd")
    lib.pbhc_gemm_debug_force_shape(-1)
    if K >= 4:
        assert torch.equal(y2, y)
    assert lib.pbhc_linear_act_fwd_out(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, M, 64, K, act, wo.data_ptr(), None, NO, out.data_ptr(), None) == _lib.K["PBHC_EINVAL"]
    assert lib.pbhc_linear_act_fwd_out(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, M, 128, K, act, wo.data_ptr(), None, 33, out.data_ptr(), None) == _lib.K["PBHC_EINVAL"]


def test_colsum_final_job_shapes():
    """`pbhc_colsum_final` with jobs of both shapes in one launch — tall (many partial rows, few columns: bias gradients) and wide (a few
    partial images of a weight gradient: the split-K sums) — incl. a wide job whose output is not 16-byte aligned and one with a ragged
    last block; every result equals the fp64 column sum rounded to fp32."""
    from pbhc_amd import _lib

    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(5)
    specs = [(512, 128, 0), (1024, 23 * 128, 0), (2, 768 * 630, 0), (32, 256 * 512, 0), (4, 512 * 380, 1), (16, 4096 + 4, 0), (64, 4096, 3), (8, 8192, 2), (65, 8192, 0), (7, 5, 0)]
    jobs = (_lib._S["PbhcColsumJob"] * _lib.K["PBHC_MAX_COLSUM_JOBS"])()
    parts, outs = [], []
    for j, (P, n, shift) in enumerate(specs):
        part = torch.randn(P * n + 1, device="cuda", generator=g)[(1 if j == 7 else 0):][:P * n].view(P, n)      # (job 7: an unaligned partial buffer)
        buf = torch.full((n + 8,), float("nan"), device="cuda")
        out = buf[shift:shift + n]
        jobs[j].part, jobs[j].out, jobs[j].num_row_blocks, jobs[j].n = part.data_ptr(), out.data_ptr(), P, n
        parts.append(part); outs.append((buf, out, shift))
    _lib.check(lib.pbhc_colsum_final(jobs, len(specs), _lib.current_stream()), "pbhc_colsum_final")
    torch.cuda.synchronize()
    for part, (buf, out, shift) in zip(parts, outs):
        assert torch.equal(out, part.double().sum(0).float())
        assert torch.isnan(buf[:shift]).all() and torch.isnan(buf[shift + out.numel():]).all()      # nothing written outside
