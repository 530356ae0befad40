"""Fused fp32-MFMA Linear kernels (pbhc_linear_act_fwd / pbhc_linear_dgrad_act) against the library path they replace
(torch.addmm + F.elu_ ; d @ W + pbhc_act_bwd_partials) at the shapes of the PPO update: correctness + time per call.

  python tools/gemm_probe.py [rows] [--shape S]
"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbhc_amd import _lib                           # noqa: E402
from pbhc_amd.agents import gemm_tuning             # noqa: E402


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 24576
    shapes = [-1]
    if "--shape" in sys.argv:
        shapes = [int(s) for s in sys.argv[sys.argv.index("--shape") + 1].split(",")]
    variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 0
    if os.environ.get("PBHC_PROBE_TUNED", "1") == "1":
        gemm_tuning.enable()
    lib = _lib.lib()
    dev = "cuda"
    torch.manual_seed(0)
    fwd = [(512, 380), (256, 512), (128, 256), (768, 630), (512, 768), (128, 512)]
    print(f"rows {M}")
    tot_lib = tot_new = 0.0
    for shape in shapes:
        lib.pbhc_gemm_debug_force_shape((shape & 0xff) | (variant << 16))
        print(f"== tile shape {shape} variant {variant} ==")
        for N, K in fwd:
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) / K ** 0.5
            b = torch.randn(N, device=dev)
            y = torch.empty(M, N, device=dev)
            st = _lib.current_stream()

            def new():
                _lib.check(lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, st), "fwd")

            def old():
                return F.elu_(torch.addmm(b, x, w.t()))
            new()
            ref = F.elu((x.double() @ w.double().t() + b.double())).float()
            err_new = (y - ref).abs().max().item()
            err_old = (old() - ref).abs().max().item()
            t_new, t_old = timeit(new), timeit(old)
            fl = 2.0 * M * N * K
            tot_lib += t_old
            tot_new += t_new
            print(f"fwd   {N:4d} x {K:4d}: fused {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF/s)  library+elu {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF/s)   err {err_new:.2e} / {err_old:.2e}")
        # dgrad: dx[M, in] = dy[M, out] W[out, in] * elu'(saved[M, in]); in = width of the layer below
        for Kout, Nin in [(23, 128), (128, 256), (256, 512), (1, 128), (128, 512), (512, 768)]:
            dy = torch.randn(M, Kout, device=dev)
            w = torch.randn(Kout, Nin, device=dev) / Kout ** 0.5
            saved = F.elu(torch.randn(M, Nin, device=dev))
            dx = torch.empty(M, Nin, device=dev)
            part = torch.empty(1024 * Nin, device=dev)
            part2 = torch.empty(1024 * Nin, device=dev)
            nb = C.c_int(0)
            nb2 = C.c_int(0)
            st = _lib.current_stream()

            def new():
                _lib.check(lib.pbhc_linear_dgrad_act(dy.data_ptr(), w.data_ptr(), saved.data_ptr(), dx.data_ptr(), part.data_ptr(), C.byref(nb), M, Nin, Kout, 1, st), "dgrad")

            def old():
                d = dy @ w
                _lib.check(lib.pbhc_act_bwd_partials(d.data_ptr(), saved.data_ptr(), M, Nin, 1, d.data_ptr(), part2.data_ptr(), C.byref(nb2), st), "act")
                return d
            new()
            ref = (dy.double() @ w.double()) * torch.where(saved > 0, torch.ones_like(saved), saved + 1).double()
            err_new = (dx - ref.float()).abs().max().item()
            d_old = old()
            err_old = (d_old - ref.float()).abs().max().item()
            cs = part[:nb.value * Nin].view(nb.value, Nin).double().sum(0)
            cs_err = ((cs - ref.sum(0)).abs().max() / ref.abs().sum(0).max()).item()
            t_new, t_old = timeit(new), timeit(old)
            fl = 2.0 * M * Nin * Kout
            tot_lib += t_old
            tot_new += t_new
            print(f"dgrad {Kout:4d} -> {Nin:4d}: fused {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF/s)  library+actbwd {t_old:7.1f} us   err {err_new:.2e} / {err_old:.2e}  colsum rel {cs_err:.1e} ({nb.value} blocks)")
        # wgrad: dW[out, in] = dy^T x over the M rows (library: the split-K batched form of fused_mlp._wgrad)
        from pbhc_amd.agents.fused_mlp import _wgrad_library
        for Nout, Kin in [(512, 380), (256, 512), (128, 256), (768, 630), (512, 768), (128, 512)]:
            dy = torch.randn(M, Nout, device=dev)
            x = torch.randn(M, Kin, device=dev)
            dw = torch.empty(Nout, Kin, device=dev)
            dw2 = torch.empty(Nout, Kin, device=dev)
            P = lib.pbhc_linear_wgrad_parts(M, Nout, Kin)
            scratch = torch.empty(max(P, 1) * Nout * Kin, device=dev)
            st = _lib.current_stream()

            def new():
                _lib.check(lib.pbhc_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), scratch.data_ptr(), M, Nout, Kin, st), "wgrad")

            def old():
                return _wgrad_library(dy, x, dw2)
            new(); old()
            ref = dy.double().t() @ x.double()
            err_new = ((dw - ref).abs().max() / ref.abs().max()).item()
            err_old = ((dw2 - ref).abs().max() / ref.abs().max()).item()
            t_new, t_old = timeit(new), timeit(old)
            fl = 2.0 * M * Nout * Kin
            tot_lib += t_old
            tot_new += t_new
            print(f"wgrad {Nout:4d} x {Kin:4d}: fused {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF/s, {P} parts)  library {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF/s)   rel err {err_new:.2e} / {err_old:.2e}")
        print(f"sum: fused {tot_new:.0f} us   library {tot_lib:.0f} us")
        tot_lib = tot_new = 0.0


if __name__ == "__main__":
    main()
