"""Backward of an encoder Conv1d layer at the update's minibatch size, piece by piece: activation backward (contiguous / between pad windows),
input gradient (per-window accumulation loop / one strided-batched GEMM per stride phase), weight gradient (windows of a contiguous / padded dz)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pbhc_amd import _lib                                   # noqa: E402
from pbhc_amd.agents import agent_modules as am             # noqa: E402

lib = _lib.lib()


def timeit(fn, n=30, warm=10):
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


for (B, T, C, O, k, s) in [(24576, 20, 60, 40, 6, 2), (24576, 8, 40, 20, 4, 2), (24576, 10, 30, 20, 4, 2)]:
    L = (T - k) // s + 1
    J = k // s
    x = torch.randn(B, T, C, device="cuda")
    w = torch.randn(O, C, k, device="cuda")
    dy = torch.randn(B, L, O, device="cuda")
    saved = torch.randn(B, L, O, device="cuda")
    st = _lib.current_stream()
    dz = torch.empty_like(dy)
    gb = torch.empty(O, device="cuda")
    scr = torch.empty(_lib.K["PBHC_ACT_MAX_BLOCKS"] * L * O, device="cuda")
    t_a0 = timeit(lambda: lib.pbhc_act_bwd_bias(dy.data_ptr(), saved.data_ptr(), B * L, O, 2, dz.data_ptr(), gb.data_ptr(), scr.data_ptr(), st))
    dzp, pad, Lp = am._padded_dz(B, L, O, J, T, s, x.device)
    dzv = dzp[:, pad:pad + L, :]
    t_a1 = timeit(lambda: dzv.copy_(dz))
    wp = w.permute(0, 2, 1).reshape(O, k * C)
    t_d0 = timeit(lambda: am._conv_window_grads(x, dz, wp, k, s, True)[0]) - timeit(lambda: am._conv_window_grads(x, dz, wp, k, s, False))
    t_d1 = timeit(lambda: am._conv_dgrad_phases(dzp, pad, w, k, s, T))
    t_w0 = timeit(lambda: am._conv_window_grads(x, dz, wp, k, s, False))
    t_w1 = timeit(lambda: am._conv_window_grads(x, dzv, wp, k, s, False))
    print(f"B {B} T {T} C {C} O {O} k {k} s {s}: act-bwd {t_a0:.1f} us, padded copy {t_a1:.1f} us   dgrad loop {t_d0:.1f} / phases {t_d1:.1f} us   wgrad {t_w0:.1f} / on padded dz {t_w1:.1f} us")
