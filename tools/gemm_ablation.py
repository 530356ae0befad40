"""Where the fused fp32-MFMA GEMM's time goes: the K loop with parts switched off (diagnosis only; results are wrong with parts off)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbhc_amd import _lib   # noqa: E402

lib = _lib.lib()
M = 24576
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def timeit(fn, n=40, warm=20):
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


for N, K in [(512, 768), (768, 630), (128, 512)]:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    st = _lib.current_stream()
    for shape in (2, 1):
        cases = [(0, "full"), (1, "no LDS-DMA in the loop"), (16, "no epilogue"), (17, "no DMA, no epilogue"), (2, "epilogue without global stores"), (4, "C as streaming (nt) stores")]
        if shape == 2:
            cases += [(d << 5, f"co-resident workgroups de-phased by {4 * d} us") for d in (2, 4, 5, 7)]
        for dbg, name in cases:
            lib.pbhc_gemm_debug_force_shape(shape | (dbg << 8) | (VARIANT << 16))
            t = timeit(lambda: lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, st))
            lib.pbhc_gemm_debug_force_shape(shape | ((dbg | 8) << 8) | (VARIANT << 16))
            lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, st)
            torch.cuda.synchronize()
            cw = y.view(-1)[:6].tolist()
            print(f"{N}x{K} shape {shape} {name:32s} {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF/s-equivalent   one tile: {cw[1] / 100:.1f} us at {cw[0] / max(cw[1], 1) * 100:.0f} MHz")
lib.pbhc_gemm_debug_force_shape(-1)
