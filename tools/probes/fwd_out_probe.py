"""pbhc_linear_act_fwd_out (128-column hidden layer + the narrow output layer in its epilogue) at the update's 24 576 rows: ring variants"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pbhc_amd import _lib
lib = _lib.lib()
M = 24576
def timeit(fn, n=50, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K, NO in [(256, 23), (512, 21)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(128, K, device="cuda") / K ** 0.5; b = torch.randn(128, device="cuda")
    wo = torch.randn(NO, 128, device="cuda"); bo = torch.randn(NO, device="cuda")
    y = torch.empty(M, 128, device="cuda"); out = torch.empty(M, NO, device="cuda")
    st = _lib.current_stream()
    ref = None
    for var, name in [(0, "BK 32 x 2"), (2, "BK 16 x 3"), (3, "BK 16 x 4")]:
        lib.pbhc_gemm_debug_force_shape(0xff | (var << 16))
        f = lambda: lib.pbhc_linear_act_fwd_out(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, 128, K, 1, wo.data_ptr(), bo.data_ptr(), NO, out.data_ptr(), st)
        t = timeit(f)
        torch.cuda.synchronize()
        if ref is None: ref = (y.clone(), out.clone())
        print(f"fwd_out 128 x {K} (+{NO}): {name}: {t:6.1f} us  {2.0 * M * 128 * K / t / 1e6:6.1f} TF/s   same result: {torch.equal(ref[0], y) and torch.equal(ref[1], out)}")
lib.pbhc_gemm_debug_force_shape(-1)
