"""What a fused-GEMM LAUNCH costs beyond its tiles (ramp-up + the last round's tail): the same layer at M and 2 M rows, t(2 M) against 2 t(M), back to back."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pbhc_amd import _lib
lib = _lib.lib()
def timeit(fn, n=60, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 24576
st = _lib.current_stream()
for N, K in [(512, 380), (256, 512), (768, 630), (512, 768), (128, 256), (128, 512)]:
    x = torch.randn(2 * M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda"); y = torch.empty(2 * M, N, device="cuda")
    one = lambda m=M: lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, m, N, K, 1, st)
    t1 = timeit(lambda: one(M)); t2 = timeit(lambda: one(2 * M))
    def two():
        lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, st)
        lib.pbhc_linear_act_fwd(x[M:].data_ptr(), w.data_ptr(), b.data_ptr(), y[M:].data_ptr(), None, M, N, K, 1, st)
    t11 = timeit(two)
    print(f"fwd {N:4d} x {K:4d}: one launch of M {t1:6.1f} us, two launches of M {t11:6.1f} us, one launch of 2 M {t2:6.1f} us -> a launch costs {t11 - t2:5.1f} us beyond its tiles")
