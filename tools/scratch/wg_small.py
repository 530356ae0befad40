import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pbhc_amd.agents import gemm_tuning
gemm_tuning.enable()
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 24576
for n, k in [(29, 256), (1, 256), (23, 128), (20, 128)]:
    d = torch.randn(B, n, device="cuda"); x = torch.randn(B, k, device="cuda"); out = torch.empty(n, k, device="cuda")
    t0 = timeit(lambda: torch.mm(d.t(), x, out=out))
    res = [f"mm {t0:.1f}"]
    for P in (8, 16, 32, 64):
        t = timeit(lambda: torch.sum(torch.bmm(d.view(P, B // P, n).transpose(1, 2), x.view(P, B // P, k)), 0, out=out))
        res.append(f"P{P} {t:.1f}")
    print(n, k, " ".join(res))
n, k = 1, 256
d = torch.randn(B, n, device="cuda"); x = torch.randn(B, k, device="cuda"); out = torch.empty(n, k, device="cuda")
print("mv", timeit(lambda: torch.mv(x.t(), d.view(-1), out=out.view(-1))), "mm", timeit(lambda: torch.mm(d.t(), x, out=out)))
ref = d.double().t() @ x.double()
torch.mv(x.t(), d.view(-1), out=out.view(-1)); print("err", (out.double() - ref).abs().max().item() / ref.abs().max().item())
