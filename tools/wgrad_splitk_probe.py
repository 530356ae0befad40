"""Weight-gradient GEMMs dW[out,in] = d^T[out,B] x[B,in] with B = 24576 and a small out x in: one GEMM vs split-K as a batched GEMM + sum."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pbhc_amd.agents import gemm_tuning
gemm_tuning.enable()
dev = "cuda:0"
B = 24576

def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for out, inn in [(256, 512), (128, 256), (23, 128), (128, 512), (21, 128), (512, 380), (512, 768)]:
    d = torch.randn(B, out, device=dev); x = torch.randn(B, inn, device=dev); g = torch.empty(out, inn, device=dev)
    t0 = bench(lambda: torch.mm(d.t(), x, out=g))
    res = [f"mm {t0:6.1f} us"]
    ref = g.clone()
    for P in (8, 16, 32, 64):
        buf = torch.empty(P, out, inn, device=dev)
        def f():
            torch.bmm(d.view(P, B // P, out).transpose(1, 2), x.view(P, B // P, inn), out=buf)
            torch.sum(buf, 0, out=g)
        t = bench(f)
        err = float((g - ref).abs().max() / ref.abs().max())
        res.append(f"P={P}: {t:6.1f} us (rel diff {err:.1e})")
    print(f"out={out} in={inn}: " + "  ".join(res), flush=True)
