"""Whole-stack forward kernel (`pbhc_mlp_fwd`) against the layer-by-layer fused chain it replaces in the rollout, run alone at N rows:
microseconds per forward (HIP events over 200 back-to-back launches) and TFLOP/s, for the v1 actor and critic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from pbhc_amd.agents import fused_mlp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def stack(dims):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            layers.append(nn.ELU())
    return nn.Sequential(*layers).cuda()


def time_us(f, reps=200):
    for _ in range(10):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, dims, pitch in [("actor", [380, 512, 256, 128, 23], 384), ("critic", [630, 768, 512, 128, 21], 640)]:
    seq = stack(dims)
    x = torch.randn(N, pitch, device="cuda")[:, :dims[0]]
    flops = 2.0 * N * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    with torch.no_grad():
        t0 = time_us(lambda: fused_mlp.forward_inference(seq, x))
        assert fused_mlp.pack_stack(seq.module if hasattr(seq, "module") else seq)
        tp = time_us(lambda: fused_mlp.pack_stack(seq), reps=50)
        t1 = time_us(lambda: fused_mlp.forward_inference(seq, x))
        fused_mlp.release_stack(seq)
    print(f"{name:6s} {N} rows {dims}: whole-stack kernel {t1:7.1f} us ({flops / t1 * 1e-6:6.1f} TFLOP/s)   layer chain {t0:7.1f} us ({flops / t0 * 1e-6:6.1f} TFLOP/s)   repack {tp:5.1f} us")
