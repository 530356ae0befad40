"""Sustained fp32 MFMA rate + shader clock of this MI355X (the ceiling the fused GEMMs are priced against)."""
import ctypes as C
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "probes", "libmfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(here, "probes", "mfma_peak.hip"), "-o", so])
lib = C.CDLL(so)
lib.mfma_peak.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
for blocks, nacc in [(256, 4), (512, 4), (768, 4), (512, 2)]:
    out = torch.zeros(blocks * 256, device="cuda")
    clk = torch.zeros(blocks * 2, dtype=torch.int64, device="cuda")
    for iters in (2000, 8000):
        st = torch.cuda.current_stream().cuda_stream
        lib.mfma_peak(out.data_ptr(), clk.data_ptr(), blocks, 100, nacc, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.mfma_peak(out.data_ptr(), clk.data_ptr(), blocks, iters, nacc, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = blocks * 4 * iters * 8 * nacc * 4096.0
        c = clk.view(blocks, 2).double()
        print(f"blocks {blocks} acc {nacc} iters {iters}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s   clock64/wall_clock64 ratio {(c[:, 0] / c[:, 1]).mean().item():.3f}  clock64 per us {(c[:, 0].mean() / (ms * 1e3)).item():.1f}")
