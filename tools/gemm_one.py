"""One fused-GEMM configuration in a loop (for rocprofv3 --pmc / --kernel-trace): python tools/gemm_one.py N K shape variant [dgrad]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbhc_amd import _lib   # noqa: E402

lib = _lib.lib()
N, K, shape, variant = (int(a) for a in sys.argv[1:5])
M = 24576
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda")
b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
st = _lib.current_stream()
lib.pbhc_gemm_debug_force_shape((shape & 0xff) | (variant << 16) | (int(os.environ.get('GEMM_DBG', '0')) << 8))      # GEMM_DBG: tools/gemm_ablation.py's flags
for _ in range(30):
    _lib.check(lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, N, K, 1, st), "fwd")
torch.cuda.synchronize()
