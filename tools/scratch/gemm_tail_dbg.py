import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pbhc_amd import _lib
lib = _lib.lib()
M, N = 64, 128
torch.manual_seed(0)
for K in (23, 22, 21, 24, 20, 36, 39):
  for shape, variant in [(2, 2), (2, 1)]:
    lib.pbhc_gemm_debug_force_shape(shape | (variant << 16))
    dy = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda")
    dx = torch.empty(M, N, device="cuda"); nb = C.c_int(0)
    st = _lib.current_stream()
    _lib.check(lib.pbhc_linear_dgrad_act(dy.data_ptr(), w.data_ptr(), None, dx.data_ptr(), None, C.byref(nb), M, N, K, 0, st), "d")
    D, W = dy.double(), w.double()
    ref = D @ W
    rem = K & 3; c0 = K - rem
    msg = f"K {K} shape {shape} variant {variant}: err {(dx.double() - ref).abs().max().item():.2e}"
    if rem:
        missing = ref - D[:, c0:] @ W[c0:]
        unrot = missing + D[:, K - 4:K - 4 + rem] @ W[c0:c0 + rem]     # LDS chunk [x_{K-4}..x_{K-1}] sitting at k slots c0..c0+3 (W rows >= K are zero)
        msg += f"   vs chunk-zero {(dx.double() - missing).abs().max().item():.2e}   vs unrotated {(dx.double() - unrot).abs().max().item():.2e}"
    print(msg)
# forward (MODE 0) with K % 4 != 0
for K in (23, 630, 37):
    lib.pbhc_gemm_debug_force_shape(2 | (2 << 16))
    x = torch.randn(M, K, device="cuda"); w = torch.randn(128, K, device="cuda"); y = torch.empty(M, 128, device="cuda")
    _lib.check(lib.pbhc_linear_act_fwd(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), None, M, 128, K, 0, _lib.current_stream()), "f")
    print("fwd K", K, "err", (y.double() - x.double() @ w.double().t()).abs().max().item())
