"""Diagnostic (GPU box): measured residuals of the two ill-conditioned quantities of the motion tables — angular velocities from
acos(2w^2-1) of a near-identity quaternion and slerp next to its sin(half angle) < 1e-3 fall-back — against the reference's goldens,
next to their conditioning.  Feeds the bounds written in tests/test_gpu_parity.py."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import GOLDEN
from tests.test_gpu_parity import _hip_motion_lib

g = dict(np.load(os.path.join(GOLDEN, "skeleton_fk_g1_23dof.npz")))
clip = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=int(g["fps"]))
sk, ml = _hip_motion_lib(clip)
D, Bx = sk.num_dof, sk.num_bodies_ext
rows = ml.frames.cpu()
F = rows.shape[0]
o = 2 * D + 2
dt = 1.0 / int(g["fps"])
gav = rows[:, o + 10 * Bx:].view(F, Bx, 3)
ref = torch.from_numpy(g["gavs_t"])
err = (gav - ref).abs()
wn = ref.norm(dim=-1, keepdim=True)
print(f"table angular velocity: max err {float(err.max()):.3e}, frac > 5e-5: {float((err > 5e-5).float().mean()):.4f}, frac > 2e-3: {float((err > 2e-3).float().mean()):.5f}")
for lo, hi in ((0, 0.05), (0.05, 0.2), (0.2, 1.0), (1.0, 100.0)):
    m = ((wn >= lo) & (wn < hi)).expand_as(err)
    if m.any():
        print(f"  |omega| in [{lo}, {hi}): n {int(m.sum())}, max err {float(err[m].max()):.3e}, max err*|omega| {float((err * wn)[m].max()):.3e}")
# conditioning: d(omega) ~ k * eps / (dt^2 * |omega|) for the raw value; the sigma=2 Gaussian filter (17 taps) averages independent errors
eps = 6e-8
from tests.test_gpu_parity import table_speed_floor
for name, w_ in (("own |omega|", wn), ("slowest |omega| in the filter window", table_speed_floor(ref))):
    for kk in (2, 4, 8, 16):
        bound = 5e-5 + kk * eps / (dt * dt * torch.clamp(w_, min=1e-3 / dt))
        print(f"  {name}, k = {kk}: elements above 5e-5 + k*eps/(dt^2 |omega|): {int((err > bound).sum())} of {err.numel()}")
rot = rows[:, o + 3 * Bx:o + 7 * Bx].view(F, Bx, 4)
print(f"table rotations: max err {float((rot - torch.from_numpy(g['grs_t'])).abs().max()):.3e}")
for tag in ("wjx_horse", "origin_walk"):
    gg = dict(np.load(os.path.join(GOLDEN, f"motion_state_{tag}.npz")))
    clip = dict(pose_aa=gg["pose_aa"], root_trans_offset=gg["root_trans_offset"], fps=int(gg["fps"]))
    if "clip_contact_mask" in gg:
        clip["contact_mask"] = gg["clip_contact_mask"]
    N = gg["times"].shape[0]
    sk, ml = _hip_motion_lib(clip, N)
    res = ml.get_motion_state(torch.arange(N, device="cuda:0"), torch.from_numpy(gg["times"]).to("cuda:0"), torch.from_numpy(gg["offset"]).to("cuda:0"))
    for k in ("rg_rot_t", "root_rot", "body_ang_vel_t", "root_ang_vel"):
        e = (res[k].cpu() - torch.from_numpy(gg[k])).abs()
        print(f"{tag}:{k}: max err {float(e.max()):.3e}, frac > 5e-5 {float((e > 5e-5).float().mean()):.4f}, frac > 3e-4 {float((e > 3e-4).float().mean()):.5f}")
