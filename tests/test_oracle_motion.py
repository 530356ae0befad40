"""oracle skeleton / FK / motion-state vs outputs of the reference's Humanoid_Batch and MotionLib
(tests/golden/skeleton_fk_*.npz, motion_state_*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle.fk import motion_fk, sim_fk
from oracle.motion_lib import MotionLib
from oracle import rotations as R


def _skel(g):
    return dict(parents=g["parents"], offsets=g["offsets"], local_rot_wxyz=g["local_rot_wxyz"],
                dof_axis=g["dof_axis"].astype(np.float32), num_bodies=int(g["num_bodies"]),
                body_names_ext=[str(x) for x in g["body_names"]])


@pytest.mark.parametrize("robot", ["g1_23dof", "g1_29dof"])
def test_motion_fk(golden_dir, robot):
    g = dict(np.load(os.path.join(golden_dir, f"skeleton_fk_{robot}.npz")))
    skel = _skel(g)
    out = motion_fk(skel, g["pose_aa"], g["root_trans_offset"], 1.0 / int(g["fps"]))
    for k, tol in [("gts_t", 2e-6), ("grs_t", 2e-6), ("gvs_t", 2e-5), ("gavs_t", 2e-4), ("dof_pos", 1e-6), ("dof_vel", 1e-5), ("local_rot", 1e-6)]:
        a, b = out[k], torch.from_numpy(g[k])
        assert a.shape == b.shape, k
        assert torch.allclose(a, b, rtol=tol, atol=tol), (k, float((a - b).abs().max()))


@pytest.mark.parametrize("robot", ["g1_23dof", "g1_29dof"])
def test_sim_fk_pose_matches_reference_chain(golden_dir, robot):
    """sim_fk(root, q) must reproduce the reference FK's body poses when q is the clip's dof_pos
    (pose_aa = axis*q identity, reference motion_tracking.py:919), and its twist must match a
    finite difference of that FK."""
    g = dict(np.load(os.path.join(golden_dir, f"skeleton_fk_{robot}.npz")))
    skel = _skel(g)
    B = skel["num_bodies"]
    F = g["gts_t"].shape[0]
    root = torch.zeros(F, 13)
    root[:, 0:3] = torch.from_numpy(g["gts_t"][:, 0])
    root[:, 3:7] = torch.from_numpy(g["grs_t"][:, 0])
    q = torch.from_numpy(g["dof_pos"])
    p, r, v, w = sim_fk(skel, root, q, torch.zeros_like(q))
    assert torch.allclose(p, torch.from_numpy(g["gts_t"][:, :B]), atol=3e-6)
    ref_r = torch.from_numpy(g["grs_t"][:, :B])
    sign = torch.sign((r * ref_r).sum(-1, keepdim=True))
    assert torch.allclose(r * sign, ref_r, atol=3e-6)
    # twist: central finite difference in double precision of the pose map
    eps = 1e-3
    qd = torch.randn(F, q.shape[1], generator=torch.Generator().manual_seed(0))
    lin = torch.randn(F, 3, generator=torch.Generator().manual_seed(1))
    ang = torch.randn(F, 3, generator=torch.Generator().manual_seed(2))
    root[:, 7:10] = lin
    root[:, 10:13] = ang
    _, _, v, w = sim_fk(skel, root, q, qd)

    def pose_at(h):
        rt = root.clone()
        rt[:, 0:3] = root[:, 0:3] + h * lin
        dq = R.quat_from_angle_axis(ang.norm(dim=-1) * h, ang)
        rt[:, 3:7] = R.quat_mul(dq, root[:, 3:7])
        return sim_fk(skel, rt, q + h * qd, qd)[:2]

    p1, r1 = pose_at(eps)
    p0, r0 = pose_at(-eps)
    v_fd = (p1 - p0) / (2 * eps)
    assert torch.allclose(v, v_fd, atol=5e-3), float((v - v_fd).abs().max())
    dq = R.quat_mul(r1, R.quat_conjugate(r0))
    dq = dq * torch.sign(dq[..., 3:4])
    w_fd = 2 * dq[..., :3] / (2 * eps)
    assert torch.allclose(w, w_fd, atol=1e-2), float((w - w_fd).abs().max())


@pytest.mark.parametrize("tag", ["wjx_horse", "origin_walk"])
def test_motion_state(golden_dir, tag):
    g = dict(np.load(os.path.join(golden_dir, f"motion_state_{tag}.npz")))
    sk = dict(np.load(os.path.join(golden_dir, "skeleton_fk_g1_23dof.npz")))
    clip = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=int(g["fps"]))
    if "clip_contact_mask" in g:
        clip["contact_mask"] = g["clip_contact_mask"]
    ml = MotionLib(_skel(sk), [clip])
    assert abs(float(ml.motion_len[0]) - float(g["motion_len"])) < 1e-6
    N = g["times"].shape[0]
    res = ml.get_motion_state(torch.zeros(N, dtype=torch.long), torch.from_numpy(g["times"]), torch.from_numpy(g["offset"]))
    for k in ["root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos_t", "rg_rot_t", "body_vel_t", "body_ang_vel_t"] + (["contact_mask"] if "contact_mask" in g else []):
        a, b = res[k], torch.from_numpy(g[k]).float()
        assert a.shape == b.shape, k
        assert torch.allclose(a, b, rtol=1e-5, atol=2e-5), (k, float((a - b).abs().max()))
    B = int(sk["num_bodies"])
    assert np.allclose(g["rg_pos"], g["rg_pos_t"][:, :B]) and np.allclose(g["body_vel"], g["body_vel_t"][:, :B], atol=1e-6)


def test_target_heading_rebase_matches_reference(golden_dir):
    """load_motions(target_heading=...) (motion_lib_base.py:445-456): the host-side re-basing of pbhc_amd.motion_lib.rebase_heading, pushed
    through the (pinned) oracle FK, reproduces the tables the reference built with the same target heading."""
    from pbhc_amd.motion_lib import rebase_heading

    g = dict(np.load(os.path.join(golden_dir, "motion_target_heading_walk.npz")))
    clip = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=int(g["fps"]))
    reb = rebase_heading(clip, g["target_heading"])
    skel = dict(np.load(os.path.join(golden_dir, "skeleton_fk_g1_23dof.npz")))
    from tests.helpers import skel_from_golden

    t = motion_fk(skel_from_golden(), reb["pose_aa"], reb["root_trans_offset"], 1.0 / int(g["fps"]))
    assert torch.allclose(t["gts_t"], torch.from_numpy(g["gts_t"]), atol=5e-6)
    assert torch.allclose(t["grs_t"].abs(), torch.from_numpy(g["grs_t"]).abs(), atol=5e-6)
    assert torch.allclose(t["dof_pos"], torch.from_numpy(g["dof_pos"]), atol=2e-6)
    assert torch.allclose(t["gvs_t"], torch.from_numpy(g["gvs_t"]), atol=1e-4)
    # the heading of the first root rotation is the target's
    q = t["grs_t"][0, 0]
    head = R.calc_heading(q[None])[0]
    assert abs(float(head) - 0.7) < 1e-5
