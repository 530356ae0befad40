"""Forward kinematics.  Test infrastructure.

(1) `motion_fk`: the reference's load-time FK of a motion clip (Humanoid_Batch.fk_batch +
    forward_kinematics_batch + _compute_velocity + _compute_angular_velocity, reference:
    humanoidverse/utils/motion_lib/torch_humanoid_batch.py:168-290).
(2) `sim_fk`: rigid-body pose + twist of every body from (root state, q, q-dot).  Isaac Gym
    did this implicitly in the reference (simulator/isaacgym/isaacgym.py:574-618 exposes the
    result); the replay sim-stub has to do it.  Pose uses the same serial chain as (1) with
    pose_aa = axis * q (the identity the reference itself uses when saving rollouts,
    envs/motion_tracking/motion_tracking.py:919); twist is propagated analytically:
    w_i = w_par + R_i axis_i qd_i,  v_i = v_par + w_par x (p_i - p_par).
"""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter1d

from . import rotations as R


def _chain(skel, rot_mats, root_pos):
    """rot_mats [..., Bx, 3, 3] local joint rotations (index 0 = root, world), root_pos [..., 3]."""
    parents = skel["parents"]
    offsets = torch.as_tensor(skel["offsets"])
    local = R.quaternion_to_matrix_wxyz(torch.as_tensor(skel["local_rot_wxyz"])).float()
    pos, rot = [], []
    for i in range(len(parents)):
        p = int(parents[i])
        if p == -1:
            pos.append(root_pos)
            rot.append(rot_mats[..., 0, :, :])
        else:
            # reference :251-252
            jpos = torch.matmul(rot[p], offsets[i][..., None]).squeeze(-1) + pos[p]
            rmat = torch.matmul(rot[p], torch.matmul(local[i], rot_mats[..., i, :, :]))
            pos.append(jpos)
            rot.append(rmat)
    return torch.stack(pos, dim=-2), torch.stack(rot, dim=-3)


def compute_velocity(p, dt):
    # reference :272-279 ; p [F, Bx, 3]
    v = np.gradient(p.numpy(), axis=-3) / dt
    return torch.from_numpy(gaussian_filter1d(v, 2, axis=-3, mode="nearest")).to(p)


def compute_angular_velocity(r_xyzw, dt):
    # reference :282-290 ; r [F, Bx, 4] xyzw
    diff = torch.zeros_like(r_xyzw)
    diff[..., 3] = 1.0
    d = R.quat_mul(r_xyzw[1:], R.quat_conjugate(r_xyzw[:-1]))
    diff[:-1] = R.normalize(d)
    angle, axis = R.quat_angle_axis(diff)
    w = axis * angle.unsqueeze(-1) / dt
    return torch.from_numpy(gaussian_filter1d(w.numpy(), 2, axis=-3, mode="nearest"))


def motion_fk(skel, pose_aa, trans, dt):
    """pose_aa [F, Bx, 3], trans [F, 3] -> dict of per-frame tables (extended bodies included)."""
    pose_aa = torch.as_tensor(pose_aa).float()[:, : len(skel["parents"])]
    trans = torch.as_tensor(trans).float()
    B = skel["num_bodies"]
    q = R.axis_angle_to_quaternion_wxyz(pose_aa)
    m = R.quaternion_to_matrix_wxyz(q)
    pos, mat = _chain(skel, m, trans)
    rot = R.wxyz_to_xyzw(R.matrix_to_quaternion_wxyz(mat))
    out = dict(
        gts_t=pos, grs_t=rot,
        gvs_t=compute_velocity(pos, dt), gavs_t=compute_angular_velocity(rot, dt),
    )
    dof_pos = pose_aa.sum(dim=-1)[:, 1:B]                       # reference :216
    dv = (dof_pos[1:] - dof_pos[:-1]) / dt                      # reference :223-224
    out["dof_pos"] = dof_pos
    out["dof_vel"] = torch.cat([dv, dv[-2:-1]], dim=0)      # sic: the reference repeats dv[F-3] (`dof_vel[:, -2:-1]`)
    out["local_rot"] = R.wxyz_to_xyzw(q)
    return out


def sim_fk(skel, root_state, dof_pos, dof_vel):
    """root_state [N,13] (pos, quat xyzw, lin vel, ang vel; world frame), dof_pos/vel [N,D]
    -> body pos [N,B,3], rot xyzw [N,B,4], lin vel [N,B,3], ang vel [N,B,3] for the B real bodies."""
    B = skel["num_bodies"]
    parents = skel["parents"][:B]
    axis = torch.as_tensor(skel["dof_axis"]).float()            # [D,3], dof d drives body d+1
    offsets = torch.as_tensor(skel["offsets"]).float()
    local_q = R.wxyz_to_xyzw(torch.as_tensor(skel["local_rot_wxyz"]).float())
    N = root_state.shape[0]
    pos = [root_state[:, 0:3]]
    # the chain hangs on the UNIT root rotation (a replay frame's quaternion can be off unit length by the reference slerp's scale error; the
    # kernels do the same: csrc/pbhc_env_step.h fk_walk); the root body itself keeps the frame's quaternion (restored below)
    rot = [R.normalize(root_state[:, 3:7])]
    vel = [root_state[:, 7:10]]
    ang = [root_state[:, 10:13]]
    for i in range(1, B):
        p = int(parents[i])
        off = offsets[i].expand(N, 3)
        p_i = pos[p] + R.quat_rotate(rot[p], off)
        qj = R.quat_from_angle_axis(dof_pos[:, i - 1], axis[i - 1].expand(N, 3))
        q_i = R.quat_mul(rot[p], R.quat_mul(local_q[i].expand(N, 4), qj))
        q_i = R.normalize(q_i)
        w_i = ang[p] + R.quat_rotate(q_i, axis[i - 1].expand(N, 3)) * dof_vel[:, i - 1 : i]
        v_i = vel[p] + torch.cross(ang[p], p_i - pos[p], dim=-1)
        pos.append(p_i); rot.append(q_i); vel.append(v_i); ang.append(w_i)
    rot[0] = root_state[:, 3:7]
    return torch.stack(pos, 1), torch.stack(rot, 1), torch.stack(vel, 1), torch.stack(ang, 1)
