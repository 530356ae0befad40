"""LeggedRobotGeneralTracking.step (KungfuBot2 / general tracking) restated on explicit state tensors (torch CPU fp32).
TEST INFRASTRUCTURE (see oracle/__init__.py).

Same step skeleton as oracle.env_v1 (legged_robot_base.py:239-338); what differs follows
humanoidverse/envs/motion_tracking/general_tracking.py:
  _get_future_motion_targets            :500-565   20 reference frames at t + {1..95} dt, anchor-local key bodies
  _pre_compute_observations_callback    :568-805   TRUE quaternion differences, root diffs, anchor-relative ("beyondmimic")
                                                    frames incl. the aliasing quirk at :748-749, 6-D rotations
  _update_reset_buf / _update_timeout_buf :226-262 ref_pos_z / ref_ori / body_z (hard-coded bodies [4,10,24,25,26])
  rewards                               :1009-1279 key-body / anchor / root terms (angle via quat_to_angle_axis)
  _reset_dofs / _reset_root_states      :411-483   dofs from t = ep_len*dt + start, root from the (ep_len+1)*dt frame
"""
import numpy as np
import torch

from . import rotations as R
from .env_v1 import MotionTrackingOracle, _f


def yaw_quat(q):
    # reference: humanoidverse/utils/torch_utils.py:239-270
    qx, qy, qz, qw = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    yaw = torch.atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz))
    out = torch.zeros_like(q)
    out[..., 2] = torch.sin(yaw / 2)
    out[..., 3] = torch.cos(yaw / 2)
    return R.normalize(out)


def matrix_from_quat(q):
    # reference: torch_utils.py:274-296 (xyzw)
    i, j, k, r = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


class GeneralTrackingOracle(MotionTrackingOracle):
    def __init__(self, cfg, skel, motion_lib, num_envs, sim_dr):
        super().__init__(cfg, skel, motion_lib, num_envs, sim_dr)
        rc = cfg.robot
        self.key_id = [self.body_list.index(l) for l in rc.key_bodies]
        anchor = rc.motion.get("anchor_link", "pelvis_link")
        self.anchor = (self.body_names.index(anchor) if anchor in self.body_names else -1) + 1       # find_rigid_body_indice(...) + 1, sic (:97-98)
        ob = cfg.obs
        self.tar_steps = torch.linspace(1, ob.future_max_steps, ob.future_num_steps).long() if ob.get("future_num_steps", 0) > 0 else None
        self.slot_clip = torch.zeros(num_envs, dtype=torch.long)
        if not self.ec.use_vec_reward:
            self.R = 1

    # ---- reference lookups go through slot -> clip -------------------------------------------------
    def _lookup(self, times, offset):
        if times.dim() == 1:
            return self.ml.get_motion_state(self.slot_clip, times, offset=offset)
        S = times.shape[1]
        return self.ml.get_motion_state(self.slot_clip[:, None].expand(-1, S), times, offset=offset)

    def _future_targets(self):
        s, N = self.s, self.N
        S = self.tar_steps.numel()
        mt = s["episode_length_buf"] * self.dt + s["motion_start_times"]
        times = self.tar_steps * self.dt + mt[:, None]
        self.fut_times = times                                   # (kept for the tests' conditioned tolerances: which frame pairs were blended)
        ref = self._lookup(times, self.env_origins[:, None, :].expand(-1, S, -1))
        root_rot, root_pos = ref["root_rot"], ref["root_pos"]
        flat_rot = root_rot.reshape(N * S, 4)
        rpy = R.get_euler_xyz(flat_rot)
        roll_pitch = rpy[:, :2].reshape(N, S, 2)
        rvel = R.quat_rotate_inverse(flat_rot, ref["root_vel"].reshape(N * S, 3)).view(N, S, 3)
        rang = R.quat_rotate_inverse(flat_rot, ref["root_ang_vel"].reshape(N * S, 3)).view(N, S, 3)
        pos, rot = ref["rg_pos_t"], ref["rg_rot_t"]
        apos = pos[..., self.anchor, :][..., None, :].expand(-1, -1, self.Bx, -1)
        aquat = rot[..., self.anchor, :][..., None, :].expand(-1, -1, self.Bx, -1)
        local_key = R.quat_apply(R.quat_conjugate(aquat), pos - apos)[..., self.key_id, :].reshape(N, S, -1)
        f = {}
        f["future_motion_root_height"] = root_pos[..., 2:3].reshape(N, -1)
        f["future_motion_roll_pitch"] = roll_pitch.reshape(N, -1)
        f["future_motion_base_lin_vel"] = rvel.reshape(N, -1)
        f["future_motion_base_yaw_vel"] = rang[..., 2:3].reshape(N, -1)
        f["future_motion_base_ang_vel"] = rang.reshape(N, -1)
        f["future_motion_dof_pos"] = ref["dof_pos"].reshape(N, -1)
        f["future_motion_local_ref_key_body_pos"] = local_key.reshape(N, -1)
        f["next_step_ref_motion"] = torch.cat((root_pos[:, 0, 2:3], roll_pitch[:, 0, :], rvel[:, 0, :], rang[:, 0, 2:3], ref["dof_pos"][:, 0, :], local_key[:, 0, :]), dim=-1)
        self.fut = f

    # ------------------------------------------------------------------------------------
    def step(self, actions, frame, body_state, u_rfi=None, reset_samples=None):
        s, N, D = self.s, self.N, self.D
        ec, cfg = self.ec, self.cfg
        log = self.log
        clipv = cfg.robot.control.action_clip_value
        s["actions"] = torch.clip(actions, -clipv, clipv)
        log["action_clip_frac"] = (s["actions"].abs() == clipv).sum() / s["actions"].numel()
        if cfg.domain_rand.randomize_ctrl_delay:
            s["action_queue"][:, 1:] = s["action_queue"][:, :-1].clone()
            s["action_queue"][:, 0] = s["actions"]
            s["actions_after_delay"] = s["action_queue"][torch.arange(N), s["action_delay_idx"]].clone()
        else:
            s["actions_after_delay"] = s["actions"].clone()
        if u_rfi is None:
            u_rfi = torch.full((N, D), 0.5)
        s["torques"] = self.compute_torques(s["actions_after_delay"], u_rfi)
        s["root_states"] = frame["root"].clone()
        s["dof_pos"] = frame["dof_pos"].clone()
        s["dof_vel"] = frame["dof_vel"].clone()
        s["contact_forces"] = frame["contact"].clone()
        bpos, brot, bvel, bang = body_state
        s["episode_length_buf"] = s["episode_length_buf"] + 1
        s["last_episode_length_buf"] = s["episode_length_buf"].clone()
        base_quat = s["root_states"][:, 3:7]
        self.rpy = R.get_euler_xyz(base_quat)
        self.base_lin_vel = R.quat_rotate_inverse(base_quat, s["root_states"][:, 7:10])
        self.base_ang_vel = R.quat_rotate_inverse(base_quat, s["root_states"][:, 10:13])
        self.projected_gravity = R.quat_rotate_inverse(base_quat, self.gravity_vec)
        s["contacts"] = (s["contact_forces"][:, self.feet, :].norm(dim=-1) > 1.0).float()
        s["contacts_filt"] = torch.logical_or(s["contacts"], s["last_contacts"]).float()
        # ---- tracking part (general_tracking.py:568-805)
        motion_times = self._motion_times(1)
        ref = self._lookup(motion_times, self.env_origins)
        if self.tar_steps is not None:
            self._future_targets()
        if self.ml.has_contact_mask:
            self.ref_contact_mask = ref["contact_mask"]
        ref_pos, ref_rot, ref_vel, ref_ang = ref["rg_pos_t"], ref["rg_rot_t"], ref["body_vel_t"], ref["body_ang_vel_t"]
        E = len(self.ext_parent)
        Bx = self.Bx
        par_rot = brot[:, self.ext_parent]
        ext_off = self.ext_pos.repeat(N, 1, 1)
        rotated = R.quat_rotate(par_rot.reshape(-1, 4), ext_off.reshape(-1, 3))
        ext_pos = R.quat_rotate(self.ext_rot.repeat(N, 1, 1).reshape(-1, 4), rotated).view(N, E, 3) + bpos[:, self.ext_parent]
        pos_x = torch.cat([bpos, ext_pos], dim=1)
        rot_x = torch.cat([brot, R.quat_mul(par_rot.reshape(-1, 4), self.ext_rot.repeat(N, 1, 1).reshape(-1, 4)).view(N, E, 4)], dim=1)
        ang_x = torch.cat([bang, bang[:, self.ext_parent]], dim=1)
        vel_x = torch.cat([bvel, bvel[:, self.ext_parent] + torch.cross(bang[:, self.ext_parent], ext_off, dim=2)], dim=1)
        self.body_x = (pos_x, rot_x, vel_x, ang_x)
        self.obs_root_height = s["root_states"][:, 2:3].clone()
        self.obs_roll_pitch = self.rpy[:, :2]
        self.ref_body_pos_extend, self.ref_body_rot_extend = ref_pos, ref_rot
        self.dif_pos = ref_pos - pos_x
        self.dif_rot = R.quat_mul(ref_rot, R.quat_conjugate(rot_x))
        self.dif_vel = ref_vel - vel_x
        self.dif_ang = ref_ang - ang_x
        self.dif_joint_angles = ref["dof_pos"] - s["dof_pos"]
        self.dif_joint_vel = ref["dof_vel"] - s["dof_vel"]
        self.dif_root_velocity = R.quat_rotate_inverse(ref["root_rot"], ref["root_vel"]) - self.base_lin_vel
        self.dif_root_rot = R.quat_mul(ref["root_rot"], R.quat_conjugate(s["root_states"][:, 3:7]))
        self.dif_root_height = ref["root_pos"][:, 2:3] - self.obs_root_height
        heading_inv = R.calc_heading_quat_inv(s["root_states"][:, 3:7])
        hexp = heading_inv.unsqueeze(1).expand(-1, Bx, -1).reshape(-1, 4)
        self.relyaw = self.rpy[:, 2:3] - self.ref_init_yaw
        dl = R.quat_rotate(hexp, self.dif_pos.reshape(-1, 3))
        self.obs_dif_local_rigid_body_pos = dl.view(N, -1)
        self.obs_dif_local_key_body_pos = dl.view(N, -1, 3)[:, self.key_id].reshape(N, -1)
        lr = R.quat_rotate(hexp, (ref_pos - s["root_states"][:, :3].view(N, 1, 3)).reshape(-1, 3))
        self.obs_local_ref_rigid_body_pos = lr.view(N, -1)
        self.obs_local_ref_key_body_pos = lr.view(N, -1, 3)[:, self.key_id].reshape(N, -1)
        vr = ref_pos[:, self.track_id, :] - s["root_states"][:, 0:3].view(N, 1, 3)
        self.obs_vr_3point_pos = R.quat_rotate(heading_inv.repeat(3, 1), vr.reshape(-1, 3)).view(N, -1)
        self.ref_motion_phase = (motion_times / self.ml.motion_len[self.slot_clip]).unsqueeze(1)
        # ---- anchor-relative frames (:738-803)
        a = self.anchor
        anchor_pos = ref_pos[:, a, :][:, None, :].expand(-1, Bx, -1)
        anchor_quat = ref_rot[:, a, :][:, None, :].expand(-1, Bx, -1)
        robot_anchor_pos = pos_x[:, a, :][:, None, :].repeat(1, Bx, 1)
        robot_anchor_quat = rot_x[:, a, :][:, None, :].expand(-1, Bx, -1)
        delta_pos = robot_anchor_pos                      # alias, sic (:748): the z overwrite below also changes robot_anchor_pos
        delta_pos[..., 2] = anchor_pos[..., 2]
        delta_ori = yaw_quat(R.quat_mul(robot_anchor_quat, R.quat_conjugate(anchor_quat)))
        self.body_pos_relative_w = delta_pos + R.quat_apply(delta_ori, ref_pos - anchor_pos)
        self.body_quat_relative_w = R.quat_mul(delta_ori, ref_rot)
        self.dif_local_body_pos = self.body_pos_relative_w - pos_x
        self.dif_local_body_rot = R.quat_mul(self.body_quat_relative_w, R.quat_conjugate(rot_x))
        inv_ra = R.quat_conjugate(robot_anchor_quat)
        self.obs_local_body_rot = matrix_from_quat(R.quat_mul(inv_ra, rot_x))[..., :2]
        self.obs_local_body_pos = R.quat_apply(inv_ra, pos_x - robot_anchor_pos)          # (robot x, robot y, REF z) subtracted, sic (:779-782)
        inv_a = R.quat_conjugate(rot_x[:, a, :])
        self.obs_anchor_ref_rot = matrix_from_quat(R.quat_mul(inv_a, ref_rot[:, a, :]))[..., :2]
        self.obs_anchor_ref_pos = R.quat_apply(inv_a, ref_pos[:, a, :] - pos_x[:, a, :])
        self.dif_anchor_body_pos = self.dif_pos[:, a, :]
        self.dif_anchor_pos_z = ref_pos[:, a, -1] - pos_x[:, a, -1]
        self.dif_anchor_ori = R.quat_rotate_inverse(ref_rot[:, a, :], self.gravity_vec)[:, 2] - R.quat_rotate_inverse(rot_x[:, a, :], self.gravity_vec)[:, 2]
        log["upper_body_diff_norm"] = self.dif_pos[:, self.upper_id].norm(dim=-1).mean()
        log["lower_body_diff_norm"] = self.dif_pos[:, self.lower_id].norm(dim=-1).mean()
        log["key_body_diff_norm"] = self.dif_pos[:, self.key_id].norm(dim=-1).mean()
        log["joint_pos_diff_norm"] = self.dif_joint_angles.norm(dim=-1).mean()
        log["local_key_body_diff_norm"] = self.dif_local_body_pos[:, self.key_id].norm(dim=-1).mean()
        # ---- termination (:226-262 on top of legged_robot_base.py:408-489)
        reset = torch.zeros(N, dtype=torch.bool)
        by = {}
        T, ts = ec.termination, ec.termination_scales
        if T.terminate_by_gravity:
            by["gravity"] = torch.norm(self.projected_gravity[:, 0:2], dim=-1) > ts.termination_gravity
            reset |= by["gravity"]
        if T.terminate_when_motion_far:
            by["motion_far"] = torch.any(torch.norm(self.dif_pos, dim=-1) > self.motion_far_thr, dim=-1)
            reset |= by["motion_far"]
        if T.get("terminate_by_ref_pos_z", False):
            by["ref_pos_z"] = torch.abs(self.dif_anchor_pos_z) > ts.get("terminate_by_ref_pos_z_threshold", 0.25)
            reset |= by["ref_pos_z"]
        if T.get("terminate_by_ref_ori", False):
            by["ref_ori"] = self.dif_anchor_ori.abs() > ts.get("terminate_by_ref_ori_threshold", 0.8)
            reset |= by["ref_ori"]
        if T.get("terminate_by_body_z", False):
            by["body_z"] = torch.any(self.dif_local_body_pos[:, [4, 10, 24, 25, 26], -1].abs() > ts.get("terminate_by_body_z_threshold", 0.25), dim=-1)
            reset |= by["body_z"]
        tout = s["episode_length_buf"] > self.max_episode_length
        by["time_out"] = tout
        if T.terminate_when_motion_end:
            by["motion_end"] = (s["episode_length_buf"] * self.dt + s["motion_start_times"]) > s["motion_len"]
            tout = tout | by["motion_end"]
            by["time_out"] = tout
        reset = reset | tout
        s["reset_buf"] = reset.long()
        s["time_out_buf"] = tout
        rmean = s["reset_buf"].float().mean()
        for k, v in by.items():
            log["terminate_by_" + k] = v.float().mean() / (rmean + 1e-15)
        # ---- reward (legged_robot_base.py:715-761)
        vec = bool(ec.use_vec_reward)
        rew_buf = torch.zeros(N, self.R) if vec else torch.zeros(N)
        for i, name in enumerate(self.reward_names):
            rew = getattr(self, "_reward_" + name)() * self.reward_scales[name]
            if name in self.penalty_names and cfg.rewards.reward_penalty_curriculum:
                rew = rew * self.penalty_scale
            if vec:
                rew_buf[:, i] += rew
            else:
                rew_buf = rew_buf + rew
            self.sums[name] = self.sums[name] + rew
        if cfg.rewards.only_positive_rewards:
            rew_buf = torch.clip(rew_buf, min=0.0)
        if "termination" in self.reward_scales:
            rew = (s["reset_buf"] * ~s["time_out_buf"]) * self.reward_scales["termination"]
            if vec:
                rew_buf[:, i] += rew
            else:
                rew_buf = rew_buf + rew
            self.sums["termination"] = self.sums["termination"] + rew
        self.rew_buf = rew_buf
        env_ids = s["reset_buf"].nonzero(as_tuple=False).flatten()
        if len(env_ids) > 0:
            self._reset(env_ids, reset_samples)
        obs = self._observations()
        s["last_actions"] = s["actions"].clone()
        s["last_dof_pos"] = s["dof_pos"].clone()
        s["last_dof_vel"] = s["dof_vel"].clone()
        s["last_contacts"] = s["contacts"].clone()
        s["last_contacts_filt"] = s["contacts_filt"].clone()
        extras = dict(time_outs=s["time_out_buf"], ref_body_pos_extend=ref_pos, ref_body_rot_extend=ref_rot, to_log=log)
        return obs, rew_buf, s["reset_buf"], extras

    # ------------------------------------------------------------------------------------
    def _reset(self, ids, samp):
        s = self.s
        # everything up to the state write-back is shared with v1; the reference lookups differ (:411-483)
        root_before = s["root_states"].clone()
        super()._reset(ids, samp)
        s["motion_len"][ids] = self.ml.motion_len[self.slot_clip[ids]]
        # _reset_dofs: its own lookup at t = ep_len*dt + start (ep_len = 0 for the reset envs)
        ref0 = self._lookup(self._motion_times(0), self.env_origins)
        s["dof_pos"][ids] = ref0["dof_pos"][ids]
        s["dof_vel"][ids] = ref0["dof_vel"][ids]
        # _reset_root_states: kick_motion_res() at (ep_len+1)*dt + start
        ref1 = self._lookup(self._motion_times(1), self.env_origins)
        n = len(ids)
        ident = torch.zeros(n, 4)
        ident[:, 3] = 1.0
        s["root_states"] = root_before
        s["root_states"][ids, 0:3] = ref1["root_pos"][ids]
        s["root_states"][ids, 3:7] = R.quat_mul(ident, ref1["root_rot"][ids])
        s["root_states"][ids, 7:10] = ref1["root_vel"][ids]
        s["root_states"][ids, 10:13] = ref1["root_ang_vel"][ids]

    # ------------------------------------------------------------------------------------
    def _get(self, key):
        s = self.s
        if key in self.cfg.obs.obs_auxiliary:                  # _get_obs_history (legged_robot_base.py:1149-1159)
            aux = self.cfg.obs.obs_auxiliary[key]
            return torch.cat([self.hist[k][:, : aux[k]].reshape(self.N, -1) for k in sorted(aux.keys())], dim=1)
        if key.startswith("future_motion_") or key == "next_step_ref_motion":
            return self.fut[key]
        if key == "dof_vel" and self.cfg.obs.get("masked_dof_vel", False):
            v = s["dof_vel"].clone()
            v[:, [4, 5, 10, 11]] = 0.0
            return v
        table = dict(
            roll_pitch=lambda: self.obs_roll_pitch, root_height=lambda: s["root_states"][:, 2:3],       # a VIEW of the root state in the reference (:608): reset envs show the reset height
             contact_mask=lambda: s["contacts_filt"],
            dr_base_mass=lambda: self.sim_dr["base_mass_scale"],
            local_key_body_pos=lambda: self.obs_local_body_pos[:, self.key_id].reshape(self.N, -1),
            local_key_body_rot=lambda: self.obs_local_body_rot[:, self.key_id].reshape(self.N, -1),
            anchor_ref_pos=lambda: self.obs_anchor_ref_pos.reshape(self.N, -1), anchor_ref_rot=lambda: self.obs_anchor_ref_rot.reshape(self.N, -1),
            dif_local_key_body_pos=lambda: self.obs_dif_local_key_body_pos, local_ref_key_body_pos=lambda: self.obs_local_ref_key_body_pos,
            dif_joint_angles=lambda: self.dif_joint_angles, dif_joint_velocities=lambda: self.dif_joint_vel,
            dif_root_velocity=lambda: self.dif_root_velocity, dif_root_rot=lambda: self.dif_root_rot, dif_root_height=lambda: self.dif_root_height,
        )
        if key in table:
            return table[key]()
        return super()._get(key)

    # ---- v2 reward terms (general_tracking.py:1109-1279) --------------------------------------
    def _angle(self, dq):
        return R.quat_to_angle_axis(dq)[0]

    def _reward_teleop_key_body_position(self):
        return self._exp((self.dif_pos[:, self.key_id] ** 2).mean(dim=-1).mean(dim=-1), "teleop_key_body_pos")

    def _reward_teleop_anchor_body_position(self):
        return self._exp((self.dif_anchor_body_pos ** 2).mean(dim=-1), "teleop_anchor_body_pos")

    def _reward_teleop_anchor_body_rotation(self):
        return self._exp(self._angle(self.dif_rot)[:, self.anchor] ** 2, "teleop_anchor_body_rot")

    def _reward_local_key_body_position(self):
        return self._exp((self.dif_local_body_pos[:, self.key_id] ** 2).mean(dim=-1).mean(dim=-1), "local_key_body_pos")

    def _reward_local_key_body_rotation(self):
        return self._exp((self._angle(self.dif_local_body_rot)[:, self.key_id] ** 2).mean(dim=-1), "local_key_body_rot")

    def _reward_teleop_body_rotation_extend(self):
        return self._exp((self._angle(self.dif_rot) ** 2).mean(dim=-1), "teleop_body_rot")

    def _reward_key_body_velocity(self):
        return self._exp((self.dif_vel[:, self.key_id] ** 2).mean(dim=-1).mean(dim=-1), "key_body_vel")

    def _reward_key_body_ang_velocity(self):
        return self._exp((self.dif_ang[:, self.key_id] ** 2).mean(dim=-1).mean(dim=-1), "key_body_ang_vel")

    def _reward_teleop_root_vel(self):
        return self._exp((self.dif_root_velocity ** 2).mean(dim=-1), "teleop_root_vel")

    def _reward_teleop_root_pose(self):
        return self._exp(self._angle(self.dif_root_rot) ** 2 + (self.dif_root_height ** 2).mean(dim=-1), "teleop_root_pose")

    def _reward_foot_slip_penalty(self):
        is_contact = torch.norm(self.s["contact_forces"][:, self.feet, :], dim=-1) > 1.0
        v = torch.linalg.norm(self.body_x[2][:, self.feet, :2], dim=-1)
        return torch.sum(is_contact * v, dim=1)

    def _sigma_update(self, err, key):
        if not self.adaptive:
            return
        a = self.cfg.rewards.adaptive_tracking_sigma
        alpha = a.alpha
        self.ema[key] = self.ema[key] * (1 - alpha) + err.mean().item() * alpha
        typ = a.get("type", "origin")
        if typ == "scale":
            self.sigma[key] = min(self.ema[key] * a.get("scale", 1.0), self.sigma[key])
        elif typ == "mean":
            self.sigma[key] = self.ema[key]                      # general_tracking.py:988-989 (differs from v1's "mean")
        else:
            self.sigma[key] = min(self.ema[key], self.sigma[key])
