"""LeggedRobotMotionTracking.step restated on explicit state tensors (torch CPU fp32).
TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows, in the reference's own order (legged_robot_base.py:239-338):
  _pre_physics_step      motion_tracking.py:749-768
  _compute_torques       legged_robot_base.py:795-838           (4x per step; state replaced by the
                                                                 replay frame at the last substep)
  _pre_compute_observations_callback   legged_robot_base.py:346-380, motion_tracking.py:583-747
  _check_termination     legged_robot_base.py:408-489, motion_tracking.py:330-357
  _compute_reward        legged_robot_base.py:715-761 + the _reward_* terms
  reset_envs_idx         legged_robot_base.py:491-517,599-686, motion_tracking.py:265-317,369-378,445-543
  _compute_observations  legged_robot_base.py:763-793, helpers.py:128-152, history_handler.py:10-48
Random draws of the reference (reset sampling, torque noise, obs noise) are INPUTS here so that a
trace of the reference can be replayed exactly.
"""
import math

import numpy as np
import torch

from . import rotations as R

TRACK_SIGMA_KEYS = [
    "teleop_max_joint_pos", "teleop_upper_body_pos", "teleop_lower_body_pos", "teleop_vr_3point_pos",
    "teleop_feet_pos", "teleop_body_rot", "teleop_body_vel", "teleop_body_ang_vel", "teleop_joint_pos", "teleop_joint_vel",
]


def _f(x):
    return torch.tensor(x, dtype=torch.float32)


class MotionTrackingOracle:
    def __init__(self, cfg, skel, motion_lib, num_envs, sim_dr):
        """cfg: resolved top-level config tree; sim_dr: dict(base_com_bias[N,3], link_mass_scale[N,22],
        friction_coeffs[N,1,1]) — the simulator-owned domain-randomisation observables."""
        self.cfg = cfg
        ec = cfg.env.config
        self.ec = ec
        self.skel = skel
        self.ml = motion_lib
        N = self.N = num_envs
        rc = cfg.robot
        self.D = D = len(rc.dof_names)
        self.B = B = skel["num_bodies"]
        self.body_names = list(skel["body_names_ext"][:B])
        self.body_list = list(skel["body_names_ext"])
        self.Bx = len(self.body_list)
        sim = cfg.simulator.config.sim
        self.sim_dt = 1.0 / sim.fps
        self.decimation = sim.control_decimation
        self.dt = self.decimation * self.sim_dt                                     # base_task.py:37
        self.max_episode_length = np.ceil(ec.max_episode_length_s / self.dt)        # base_task.py:39
        # --- body index sets (base_task.py:169-205, motion_tracking.py:203-232) ---
        find = lambda n: self.body_names.index(n)
        self.feet = [find(s) for s in self.body_names if rc.foot_name in s]
        pen = []
        for name in rc.penalize_contacts_on:
            pen.extend([s for s in self.body_names if name in s])
        self.penalised = [find(s) for s in pen]
        tcon = []
        for n in rc.get("terminate_after_contacts_on", []):                     # base_task.py:178-180
            tcon.extend([s_ for s_ in self.body_names if n in s_])
        self.termination_contact = [find(s_) for s_ in tcon]
        m = rc.motion
        self.ext_parent = [self.body_list.index(e["parent_name"]) for e in m.extend_config]
        self.ext_pos = _f([list(e["pos"]) for e in m.extend_config])
        ext_rot_wxyz = _f([list(e["rot"]) for e in m.extend_config])
        self.ext_rot = ext_rot_wxyz[:, [1, 2, 3, 0]]
        self.track_id = [self.body_list.index(l) for l in m.motion_tracking_link]
        self.lower_id = [self.body_list.index(l) for l in m.lower_body_link]
        self.upper_id = [self.body_list.index(l) for l in m.upper_body_link]
        # --- gains / limits (legged_robot_base.py:74-108, isaacgym.py:_process_dof_props) ---
        asc = rc.control.action_scale                      # scalar, or per-joint-group dict (legged_robot_base.py:99-100,805-808)
        self.action_scale = asc if isinstance(asc, (int, float)) else torch.ones(D)
        self.default_dof_pos = torch.zeros(D)
        self.p_gains = torch.zeros(D)
        self.d_gains = torch.zeros(D)
        for i, name in enumerate(rc.dof_names):
            self.default_dof_pos[i] = rc.init_state.default_joint_angles[name]
            for k in rc.control.stiffness.keys():
                if k in name:
                    self.p_gains[i] = rc.control.stiffness[k]
                    self.d_gains[i] = rc.control.damping[k]
                    if not isinstance(asc, (int, float)):
                        self.action_scale[i] = asc[k]
        lo = _f(list(rc.dof_pos_lower_limit_list))
        hi = _f(list(rc.dof_pos_upper_limit_list))
        self.hard_limits = torch.stack([lo, hi], -1)
        self.dof_vel_limits = _f(list(rc.dof_vel_limit_list))
        self.torque_limits = _f(list(rc.dof_effort_limit_list))
        self.soft_limits = torch.zeros(D, 2)
        for i in range(D):
            mid = (self.hard_limits[i, 0] + self.hard_limits[i, 1]) / 2
            r = self.hard_limits[i, 1] - self.hard_limits[i, 0]
            self.soft_limits[i, 0] = mid - 0.5 * r * cfg.rewards.reward_limit.soft_dof_pos_limit
            self.soft_limits[i, 1] = mid + 0.5 * r * cfg.rewards.reward_limit.soft_dof_pos_limit
        # --- rewards (legged_robot_base.py:167-233) ---
        rw = cfg.rewards
        self.reward_scales = {}
        for k, v in rw.reward_scales.items():
            if v != 0:
                self.reward_scales[k] = v * self.dt
        self.reward_names = [k for k in self.reward_scales if k != "termination"]
        self.R = len(self.reward_names) + 1 if ec.use_vec_reward else 1
        self.penalty_names = set(rw.reward_penalty_reward_names)
        # --- observations (helpers.py:47-80, history_handler.py:10-31) ---
        ob = cfg.obs
        self.obs_dims = {k: v for d in ob.obs_dims for k, v in d.items()} if isinstance(ob.obs_dims, list) else dict(ob.obs_dims)
        self.hist_len = {}
        for aux in ob.obs_auxiliary.values():
            for k, n in aux.items():
                self.hist_len[k] = max(self.hist_len.get(k, 0), n)
        # --- globals the reference keeps on the host ---
        self.sigma = {k: float(v) for k, v in rw.reward_tracking_sigma.items()}
        self.adaptive = bool(rw.get("adaptive_tracking_sigma", {}).get("enable", False))
        self.ema = dict(self.sigma)
        self.penalty_scale = rw.reward_initial_penalty_scale
        self.noise_curriculum = bool(ob.get("add_noise_currculum", False))          # legged_robot_base.py:119-124
        self.noise_cur = float(ob.noise_initial_value) if self.noise_curriculum else 1.0
        self.avg_ep_len = 0.0
        tc = ec.termination_curriculum
        self.motion_far_thr = tc.terminate_when_motion_far_initial_threshold if (ec.termination.terminate_when_motion_far and tc.terminate_when_motion_far_curriculum) else ec.termination_scales.termination_motion_far_threshold
        lc = rw.reward_limit.reward_limits_curriculum
        self.soft_pos_val = lc.soft_dof_pos_initial_limit
        self.soft_vel_val = lc.soft_dof_vel_initial_limit
        self.soft_tau_val = lc.soft_torque_initial_limit
        self.ref_init_yaw = 0.0
        self.sim_dr = sim_dr
        self.log = {}
        # --- per-env state ---
        z = lambda *s: torch.zeros(*s)
        self.s = dict(
            root_states=z(N, 13), dof_pos=z(N, D), dof_vel=z(N, D), contact_forces=z(N, B, 3),
            actions=z(N, D), last_actions=z(N, D), actions_after_delay=z(N, D),
            action_queue=z(N, cfg.domain_rand.ctrl_delay_step_range[1] + 1, D), action_delay_idx=torch.zeros(N, dtype=torch.long),
            last_dof_pos=z(N, D), last_dof_vel=z(N, D), torques=z(N, D), feet_air_time=z(N, 2),
            contacts=z(N, 2), contacts_filt=z(N, 2), last_contacts=z(N, 2), last_contacts_filt=z(N, 2),
            kp_scale=torch.ones(N, D), kd_scale=torch.ones(N, D), rfi_lim_scale=torch.ones(N, D), rao_scale=torch.ones(N, D),
            episode_length_buf=torch.zeros(N, dtype=torch.long), last_episode_length_buf=torch.zeros(N, dtype=torch.long),
            motion_start_times=z(N), motion_len=z(N), end_time_ratio_buf=z(N),
            reset_buf=torch.ones(N, dtype=torch.long), time_out_buf=torch.zeros(N, dtype=torch.bool),
        )
        self.sums = {k: z(N) for k in self.reward_scales}
        self.hist = {k: z(N, n, self.obs_dims[k]) for k, n in self.hist_len.items()}
        self.motion_ids = torch.zeros(N, dtype=torch.long)       # every slot plays clip 0 (single-motion lib)
        self.env_origins = z(N, 3)
        self.gravity_vec = _f([0.0, 0.0, -1.0]).repeat(N, 1)

    # ------------------------------------------------------------------------------------
    def load_state(self, st):
        """st: dict name -> numpy (names as in oracle/ref_harness/gen_env_golden.snapshot)."""
        for k in self.s:
            v = torch.from_numpy(np.asarray(st[k]))
            self.s[k] = v.to(self.s[k].dtype).clone()
        for k in self.sums:
            self.sums[k] = torch.from_numpy(st["sum__" + k]).float().clone()
        for k in self.hist:
            self.hist[k] = torch.from_numpy(st["hist__" + k]).float().clone()
        for k in self.sigma:
            self.sigma[k] = float(st["sigma__" + k])
            if "ema__" + k in st:
                self.ema[k] = float(st["ema__" + k])
        self.penalty_scale = float(st["reward_penalty_scale"])
        if "noise_curriculum_value" in st:
            self.noise_cur = float(st["noise_curriculum_value"])
        self.avg_ep_len = torch.tensor(float(st["average_episode_length"]), dtype=torch.float32)
        self.motion_far_thr = float(st["motion_far_threshold"])

    # ------------------------------------------------------------------------------------
    def compute_torques(self, actions, u_rfi):
        s = self.s
        dr = self.cfg.domain_rand
        a = actions * self.action_scale
        ct = self.cfg.robot.control.control_type                      # legged_robot_base.py:809-817
        if ct == "P":
            tq = s["kp_scale"] * self.p_gains * (a + self._default(s) - s["dof_pos"]) - s["kd_scale"] * self.d_gains * s["dof_vel"]
        elif ct == "V":
            tq = s["kp_scale"] * self.p_gains * (a - s["dof_vel"]) - s["kd_scale"] * self.d_gains * (s["dof_vel"] - s["last_dof_vel"]) / self.sim_dt
        elif ct == "T":
            tq = a
        else:
            raise NameError(f"Unknown controller type: {ct}")
        if dr.randomize_torque_rfi:
            tq = tq + (u_rfi * 2.0 - 1.0) * dr.rfi_lim * s["rfi_lim_scale"] * self.torque_limits
        if dr.use_rao:
            tq = tq + s["rao_scale"] * self.torque_limits
        if self.cfg.robot.control.clip_torques:
            tq = torch.clip(tq, -self.torque_limits, self.torque_limits)
        return tq

    def _motion_times(self, offset_steps=1):
        s = self.s
        return (s["episode_length_buf"] + offset_steps) * self.dt + s["motion_start_times"]

    def _sigma_update(self, err, key):
        if not self.adaptive:
            return
        alpha = self.cfg.rewards.adaptive_tracking_sigma.alpha
        self.ema[key] = self.ema[key] * (1 - alpha) + err.mean().item() * alpha
        self.sigma[key] = min(self.ema[key], self.sigma[key])        # type "origin" (motion_tracking.py:1046-1048)

    # ------------------------------------------------------------------------------------
    def step(self, actions, frame, body_state, u_rfi=None, reset_samples=None, gate_u=None, redraw_samples=None):
        """actions [N,D]; frame: dict(root[N,13], dof_pos, dof_vel, contact[N,B,3]) = the replay
        frame the sim switches to at the end of the physics step; body_state: (pos,rot,vel,ang)
        of the B bodies for that frame (oracle.fk.sim_fk of the frame); reset_samples: dict of
        [N,...] tensors (motion_start_times, kp_scale, kd_scale, rfi_lim_scale, rao_scale,
        action_delay_idx) holding the values the reference drew, consumed for resetting envs; redraw_samples: the same kind of dict when
        domain_rand.reinit_epis_rand fires in this step — `_update_tasks_callback` re-draws the episodic DR of EVERY env
        (legged_robot_base.py:390-395) between the pre-computed observations and the termination check."""
        s, N, D = self.s, self.N, self.D
        ec, cfg = self.ec, self.cfg
        log = self.log
        # ---- _pre_physics_step
        clipv = cfg.robot.control.action_clip_value
        s["actions"] = torch.clip(actions, -clipv, clipv)
        log["action_clip_frac"] = (s["actions"].abs() == clipv).sum() / s["actions"].numel()
        if cfg.domain_rand.randomize_ctrl_delay:
            s["action_queue"][:, 1:] = s["action_queue"][:, :-1].clone()
            s["action_queue"][:, 0] = s["actions"]
            s["actions_after_delay"] = s["action_queue"][torch.arange(N), s["action_delay_idx"]].clone()
        else:
            s["actions_after_delay"] = s["actions"].clone()
        # ---- _physics_step: torques from the pre-step state, then the replay frame lands
        if u_rfi is None:
            u_rfi = torch.full((N, D), 0.5)
        s["torques"] = self.compute_torques(s["actions_after_delay"], u_rfi)
        s["root_states"] = frame["root"].clone()
        s["dof_pos"] = frame["dof_pos"].clone()
        s["dof_vel"] = frame["dof_vel"].clone()
        s["contact_forces"] = frame["contact"].clone()
        bpos, brot, bvel, bang = body_state
        # ---- _post_physics_step
        s["episode_length_buf"] = s["episode_length_buf"] + 1
        s["last_episode_length_buf"] = s["episode_length_buf"].clone()
        base_quat = s["root_states"][:, 3:7]
        self.rpy = R.get_euler_xyz(base_quat)
        self.base_lin_vel = R.quat_rotate_inverse(base_quat, s["root_states"][:, 7:10])
        self.base_ang_vel = R.quat_rotate_inverse(base_quat, s["root_states"][:, 10:13])
        self.projected_gravity = R.quat_rotate_inverse(base_quat, self.gravity_vec)
        feet_f = s["contact_forces"][:, self.feet, :]
        s["contacts"] = (feet_f.norm(dim=-1) > 1.0).float()
        s["contacts_filt"] = torch.logical_or(s["contacts"], s["last_contacts"]).float()
        # tracking part
        motion_times = self._motion_times(1)
        ref = self.ml.get_motion_state(self.motion_ids, motion_times, offset=self.env_origins)
        if self.ml.has_contact_mask:
            self.ref_contact_mask = ref["contact_mask"]
        ref_pos, ref_rot, ref_vel, ref_ang = ref["rg_pos_t"], ref["rg_rot_t"], ref["body_vel_t"], ref["body_ang_vel_t"]
        E = len(self.ext_parent)
        par_rot = brot[:, self.ext_parent]
        ext_off = self.ext_pos.repeat(N, 1, 1)
        rotated = R.quat_rotate(par_rot.reshape(-1, 4), ext_off.reshape(-1, 3))
        ext_pos = R.quat_rotate(self.ext_rot.repeat(N, 1, 1).reshape(-1, 4), rotated).view(N, E, 3) + bpos[:, self.ext_parent]
        pos_x = torch.cat([bpos, ext_pos], dim=1)
        ext_rot = R.quat_mul(par_rot.reshape(-1, 4), self.ext_rot.repeat(N, 1, 1).reshape(-1, 4)).view(N, E, 4)
        rot_x = torch.cat([brot, ext_rot], dim=1)
        ang_x = torch.cat([bang, bang[:, self.ext_parent]], dim=1)
        ext_vel = bvel[:, self.ext_parent] + torch.cross(bang[:, self.ext_parent], ext_off, dim=2)   # un-rotated offset, sic (:641)
        vel_x = torch.cat([bvel, ext_vel], dim=1)
        self.body_x = (pos_x, rot_x, vel_x, ang_x)
        self.ref_body_pos_extend = ref_pos
        self.ref_body_rot_extend = ref_rot
        self.dif_pos = ref_pos - pos_x
        self.dif_rot = ref_rot - rot_x                    # elementwise quaternion difference, sic (:651)
        self.dif_vel = ref_vel - vel_x
        self.dif_ang = ref_ang - ang_x
        self.dif_joint_angles = ref["dof_pos"] - s["dof_pos"]
        self.dif_joint_vel = ref["dof_vel"] - s["dof_vel"]
        root_q = s["root_states"][:, 3:7]
        heading_inv = R.calc_heading_quat_inv(root_q)
        hexp = heading_inv.unsqueeze(1).expand(-1, self.Bx, -1).reshape(-1, 4)
        self.relyaw = self.rpy[:, 2:3] - self.ref_init_yaw
        self.obs_dif_local_rigid_body_pos = R.quat_rotate(hexp, (ref_pos - pos_x).reshape(-1, 3)).view(N, -1)
        glob_ref = ref_pos - s["root_states"][:, :3].view(N, 1, 3)
        self.obs_local_ref_rigid_body_pos = R.quat_rotate(hexp, glob_ref.reshape(-1, 3)).view(N, -1)
        vr = ref_pos[:, self.track_id, :] - s["root_states"][:, 0:3].view(N, 1, 3)
        self.obs_vr_3point_pos = R.quat_rotate(heading_inv.repeat(3, 1), vr.reshape(-1, 3)).view(N, -1)
        self.ref_motion_phase = (motion_times / self.ml.get_motion_length(self.motion_ids)).unsqueeze(1)
        log["upper_body_diff_norm"] = self.dif_pos[:, self.upper_id].norm(dim=-1).mean()
        log["lower_body_diff_norm"] = self.dif_pos[:, self.lower_id].norm(dim=-1).mean()
        log["vr_3point_diff_norm"] = self.dif_pos[:, self.track_id].norm(dim=-1).mean()
        log["joint_pos_diff_norm"] = self.dif_joint_angles.norm(dim=-1).mean()
        # ---- _update_tasks_callback: reinit_epis_rand (legged_robot_base.py:390-395)
        if redraw_samples is not None:
            self._episodic_dr(torch.arange(N), redraw_samples)
        # ---- _check_termination
        reset = torch.zeros(N, dtype=torch.bool)
        by = {}
        T = ec.termination
        if T.get("terminate_by_contact", False):              # legged_robot_base.py:434-436
            by["contact"] = torch.any(torch.norm(frame["contact"][:, self.termination_contact, :], dim=-1) > 1.0, dim=1)
            reset |= by["contact"]
        if T.terminate_by_gravity:
            by["gravity"] = torch.norm(self.projected_gravity[:, 0:2], dim=-1) > ec.termination_scales.termination_gravity
            reset |= by["gravity"]
        if T.get("terminate_by_low_height", False):           # :442-444
            by["low_height"] = frame["root"][:, 2] < ec.termination_scales.termination_min_base_height
            reset |= by["low_height"]
        if T.terminate_when_motion_far:
            by["motion_far"] = torch.any(torch.norm(self.dif_pos, dim=-1) > self.motion_far_thr, dim=-1)
            reset |= by["motion_far"]
            log["terminate_when_motion_far_threshold"] = torch.tensor(self.motion_far_thr, dtype=torch.float)
        # probabilistic terminations near the joint limits (legged_robot_base.py:449-479): one uniform per gate and step (`gate_u`), the same for
        # all envs; position limits m -/+ 0.5 r scale from the hard limits (isaacgym.py:380-388)
        TS, TP = ec.termination_scales, ec.get("termination_probality", {})
        gu = [0.0, 0.0, 0.0] if gate_u is None else [float(v) for v in gate_u]
        if T.get("terminate_when_close_to_dof_pos_limit", False):
            lo = torch.tensor([float(v) for v in self.cfg.robot.dof_pos_lower_limit_list]); hi = torch.tensor([float(v) for v in self.cfg.robot.dof_pos_upper_limit_list])
            m_, r_ = (lo + hi) / 2, hi - lo
            lo_t, hi_t = m_ - 0.5 * r_ * TS.termination_close_to_dof_pos_limit, m_ + 0.5 * r_ * TS.termination_close_to_dof_pos_limit
            out = (-(frame["dof_pos"] - lo_t).clip(max=0.0) + (frame["dof_pos"] - hi_t).clip(min=0.0)).sum(dim=1)
            by["dof_pos_limit"] = (out > 0.0) if gu[0] < TP.terminate_when_close_to_dof_pos_limit else torch.zeros(N, dtype=torch.bool)
            reset |= by["dof_pos_limit"]
        if T.get("terminate_when_close_to_dof_vel_limit", False):
            out = (frame["dof_vel"].abs() - self.dof_vel_limits * TS.termination_close_to_dof_vel_limit).clip(min=0.0, max=1.0).sum(dim=1)
            by["dof_vel_limit"] = (out > 0.0) if gu[1] < TP.terminate_when_close_to_dof_vel_limit else torch.zeros(N, dtype=torch.bool)
            reset |= by["dof_vel_limit"]
        if T.get("terminate_when_close_to_torque_limit", False):
            out = (s["torques"].abs() - self.torque_limits * TS.termination_close_to_torque_limit).clip(min=0.0, max=1.0).sum(dim=1)
            by["torque_limit"] = (out > 0.0) if gu[2] < TP.terminate_when_close_to_torque_limit else torch.zeros(N, dtype=torch.bool)
            reset |= by["torque_limit"]
        tout = s["episode_length_buf"] > self.max_episode_length
        by["time_out"] = tout
        if T.terminate_when_motion_end:
            cur_t = s["episode_length_buf"] * self.dt + s["motion_start_times"]
            by["motion_end"] = cur_t > s["motion_len"]
            tout = tout | by["motion_end"]
            by["time_out"] = tout
        reset = reset | tout
        s["reset_buf"] = reset.long()
        s["time_out_buf"] = tout
        rmean = s["reset_buf"].float().mean()
        for k, v in by.items():
            log["terminate_by_" + k] = v.float().mean() / (rmean + 1e-15)
        # ---- _compute_reward
        rew_buf = torch.zeros(N, self.R)
        for i, name in enumerate(self.reward_names):
            rew = getattr(self, "_reward_" + name)() * self.reward_scales[name]
            if name in self.penalty_names and cfg.rewards.reward_penalty_curriculum:
                rew = rew * self.penalty_scale
            rew_buf[:, i] += rew
            self.sums[name] = self.sums[name] + rew
        if cfg.rewards.only_positive_rewards:
            rew_buf = torch.clip(rew_buf, min=0.0)
        if "termination" in self.reward_scales:
            rew = (s["reset_buf"] * ~s["time_out_buf"]) * self.reward_scales["termination"]
            rew_buf[:, i] += rew                     # column of the last loop term, sic (:743-744)
            self.sums["termination"] = self.sums["termination"] + rew
        log["penalty_scale"] = torch.tensor(self.penalty_scale, dtype=torch.float)
        log["average_episode_length"] = torch.as_tensor(self.avg_ep_len, dtype=torch.float)
        self.rew_buf = rew_buf
        # ---- reset_envs_idx
        env_ids = s["reset_buf"].nonzero(as_tuple=False).flatten()
        if len(env_ids) > 0:
            self._reset(env_ids, reset_samples)
        # ---- _compute_observations
        obs = self._observations()
        # ---- _post_compute_observations_callback
        s["last_actions"] = s["actions"].clone()
        s["last_dof_pos"] = s["dof_pos"].clone()
        s["last_dof_vel"] = s["dof_vel"].clone()
        s["last_contacts"] = s["contacts"].clone()
        s["last_contacts_filt"] = s["contacts_filt"].clone()
        extras = dict(time_outs=s["time_out_buf"], ref_body_pos_extend=self.ref_body_pos_extend, ref_body_rot_extend=self.ref_body_rot_extend, to_log=log)
        return obs, rew_buf, s["reset_buf"], extras

    # ------------------------------------------------------------------------------------
    def _reset(self, ids, samp):
        s, cfg, ec = self.s, self.cfg, self.ec
        n = len(ids)
        # _reset_buffers_callback (legged_robot_base.py:670-686)
        for k in ["actions", "last_actions", "actions_after_delay", "last_dof_pos", "last_dof_vel", "feet_air_time", "contacts", "contacts_filt", "last_contacts", "last_contacts_filt"]:
            s[k][ids] = 0.0
        s["episode_length_buf"][ids] = 0
        s["reset_buf"][ids] = 1
        cur = torch.mean(s["last_episode_length_buf"][ids], dtype=torch.float)
        K = cfg.rewards.num_compute_average_epl
        self.avg_ep_len = self.avg_ep_len * (1 - n / K) + cur * (n / K)
        for k in self.hist:
            self.hist[k][ids] *= 0.0
        # _reset_tasks_callback: episodic DR (legged_robot_base.py:599-635)
        self._episodic_dr(ids, samp)
        rw = cfg.rewards
        if rw.reward_penalty_curriculum:                    # legged_robot_base.py:882-900
            if self.avg_ep_len < rw.reward_penalty_level_down_threshold:
                self.penalty_scale *= 1 - rw.reward_penalty_degree
            elif self.avg_ep_len > rw.reward_penalty_level_up_threshold:
                self.penalty_scale *= 1 + rw.reward_penalty_degree
            self.penalty_scale = float(np.clip(self.penalty_scale, rw.reward_min_penalty_scale, rw.reward_max_penalty_scale))
        lc = rw.reward_limit.reward_limits_curriculum        # legged_robot_base.py:902-939
        for attr, pre in [("soft_pos_val", "soft_dof_pos"), ("soft_vel_val", "soft_dof_vel"), ("soft_tau_val", "soft_torque")]:
            if lc[pre + "_curriculum"]:
                v = getattr(self, attr)
                if self.avg_ep_len < lc[pre + "_curriculum_level_down_threshold"]:
                    v *= 1 + lc[pre + "_curriculum_degree"]
                elif self.avg_ep_len > lc[pre + "_curriculum_level_up_threshold"]:
                    v *= 1 - lc[pre + "_curriculum_degree"]
                setattr(self, attr, float(np.clip(v, lc[pre + "_min_limit"], lc[pre + "_max_limit"])))
        if self.noise_curriculum:                            # _update_obs_noise_curriculum, legged_robot_base.py:1117-1126
            if self.avg_ep_len < cfg.obs.soft_dof_pos_curriculum_level_down_threshold:
                self.noise_cur *= 1 - cfg.obs.soft_dof_pos_curriculum_degree
            elif self.avg_ep_len > rw.reward_penalty_level_up_threshold:
                self.noise_cur *= 1 + cfg.obs.soft_dof_pos_curriculum_degree
            self.noise_cur = float(np.clip(self.noise_cur, cfg.obs.noise_value_min, cfg.obs.noise_value_max))
            self.log["current_noise_curriculum_value"] = torch.tensor(self.noise_cur, dtype=torch.float)
        # tracking part (motion_tracking.py:265-287)
        end_time = s["last_episode_length_buf"][ids] * self.dt + s["motion_start_times"][ids]
        s["end_time_ratio_buf"][ids] = end_time / s["motion_len"][ids]
        self.log["end_time_ratio"] = s["end_time_ratio_buf"].mean()
        self.log["end_time_ratio_std"] = s["end_time_ratio_buf"].std()
        s["motion_len"][ids] = self.ml.get_motion_length(self.motion_ids[ids])
        s["motion_start_times"][ids] = samp["motion_start_times"][ids]
        tc = ec.termination_curriculum
        if ec.termination.terminate_when_motion_far and tc.terminate_when_motion_far_curriculum:
            if self.avg_ep_len < tc.terminate_when_motion_far_curriculum_level_down_threshold:
                self.motion_far_thr *= 1 + tc.terminate_when_motion_far_curriculum_degree
            elif self.avg_ep_len > tc.terminate_when_motion_far_curriculum_level_up_threshold:
                self.motion_far_thr *= 1 - tc.terminate_when_motion_far_curriculum_degree
            self.motion_far_thr = float(np.clip(self.motion_far_thr, tc.terminate_when_motion_far_threshold_min, tc.terminate_when_motion_far_threshold_max))
        # _reset_robot_states_callback: second full lookup with ep_len = 0 (motion_tracking.py:445-543)
        ref = self.ml.get_motion_state(self.motion_ids, self._motion_times(1), offset=self.env_origins)
        s["dof_pos"][ids] = ref["dof_pos"][ids]
        s["dof_vel"][ids] = ref["dof_vel"][ids]
        ident = torch.zeros(n, 4)
        ident[:, 3] = 1.0
        s["root_states"][ids, 0:3] = ref["root_pos"][ids]
        s["root_states"][ids, 3:7] = R.quat_mul(ident, ref["root_rot"][ids])
        s["root_states"][ids, 7:10] = ref["root_vel"][ids]
        s["root_states"][ids, 10:13] = ref["root_ang_vel"][ids]
        self.episode_extras = {"rew_" + k: self.sums[k][ids] / ec.max_episode_length_s for k in self.sums}
        for k in self.sums:
            self.sums[k][ids] = 0.0

    def _episodic_dr(self, ids, samp):
        """_episodic_domain_randomization (legged_robot_base.py:599-635) with the reference's draws handed in"""
        s, cfg = self.s, self.cfg
        dr = cfg.domain_rand
        if dr.randomize_pd_gain:
            s["kp_scale"][ids] = samp["kp_scale"][ids]
            s["kd_scale"][ids] = samp["kd_scale"][ids]
        if dr.randomize_rfi_lim:
            s["rfi_lim_scale"][ids] = samp["rfi_lim_scale"][ids]
        if dr.use_rao:
            s["rao_scale"][ids] = samp["rao_scale"][ids]
        if dr.randomize_ctrl_delay:
            s["action_queue"][ids] *= 0.0
            s["action_delay_idx"][ids] = samp["action_delay_idx"][ids]
        if dr.get("randomize_default_dof_pos", False):          # legged_robot_base.py:632-635: default = raw default + U(dof_pos_range)
            if "default_dof_pos" not in s:
                s["default_dof_pos"] = self.default_dof_pos.repeat(self.N, 1).clone()
            s["default_dof_pos"][ids] = samp["dof_pos_bias"][ids] + self.default_dof_pos

    # ------------------------------------------------------------------------------------
    def _default(self, s):
        """default joint angles: per env once randomize_default_dof_pos has drawn them (legged_robot_base.py:632-635), else the config's"""
        return s["default_dof_pos"] if "default_dof_pos" in s else self.default_dof_pos

    def _get(self, key):
        s = self.s
        if key.startswith("history_"):
            aux = self.cfg.obs.obs_auxiliary[key]
            return torch.cat([self.hist[k][:, : aux[k]].reshape(self.N, -1) for k in sorted(aux.keys())], dim=1)
        table = dict(
            base_lin_vel=lambda: self.base_lin_vel, base_ang_vel=lambda: self.base_ang_vel,
            projected_gravity=lambda: self.projected_gravity, dof_pos=lambda: s["dof_pos"] - self._default(s),
            dof_vel=lambda: s["dof_vel"], actions=lambda: s["actions"], ref_motion_phase=lambda: self.ref_motion_phase,
            dif_local_rigid_body_pos=lambda: self.obs_dif_local_rigid_body_pos,
            local_ref_rigid_body_pos=lambda: self.obs_local_ref_rigid_body_pos, vr_3point_pos=lambda: self.obs_vr_3point_pos,
            dr_base_com=lambda: self.sim_dr["base_com_bias"], dr_link_mass=lambda: self.sim_dr["link_mass_scale"],
            dr_friction=lambda: self.sim_dr["friction_coeffs"].reshape(self.N, -1),
            dr_kp=lambda: s["kp_scale"], dr_kd=lambda: s["kd_scale"],
            dr_ctrl_delay=lambda: s["action_delay_idx"].reshape(self.N, -1).float(),
        )
        return table[key]()

    def _parse(self, keys, noise_u=None):
        ob = self.cfg.obs
        out = {}
        for key in keys:
            k = key[:-4] if key.endswith("_raw") else key
            noise = 0.0 if key.endswith("_raw") else ob.noise_scales[k] * self.noise_cur      # noise_extra_scale, legged_robot_base.py:760-763
            x = self._get(k).clone()
            u = torch.full_like(x, 0.5) if noise_u is None else noise_u[k]
            out[k] = (x + (u * 2.0 - 1.0) * noise) * ob.obs_scales[k]
        return out

    def _observations(self):
        ob = self.cfg.obs
        obs = {}
        for group, keys in ob.obs_dict.items():
            raw = self._parse(keys)
            obs[group] = torch.cat([raw[k[:-4] if k.endswith("_raw") else k] for k in sorted(keys)], dim=-1)
        hist_new = self._parse(list(self.hist.keys()))
        clipv = self.ec.normalization.clip_observations
        obs = {k: torch.clip(v, -clipv, clipv) for k, v in obs.items()}
        for k in self.hist:
            old = self.hist[k].clone()
            self.hist[k][:, 1:] = old[:, :-1]
            self.hist[k][:, 0] = hist_new[k]
        return obs

    # ---- reward terms ------------------------------------------------------------------
    def _exp(self, err, key):
        r = torch.exp(-err / self.sigma[key])
        self._sigma_update(err, key)
        return r

    def _reward_teleop_contact_mask(self):
        return 1 - (self.s["contacts_filt"] - self.ref_contact_mask).abs().mean(dim=-1)

    def _reward_teleop_max_joint_position(self):
        return self._exp(self.dif_joint_angles.abs().max(dim=-1)[0], "teleop_max_joint_pos")

    def _reward_teleop_body_position_extend(self):
        up = (self.dif_pos[:, self.upper_id] ** 2).mean(dim=-1).mean(dim=-1)
        lo = (self.dif_pos[:, self.lower_id] ** 2).mean(dim=-1).mean(dim=-1)
        r_up = torch.exp(-up / self.sigma["teleop_upper_body_pos"])
        r_lo = torch.exp(-lo / self.sigma["teleop_lower_body_pos"])
        r = r_lo * self.cfg.rewards.teleop_body_pos_lowerbody_weight + r_up * self.cfg.rewards.teleop_body_pos_upperbody_weight
        self._sigma_update(up, "teleop_upper_body_pos")
        self._sigma_update(lo, "teleop_lower_body_pos")
        return r

    def _reward_teleop_vr_3point(self):
        return self._exp((self.dif_pos[:, self.track_id] ** 2).mean(dim=-1).mean(dim=-1), "teleop_vr_3point_pos")

    def _reward_teleop_body_position_feet(self):
        return self._exp((self.dif_pos[:, self.feet] ** 2).mean(dim=-1).mean(dim=-1), "teleop_feet_pos")

    def _reward_teleop_body_rotation_extend(self):
        return self._exp((self.dif_rot ** 2).mean(dim=-1).mean(dim=-1), "teleop_body_rot")

    def _reward_teleop_body_velocity_extend(self):
        return self._exp((self.dif_vel ** 2).mean(dim=-1).mean(dim=-1), "teleop_body_vel")

    def _reward_teleop_body_ang_velocity_extend(self):
        return self._exp((self.dif_ang ** 2).mean(dim=-1).mean(dim=-1), "teleop_body_ang_vel")

    def _reward_teleop_joint_position(self):
        return self._exp((self.dif_joint_angles ** 2).mean(dim=-1), "teleop_joint_pos")

    def _reward_teleop_joint_velocity(self):
        return self._exp((self.dif_joint_vel ** 2).mean(dim=-1), "teleop_joint_vel")

    def _reward_penalty_torques(self):
        return torch.sum(torch.square(self.s["torques"]), dim=1)

    def _reward_penalty_dof_vel(self):
        return torch.sum(torch.square(self.s["dof_vel"]), dim=1)

    def _reward_penalty_dof_acc(self):
        return torch.sum(torch.square((self.s["last_dof_vel"] - self.s["dof_vel"]) / self.dt), dim=1)

    def _reward_penalty_action_rate(self):
        return torch.sum(torch.square(self.s["last_actions"] - self.s["actions"]), dim=1)

    def _reward_feet_air_time(self):
        s = self.s
        contact = s["contact_forces"][:, self.feet, 2] > 1.0
        contact_filt = torch.logical_or(contact, s["last_contacts"])
        first = (s["feet_air_time"] > 0.0) * contact_filt
        s["feet_air_time"] = s["feet_air_time"] + self.dt
        r = torch.sum((s["feet_air_time"] - self.cfg.rewards.desired_feet_air_time) * first, dim=1)
        s["feet_air_time"] = s["feet_air_time"] * ~contact_filt
        return r

    def _reward_penalty_feet_contact_forces(self):
        f = torch.norm(self.s["contact_forces"][:, self.feet, :], dim=-1)
        return torch.sum((f - self.cfg.rewards.locomotion_max_contact_force).clip(min=0.0), dim=1)

    def _reward_penalty_stumble(self):
        cf = self.s["contact_forces"][:, self.feet]
        return torch.any(torch.norm(cf[..., :2], dim=2) > 5 * torch.abs(cf[..., 2]), dim=1)

    # ---- foot orientation terms (legged_robot_base.py:1030-1079; quat_apply / quat_rotate_inverse: utils/torch_utils.py, wrap_to_pi rotations.py:50-53)
    def _foot_heading_diffs(self):
        fwd = torch.tensor([1.0, 0.0, 0.0]).repeat(self.N, 1)
        rot = self.body_x[1]
        root_f = R.quat_apply(self.s["root_states"][:, 3:7], fwd)
        h_root = torch.atan2(root_f[:, 1], root_f[:, 0])
        out = []
        for f in self.feet:
            ff = R.quat_apply(rot[:, f], fwd)
            a = torch.atan2(ff[:, 1], ff[:, 0]) - h_root
            a = a % (2 * np.pi)
            a = a - 2 * np.pi * (a > np.pi)
            out.append(torch.abs(a))
        return out

    def _foot_tilts(self):
        rot = self.body_x[1]
        return [torch.sum(torch.square(R.quat_rotate_inverse(rot[:, f], self.gravity_vec)[:, :2]), dim=1) ** 0.5 for f in self.feet]

    def _reward_feet_heading_alignment(self):
        l, r = self._foot_heading_diffs()
        return l + r

    def _reward_feet_heading_alignment_contact(self):
        l, r = self._foot_heading_diffs()
        return l * self.s["contacts_filt"][:, 0] + r * self.s["contacts_filt"][:, 1]

    def _reward_penalty_feet_ori(self):
        l, r = self._foot_tilts()
        return l + r

    def _reward_penalty_feet_ori_contact(self):
        l, r = self._foot_tilts()
        return l * self.s["contacts_filt"][:, 0] + r * self.s["contacts_filt"][:, 1]

    def _reward_penalty_slippage(self):
        foot_vel = self.body_x[2][:, self.feet]
        return torch.sum(torch.norm(foot_vel, dim=-1) * (torch.norm(self.s["contact_forces"][:, self.feet, :], dim=-1) > 1.0), dim=1)

    def _reward_limits_dof_pos(self):
        q = self.s["dof_pos"]
        if self.cfg.rewards.reward_limit.reward_limits_curriculum.soft_dof_pos_curriculum:
            mid = (self.hard_limits[:, 0] + self.hard_limits[:, 1]) / 2
            r = self.hard_limits[:, 1] - self.hard_limits[:, 0]
            lo = mid - 0.5 * r * self.soft_pos_val
            hi = mid + 0.5 * r * self.soft_pos_val
        else:
            lo, hi = self.soft_limits[:, 0], self.soft_limits[:, 1]
        out = -(q - lo).clip(max=0.0)
        out = out + (q - hi).clip(min=0.0)
        return torch.sum(out, dim=1)

    def _reward_limits_dof_vel(self):
        lc = self.cfg.rewards.reward_limit
        v = self.soft_vel_val if lc.reward_limits_curriculum.soft_dof_vel_curriculum else lc.soft_dof_vel_limit
        return torch.sum((torch.abs(self.s["dof_vel"]) - self.dof_vel_limits * v).clip(min=0.0, max=1.0), dim=1)

    def _reward_limits_torque(self):
        lc = self.cfg.rewards.reward_limit
        if lc.reward_limits_curriculum.soft_torque_curriculum:
            return torch.sum((torch.abs(self.s["torques"]) - self.torque_limits * self.soft_tau_val).clip(min=0.0, max=1.0), dim=1)
        return torch.sum((torch.abs(self.s["torques"]) - self.torque_limits * lc.soft_torque_limit).clip(min=0.0), dim=1)

    def _reward_collision(self):
        return torch.sum(1.0 * (torch.norm(self.s["contact_forces"][:, self.penalised, :], dim=-1) > 0.1), dim=1)
