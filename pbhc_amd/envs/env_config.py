"""Config tree (the reference's YAML schema) -> PbhcEnvConfig + device-side output maps.

Host-side, load-time.  Mirrors what the reference derives at construction:
  * obs dims / sorted-key layout      helpers.determine_obs_dim (utils/helpers.py:47-80),
                                      _post_config_observation_callback (legged_robot_base.py:787-793)
  * history buffers                   HistoryHandler.__init__ (envs/env_utils/history_handler.py:12-31)
  * reward list, scales * dt          _prepare_reward_function (legged_robot_base.py:167-233)
  * body index sets                   _setup_robot_body_indices (base_task.py:169-205),
                                      _init_tracking_config / _init_motion_extend (motion_tracking.py:203-242)
  * gains / limits                    _init_buffers (legged_robot_base.py:74-108), isaacgym._process_dof_props
A reward or observation name the kernels do not implement raises NotImplementedError — the
reference would call `_reward_<name>` / `_get_obs_<name>`; we refuse instead of silently skipping.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib

K = _lib.K

SIGMA_KEYS = ["teleop_max_joint_pos", "teleop_upper_body_pos", "teleop_lower_body_pos", "teleop_vr_3point_pos", "teleop_feet_pos",
              "teleop_body_rot", "teleop_body_vel", "teleop_body_ang_vel", "teleop_joint_pos", "teleop_joint_vel",
              # general tracking (rewards/motion_tracking/general_main.yaml)
              "teleop_key_body_pos", "teleop_anchor_body_pos", "teleop_anchor_body_rot", "local_key_body_pos", "local_key_body_rot",
              "key_body_vel", "key_body_ang_vel", "teleop_root_vel", "teleop_root_pose"]
TERM_SIGMAS = {
    "teleop_max_joint_position": [0], "teleop_body_position_extend": [1, 2], "teleop_vr_3point": [3], "teleop_body_position_feet": [4],
    "teleop_body_rotation_extend": [5], "teleop_body_velocity_extend": [6], "teleop_body_ang_velocity_extend": [7],
    "teleop_joint_position": [8], "teleop_joint_velocity": [9],
    "teleop_key_body_position": [10], "teleop_anchor_body_position": [11], "teleop_anchor_body_rotation": [12], "local_key_body_position": [13],
    "local_key_body_rotation": [14], "key_body_velocity": [15], "key_body_ang_velocity": [16], "teleop_root_vel": [17], "teleop_root_pose": [18],
}
V2_ONLY_TERMS = {"teleop_key_body_position", "teleop_anchor_body_position", "teleop_anchor_body_rotation", "local_key_body_position",
                 "local_key_body_rotation", "key_body_velocity", "key_body_ang_velocity", "teleop_root_vel", "teleop_root_pose", "teleop_contact_mask_v2"}
OBS_FEATURES = {
    "base_lin_vel": "BASE_LIN_VEL", "base_ang_vel": "BASE_ANG_VEL", "projected_gravity": "PROJECTED_GRAVITY", "dof_pos": "DOF_POS",
    "dof_vel": "DOF_VEL", "actions": "ACTIONS", "ref_motion_phase": "REF_MOTION_PHASE",
    "dif_local_rigid_body_pos": "DIF_LOCAL_RIGID_BODY_POS", "local_ref_rigid_body_pos": "LOCAL_REF_RIGID_BODY_POS",
    "vr_3point_pos": "VR_3POINT_POS", "dr_base_com": "DR_BASE_COM", "dr_link_mass": "DR_LINK_MASS", "dr_kp": "DR_KP", "dr_kd": "DR_KD",
    "dr_friction": "DR_FRICTION", "dr_ctrl_delay": "DR_CTRL_DELAY", "relyaw": "RELYAW", "base_pos_z": "BASE_POS_Z",
    "dif_joint_angles": "DIF_JOINT_ANGLES", "dif_joint_velocities": "DIF_JOINT_VELOCITIES",
    "local_ref_rigid_body_vel": "LOCAL_REF_RIGID_BODY_VEL", "global_ref_rigid_body_vel": "GLOBAL_REF_RIGID_BODY_VEL",
}
# general tracking getters (general_tracking.py:821-954): plain features ...
OBS_FEATURES_V2 = {
    "roll_pitch": "ROLL_PITCH", "root_height": "BASE_POS_Z", "contact_mask": "CONTACT_MASK", "ref_contact_mask": "REF_CONTACT_MASK",
    "dr_base_mass": "DR_BASE_MASS", "anchor_ref_pos": "ANCHOR_REF_POS", "anchor_ref_rot": "ANCHOR_REF_ROT",
    "dif_root_velocity": "DIF_ROOT_VELOCITY", "dif_root_rot": "DIF_ROOT_ROT", "dif_root_height": "DIF_ROOT_HEIGHT",
}
# ... and keys that are gathers out of per-body / per-step feature tables: key -> features they read
OBS_GATHERS_V2 = {
    "local_key_body_pos": ["LOCAL_BODY_POS"], "local_key_body_rot": ["LOCAL_BODY_ROT"], "dif_local_key_body_pos": ["DIF_LOCAL_RIGID_BODY_POS"],
    "local_ref_key_body_pos": ["LOCAL_REF_RIGID_BODY_POS"], "future_motion_root_height": ["FUT_ROOT_HEIGHT"], "future_motion_roll_pitch": ["FUT_ROLL_PITCH"],
    "future_motion_base_lin_vel": ["FUT_BASE_LIN_VEL"], "future_motion_base_ang_vel": ["FUT_BASE_ANG_VEL"], "future_motion_base_yaw_vel": ["FUT_BASE_ANG_VEL"],
    "future_motion_dof_pos": ["FUT_DOF_POS"], "future_motion_local_ref_key_body_pos": ["FUT_LOCAL_KEY_POS"],
    "next_step_ref_motion": ["FUT_ROOT_HEIGHT", "FUT_ROLL_PITCH", "FUT_BASE_LIN_VEL", "FUT_BASE_ANG_VEL", "FUT_DOF_POS", "FUT_LOCAL_KEY_POS"],
}


def flatten_obs_dims(obs_cfg):
    d = obs_cfg.obs_dims
    if isinstance(d, list):
        return {k: int(v) for item in d for k, v in item.items()}
    return {k: int(v) for k, v in d.items()}


def determine_obs_dim(cfg):
    """helpers.determine_obs_dim: fills cfg.robot.algo_obs_dim_dict and returns (group dims, key dims, aux dims)."""
    ob = cfg.obs
    assert set(ob.noise_scales.keys()) == set(ob.obs_scales.keys())
    dims = flatten_obs_dims(ob)
    ob.obs_dims = dims
    aux = {}
    for aux_key, aux_cfg in ob.obs_auxiliary.items():
        aux[aux_key] = sum(dims[k] * n for k, n in aux_cfg.items())
    groups = {}
    for g, keys in ob.obs_dict.items():
        tot = 0
        for key in keys:
            k = key[:-4] if key.endswith("_raw") else key
            tot += dims[k] if k in dims else aux[k]
        groups[g] = tot
    cfg.robot.algo_obs_dim_dict = groups
    return groups, dims, aux


import os as _os

PACKED_MAPS = _os.environ.get("PBHC_PACKED_MAPS", "1") != "0"     # compact 16-bit observation maps staged in LDS (PbhcEnvConfig.map_image); 0: diagnosis only


class EnvLayout:
    """Everything the host needs to know about the layouts the kernels use."""


def build(cfg, skel, motion_lib, num_envs, device, sim_link_mass_dim, seed=0, mode=0):
    """mode 0: LeggedRobotMotionTracking, mode 1: LeggedRobotGeneralTracking."""
    ec = cfg.env.config
    rc = cfg.robot
    rw = cfg.rewards
    ob = cfg.obs
    dr = cfg.domain_rand
    D, B, Bx = skel.num_dof, skel.num_bodies, skel.num_bodies_ext
    if list(rc.dof_names) and len(rc.dof_names) != D:
        raise _lib.PbhcError("config dof_names do not match the skeleton")
    c = _lib.PbhcEnvConfig()
    L = EnvLayout()
    c.abi_version = K["PBHC_ABI_VERSION"]
    c.tracking_mode = mode
    c.num_envs = num_envs
    c.skel = skel.to_c()
    sim = cfg.simulator.config.sim
    dt = sim.control_decimation * (1.0 / sim.fps)
    c.dt = dt
    L.dt = dt
    L.max_episode_length = float(np.ceil(ec.max_episode_length_s / dt))
    c.max_episode_length = L.max_episode_length
    c.max_episode_length_s = float(ec.max_episode_length_s)
    # ---- control
    ctrl = rc.control
    if ctrl.control_type not in ("P", "V", "T"):
        raise NameError(f"Unknown controller type: {ctrl.control_type}")            # legged_robot_base.py:817
    c.control_type = {"P": 0, "V": 1, "T": 2}[ctrl.control_type]
    c.sim_dt = 1.0 / sim.fps
    for i, name in enumerate(rc.dof_names):
        c.default_dof_pos[i] = float(rc.init_state.default_joint_angles[name])
        found = False
        for k in ctrl.stiffness.keys():
            if k in name:
                c.p_gains[i] = float(ctrl.stiffness[k])
                c.d_gains[i] = float(ctrl.damping[k])
                found = True
                c.action_scale[i] = float(ctrl.action_scale if isinstance(ctrl.action_scale, (int, float)) else ctrl.action_scale[k])
        if not found:
            raise ValueError(f"PD gain of joint {name} were not defined. Should be defined in the yaml file.")
        c.torque_limits[i] = float(rc.dof_effort_limit_list[i])
        c.dof_vel_limits[i] = float(rc.dof_vel_limit_list[i])
        lo, hi = np.float32(rc.dof_pos_lower_limit_list[i]), np.float32(rc.dof_pos_upper_limit_list[i])
        c.hard_dof_pos_limits[i][0], c.hard_dof_pos_limits[i][1] = float(lo), float(hi)
        m = np.float32((lo + hi) / np.float32(2))
        r = np.float32(hi - lo)
        s = rw.reward_limit.soft_dof_pos_limit
        c.soft_dof_pos_limits[i][0] = float(np.float32(m - np.float32(np.float32(0.5) * r) * np.float32(s)))
        c.soft_dof_pos_limits[i][1] = float(np.float32(m + np.float32(np.float32(0.5) * r) * np.float32(s)))
    c.action_clip_value = float(ctrl.action_clip_value)
    c.clip_torques = int(bool(ctrl.clip_torques))
    c.randomize_torque_rfi = int(bool(dr.randomize_torque_rfi))
    c.rfi_lim = float(dr.get("rfi_lim", 0.0))
    c.use_rao = int(bool(dr.use_rao))
    c.rao_lim = float(dr.get("rao_lim", 0.0))
    c.randomize_ctrl_delay = int(bool(dr.randomize_ctrl_delay))
    c.queue_len = int(dr.ctrl_delay_step_range[1]) + 1 if dr.randomize_ctrl_delay else 1
    c.ctrl_delay_range[0], c.ctrl_delay_range[1] = int(dr.ctrl_delay_step_range[0]), int(dr.ctrl_delay_step_range[1])
    c.randomize_pd_gain = int(bool(dr.randomize_pd_gain))
    c.kp_range[0], c.kp_range[1] = float(dr.kp_range[0]), float(dr.kp_range[1])
    c.kd_range[0], c.kd_range[1] = float(dr.kd_range[0]), float(dr.kd_range[1])
    c.randomize_rfi_lim = int(bool(dr.randomize_rfi_lim))
    c.rfi_lim_range[0], c.rfi_lim_range[1] = float(dr.rfi_lim_range[0]), float(dr.rfi_lim_range[1])
    for unsupported in ("parallel_serial_pd", "parallel_serial_tau"):
        if unsupported in dr and dr[unsupported].get("enable", False):
            raise NotImplementedError(f"domain_rand.{unsupported}")
    c.randomize_default_dof_pos = int(bool(dr.get("randomize_default_dof_pos", False)))     # legged_robot_base.py:632-635
    if c.randomize_default_dof_pos:
        c.dof_pos_range[0], c.dof_pos_range[1] = float(dr.dof_pos_range[0]), float(dr.dof_pos_range[1])
    # ---- body index sets
    names = skel.body_names
    ext = skel.body_names_ext
    feet = [names.index(s) for s in names if rc.foot_name in s]
    if len(feet) != 2:
        raise _lib.PbhcError(f"expected 2 feet, found {len(feet)}")
    c.num_feet = len(feet)
    for i, f in enumerate(feet):
        c.feet[i] = f
    pen = []
    for n in rc.penalize_contacts_on:
        pen.extend([names.index(s) for s in names if n in s])
    c.num_penalised = len(pen)
    for i, p in enumerate(pen):
        c.penalised[i] = p
    m = rc.motion
    upper = [ext.index(l) for l in m.get("upper_body_link", [])]
    lower = [ext.index(l) for l in m.get("lower_body_link", [])]
    track = [ext.index(l) for l in m.get("motion_tracking_link", [])]
    c.num_upper, c.num_lower, c.num_track = len(upper), len(lower), len(track)
    for i, b in enumerate(upper):
        c.upper[i] = b
    for i, b in enumerate(lower):
        c.lower[i] = b
    for i, b in enumerate(track):
        c.track[i] = b
    key_ids, body_z = [], []
    if mode == 1:
        key_ids = [ext.index(l) for l in rc.key_bodies]                      # general_tracking.py:94-95
        anchor_link = m.get("anchor_link", "pelvis_link")
        c.anchor_index = (names.index(anchor_link) if anchor_link in names else -1) + 1      # find_rigid_body_indice(...) + 1, sic (:97-98)
        c.num_key = len(key_ids)
        for i, b in enumerate(key_ids):
            c.key[i] = b
        body_z = [4, 10, 24, 25, 26]                                          # hard-coded in the reference (:253)
    for b in range(Bx):
        c.body_flags[b] = ((1 if b in upper else 0) | (2 if b in lower else 0) | (4 if b in track else 0) | (8 if b in feet else 0)
                           | (16 if b in key_ids else 0) | (32 if b in body_z else 0))
        c.track_slot[b] = track.index(b) if b in track else -1
        c.key_slot[b] = key_ids.index(b) if b in key_ids else -1
    L.feet, L.penalised, L.upper, L.lower, L.track, L.key = feet, pen, upper, lower, track, key_ids
    # ---- termination
    T = ec.termination
    if T.get("terminate_when_dof_far", False):
        # motion_tracking.py:345 reduces torch.any(norm(dif_joint_angles) > threshold) over the ENV axis: one env past the threshold resets all
        # of them — a batch-global decision inside a step (a grid-wide exchange before the reset path; a collective per step under data parallelism)
        raise NotImplementedError("termination.terminate_when_dof_far")
    if float(ec.get("noise_to_initial_level", 0) or 0) != 0.0:
        # motion_tracking.py:481-484,538-539 / general_tracking.py:413-416,478-479: normal noise on the state an env is reset to (and on the
        # observations of that step); 0 in every shipped yaml.  The in-kernel reset path writes the reference state as it is.
        raise NotImplementedError("noise_to_initial_level != 0 (noise on the reset state)")
    if ec.get("use_teleop_control", False):
        raise NotImplementedError("use_teleop_control (a ROS subscriber feeding marker coordinates, motion_tracking.py:112-118)")
    sdc = ec.get("soft_dynamic_correction", None)
    if sdc is not None and sdc.get("enable", False):
        # motion_tracking.py:806-860 blends the SIMULATOR's state towards the reference between physics sub-steps: it acts on the physics, which
        # the replay stub does not have (the states it hands out are whatever the replay holds)
        raise NotImplementedError("soft_dynamic_correction (acts inside the simulator's physics sub-steps; the replay stub has none)")
    # legged_robot_base.py:449-479 + isaacgym.py:387-388: probabilistic terminations near the joint limits (one uniform per gate and STEP)
    TS, TP = ec.termination_scales, ec.get("termination_probality", {})
    c.terminate_close_pos = int(bool(T.get("terminate_when_close_to_dof_pos_limit", False)))
    c.terminate_close_vel = int(bool(T.get("terminate_when_close_to_dof_vel_limit", False)))
    c.terminate_close_tau = int(bool(T.get("terminate_when_close_to_torque_limit", False)))
    if c.terminate_close_pos:
        c.term_close_prob[0] = float(TP.terminate_when_close_to_dof_pos_limit)
        for i in range(D):
            lo, hi = np.float32(rc.dof_pos_lower_limit_list[i]), np.float32(rc.dof_pos_upper_limit_list[i])
            m, r = np.float32((lo + hi) / np.float32(2)), np.float32(hi - lo)
            sc = np.float32(TS.termination_close_to_dof_pos_limit)
            c.dof_pos_limits_termination[i][0] = float(np.float32(m - np.float32(np.float32(0.5) * r) * sc))
            c.dof_pos_limits_termination[i][1] = float(np.float32(m + np.float32(np.float32(0.5) * r) * sc))
    if c.terminate_close_vel:
        c.term_close_prob[1] = float(TP.terminate_when_close_to_dof_vel_limit)
        c.term_close_vel_scale = float(TS.termination_close_to_dof_vel_limit)
    if c.terminate_close_tau:
        c.term_close_prob[2] = float(TP.terminate_when_close_to_torque_limit)
        c.term_close_tau_scale = float(TS.termination_close_to_torque_limit)
    c.terminate_by_contact = int(bool(T.get("terminate_by_contact", False)))                 # legged_robot_base.py:434-436
    tcon = []
    for n in rc.get("terminate_after_contacts_on", []):                                     # base_task.py:178-180,195-197
        tcon.extend([names.index(s_) for s_ in names if n in s_])
    if c.terminate_by_contact and len(tcon) > K["PBHC_MAX_IDX"]:
        raise _lib.PbhcError("too many terminate_after_contacts_on bodies")
    L.termination_contact = tcon                                                             # (the kernel gets the list only when the switch is on)
    c.num_term_contact = len(tcon) if c.terminate_by_contact else 0
    for i, b_ in enumerate(tcon[:c.num_term_contact]):
        c.term_contact[i] = b_
    c.terminate_by_low_height = int(bool(T.get("terminate_by_low_height", False)))           # :442-444
    c.termination_min_base_height = float(ec.termination_scales.get("termination_min_base_height", 0.0))
    c.terminate_by_gravity = int(bool(T.terminate_by_gravity))
    c.termination_gravity = float(ec.termination_scales.termination_gravity)
    c.terminate_when_motion_far = int(bool(T.terminate_when_motion_far))
    c.terminate_when_motion_end = int(bool(T.terminate_when_motion_end))
    ts = ec.termination_scales
    if mode == 1:                                                             # general_tracking.py:241-254
        c.terminate_by_ref_pos_z = int(bool(T.get("terminate_by_ref_pos_z", False)))
        c.terminate_by_ref_ori = int(bool(T.get("terminate_by_ref_ori", False)))
        c.terminate_by_body_z = int(bool(T.get("terminate_by_body_z", False)))
        c.ref_pos_z_threshold = float(ts.get("terminate_by_ref_pos_z_threshold", 0.25))
        c.ref_ori_threshold = float(ts.get("terminate_by_ref_ori_threshold", 0.8))
        c.body_z_threshold = float(ts.get("terminate_by_body_z_threshold", 0.25))
        if c.terminate_by_body_z and max(body_z) >= Bx:
            raise IndexError(f"terminate_by_body_z indexes body {max(body_z)} of {Bx}")      # the reference would raise the same way
        if T.get("terminate_when_local_motion_far", False):
            pass                                                              # read by nobody in the reference either
    elif any(T.get(k, False) for k in ("terminate_by_ref_pos_z", "terminate_by_ref_ori", "terminate_by_body_z")):
        raise NotImplementedError("general-tracking terminations need env._target_ ...general_tracking.LeggedRobotGeneralTracking")
    tc = ec.termination_curriculum
    c.motion_far_curriculum = int(bool(tc.terminate_when_motion_far_curriculum))
    c.motion_far_degree = float(tc.terminate_when_motion_far_curriculum_degree)
    c.motion_far_down = float(tc.terminate_when_motion_far_curriculum_level_down_threshold)
    c.motion_far_up = float(tc.terminate_when_motion_far_curriculum_level_up_threshold)
    c.motion_far_min = float(tc.terminate_when_motion_far_threshold_min)
    c.motion_far_max = float(tc.terminate_when_motion_far_threshold_max)
    # ---- rewards
    scales = {}
    for k, v in rw.reward_scales.items():
        if v != 0:
            scales[k] = v * dt
    L.reward_scales = scales
    L.reward_names = [k for k in scales if k != "termination"]
    L.sum_names = list(scales.keys())
    c.num_terms = len(L.reward_names)
    if c.num_terms > min(K["PBHC_MAX_TERMS"], 31):
        raise _lib.PbhcError("too many reward terms")
    c.use_vec_reward = int(bool(ec.use_vec_reward))
    c.num_rew_cols = c.num_terms + 1 if ec.use_vec_reward else 1
    L.num_rew_fn = c.num_rew_cols
    pen_names = set(rw.reward_penalty_reward_names)
    for i, name in enumerate(L.reward_names):
        key = "PBHC_R_" + name.upper()
        if key not in K or (mode == 0 and name in V2_ONLY_TERMS):
            raise NotImplementedError(f"reward term {name!r} has no HIP implementation")
        c.term_id[i] = K[key]
        if name in ("feet_heading_alignment", "feet_heading_alignment_contact", "penalty_feet_ori", "penalty_feet_ori_contact"):
            c.foot_ori_terms = 1
        c.term_scale[i] = float(scales[name])
        c.term_penalty[i] = int(name in pen_names and bool(rw.reward_penalty_curriculum))
        c.term_sum_col[i] = L.sum_names.index(name)
        for s in TERM_SIGMAS.get(name, []):
            c.sigma_active[s] = 1
    c.has_termination = int("termination" in scales)
    if c.has_termination:
        c.termination_scale = float(scales["termination"])
        c.termination_sum_col = L.sum_names.index("termination")
    c.num_sum_cols = len(L.sum_names)
    c.only_positive_rewards = int(bool(rw.only_positive_rewards))
    c.body_pos_lower_weight = float(rw.get("teleop_body_pos_lowerbody_weight", 1.0))
    c.body_pos_upper_weight = float(rw.get("teleop_body_pos_upperbody_weight", 1.0))
    c.desired_feet_air_time = float(rw.get("desired_feet_air_time", 0.0))
    c.max_contact_force = float(rw.get("locomotion_max_contact_force", 0.0))
    ats = rw.get("adaptive_tracking_sigma", {})
    c.adaptive_sigma = int(bool(ats.get("enable", False)))
    atype = ats.get("type", "origin")
    if atype not in ("origin", "mean", "scale"):
        atype = None                                       # the reference's if/elif chain leaves sigma untouched (only the EMA moves)
        raise NotImplementedError(f"adaptive_tracking_sigma.type {ats.get('type')!r}")
    c.adaptive_type = {"origin": 0, "mean": 3 if mode == 1 else 1, "scale": 2}[atype]
    c.adaptive_scale = float(ats.get("scale", 1.0))
    c.adaptive_alpha = float(ats.get("alpha", 0.0))
    c.penalty_curriculum = int(bool(rw.reward_penalty_curriculum))
    c.penalty_degree = float(rw.reward_penalty_degree)
    c.penalty_down = float(rw.reward_penalty_level_down_threshold)
    c.penalty_up = float(rw.reward_penalty_level_up_threshold)
    c.penalty_min = float(rw.reward_min_penalty_scale)
    c.penalty_max = float(rw.reward_max_penalty_scale)
    c.noise_curriculum = int(bool(ob.get("add_noise_currculum", False)))
    if c.noise_curriculum:                                   # legged_robot_base.py:1117-1126 (the up-threshold is the penalty curriculum's)
        c.noise_degree = float(ob.soft_dof_pos_curriculum_degree)
        c.noise_down = float(ob.soft_dof_pos_curriculum_level_down_threshold)
        c.noise_up = float(rw.reward_penalty_level_up_threshold)
        c.noise_min = float(ob.noise_value_min)
        c.noise_max = float(ob.noise_value_max)
    c.num_compute_average_epl = int(rw.num_compute_average_epl)
    lc = rw.reward_limit.reward_limits_curriculum
    c.soft_pos_curriculum = int(bool(lc.soft_dof_pos_curriculum))
    c.soft_vel_curriculum = int(bool(lc.soft_dof_vel_curriculum))
    c.soft_tau_curriculum = int(bool(lc.soft_torque_curriculum))
    for q, pre in enumerate(("soft_dof_pos", "soft_dof_vel", "soft_torque")):       # legged_robot_base.py:902-939 (device rule in k_env_finalize)
        if lc[pre + "_curriculum"]:
            c.soft_cur_degree[q] = float(lc[pre + "_curriculum_degree"])
            c.soft_cur_down[q] = float(lc[pre + "_curriculum_level_down_threshold"])
            c.soft_cur_up[q] = float(lc[pre + "_curriculum_level_up_threshold"])
            c.soft_cur_min[q] = float(lc[pre + "_min_limit"])
            c.soft_cur_max[q] = float(lc[pre + "_max_limit"])
    c.soft_dof_vel_limit = float(rw.reward_limit.soft_dof_vel_limit)
    c.soft_torque_limit = float(rw.reward_limit.soft_torque_limit)
    # ---- features
    groups, dims, aux = determine_obs_dim(cfg)
    S = int(ob.get("future_num_steps", 0)) if mode == 1 else 0
    if S:
        steps = torch.linspace(start=1, end=ob.future_max_steps, steps=S, dtype=torch.long).tolist()      # general_tracking.py:501-507
        if S > K["PBHC_MAX_FUTURE"]:
            raise _lib.PbhcError("too many future steps")
        c.future_num_steps = S
        for i, v in enumerate(steps):
            c.future_steps[i] = int(v)
        L.future_steps = steps
    Kb = len(key_ids)
    feats = dict(OBS_FEATURES)
    if mode == 1:
        feats.update(OBS_FEATURES_V2)
    mult = lambda k: S if (k.startswith("future_motion_") and S) else 1       # future keys list their PER-STEP dim (obs_ppo_teacher.yaml)
    # widths of the tensors the env hands out: the future group is [N, S * per-step dim] (ppo_mimic.py:206-216)
    L.obs_dims = dims
    L.group_dims = {}
    for g, keys in ob.obs_dict.items():
        tot = 0
        for key_ in keys:
            k_ = key_[:-4] if key_.endswith("_raw") else key_
            tot += dims[k_] * mult(k_) if k_ in dims else aux[k_]
        L.group_dims[g] = tot
    groups = L.group_dims
    hist_len = {}
    for aux_cfg in ob.obs_auxiliary.values():
        for k, n in aux_cfg.items():
            hist_len[k] = max(hist_len.get(k, 0), int(n))
    hist_keys = sorted(hist_len.keys())
    hist_off, o = {}, 0
    for k in hist_keys:
        hist_off[k] = o
        o += hist_len[k] * dims[k]
    c.hist_dim = max(o, 1)
    L.hist_keys, L.hist_len, L.hist_off, L.hist_dim = hist_keys, hist_len, hist_off, c.hist_dim
    fdim = {
        "BASE_LIN_VEL": 3, "BASE_ANG_VEL": 3, "PROJECTED_GRAVITY": 3, "DOF_POS": D, "DOF_VEL": D, "ACTIONS": D, "REF_MOTION_PHASE": 1,
        "DIF_LOCAL_RIGID_BODY_POS": 3 * Bx, "LOCAL_REF_RIGID_BODY_POS": 3 * Bx, "VR_3POINT_POS": 3 * max(len(track), 1),
        "DR_BASE_COM": 3, "DR_LINK_MASS": max(sim_link_mass_dim, 1), "DR_KP": D, "DR_KD": D, "DR_FRICTION": 1, "DR_CTRL_DELAY": 1,
        "RELYAW": 1, "BASE_POS_Z": 1, "DIF_JOINT_ANGLES": D, "DIF_JOINT_VELOCITIES": D, "LOCAL_REF_RIGID_BODY_VEL": 3 * Bx,
        "GLOBAL_REF_RIGID_BODY_VEL": 3 * Bx, "HISTORY": c.hist_dim, "ZERO": 1,
        "ROLL_PITCH": 2, "CONTACT_MASK": 2, "DR_BASE_MASS": 1, "LOCAL_BODY_POS": 3 * Bx, "LOCAL_BODY_ROT": 6 * Bx, "ANCHOR_REF_POS": 3,
        "ANCHOR_REF_ROT": 6, "DIF_ROOT_VELOCITY": 3, "DIF_ROOT_ROT": 4, "DIF_ROOT_HEIGHT": 1, "REF_CONTACT_MASK": 2,
        "FUT_ROOT_HEIGHT": max(S, 1), "FUT_ROLL_PITCH": max(2 * S, 1), "FUT_BASE_LIN_VEL": max(3 * S, 1), "FUT_BASE_ANG_VEL": max(3 * S, 1),
        "FUT_DOF_POS": max(S * D, 1), "FUT_LOCAL_KEY_POS": max(S * Kb * 3, 1),
    }
    # which features do the observation maps read?  the kernel skips the others (feat_off = -1)
    used = {"HISTORY", "ZERO"}
    def mark(k):
        if k in feats:
            used.add(feats[k])
        elif mode == 1 and k in OBS_GATHERS_V2:
            used.update(OBS_GATHERS_V2[k])

    for keys in ob.obs_dict.values():
        for key in keys:
            mark(key[:-4] if key.endswith("_raw") else key)
    for hk in hist_keys:
        mark(hk)
    off = 0
    feat_off = {}
    for name, n in fdim.items():
        if name in used and name != "HISTORY":
            feat_off[name] = off
            c.feat_off[K["PBHC_F_" + name]] = off
            off += n
    trash = off                       # features nobody reads share one scratch region at the end of the row
    for name, n in fdim.items():
        if name not in used:
            c.feat_off[K["PBHC_F_" + name]] = trash
            off = max(off, trash + n)
    # HISTORY is the LAST block of the feature index space: a specialised kernel keeps it out of the LDS feature row (csrc/pbhc_env_step.h:
    # step_lds_plan — the old history waits in registers and is staged over dead arrays once the termination flags are known)
    feat_off["HISTORY"] = off
    c.feat_off[K["PBHC_F_HISTORY"]] = off
    off += fdim["HISTORY"]
    c.feat_dim = off
    # readiness class of every feature word (see the compact maps below): which phase of the step kernel produces it
    CLASS0 = {"HISTORY", "ZERO", "BASE_LIN_VEL", "BASE_ANG_VEL", "PROJECTED_GRAVITY", "REF_MOTION_PHASE", "RELYAW", "ROLL_PITCH", "DR_BASE_COM",
              "DR_LINK_MASS", "DR_FRICTION", "DR_BASE_MASS", "REF_CONTACT_MASK", "FUT_ROOT_HEIGHT", "FUT_ROLL_PITCH", "FUT_BASE_LIN_VEL",
              "FUT_BASE_ANG_VEL", "FUT_DOF_POS", "FUT_LOCAL_KEY_POS"}
    CLASS2 = {"DOF_POS", "DOF_VEL", "ACTIONS", "DR_KP", "DR_KD", "DR_CTRL_DELAY", "BASE_POS_Z", "CONTACT_MASK"}
    feat_class = np.ones(max(off, 1), dtype=np.int64)
    for name, o_ in feat_off.items():
        feat_class[o_:o_ + fdim[name]] = 0 if name in CLASS0 else (2 if name in CLASS2 else 1)
    c.dr_link_mass_dim = sim_link_mass_dim
    L.feat_off, L.feat_dim_each = feat_off, fdim

    def key_sources(key):
        """feature-row indices of observation key `key` (flat, in the reference's element order)."""
        if key in ob.obs_auxiliary:                      # _get_obs_history_* (motion_tracking.py:993-1015)
            idx = []
            a = ob.obs_auxiliary[key]
            for hk in sorted(a.keys()):
                n = int(a[hk])
                base = feat_off["HISTORY"] + hist_off[hk]
                idx.extend(range(base, base + n * dims[hk]))
            return idx
        if mode == 1 and key in OBS_GATHERS_V2:
            fo = lambda f: feat_off[f]
            per_body = lambda f, w: [fo(f) + w * b + j for b in key_ids for j in range(w)]
            if key == "local_key_body_pos":
                idx = per_body("LOCAL_BODY_POS", 3)
            elif key == "local_key_body_rot":
                idx = per_body("LOCAL_BODY_ROT", 6)
            elif key == "dif_local_key_body_pos":
                idx = per_body("DIF_LOCAL_RIGID_BODY_POS", 3)
            elif key == "local_ref_key_body_pos":
                idx = per_body("LOCAL_REF_RIGID_BODY_POS", 3)
            elif key == "future_motion_base_yaw_vel":
                idx = [fo("FUT_BASE_ANG_VEL") + 3 * st + 2 for st in range(S)]
            elif key == "next_step_ref_motion":                       # step-0 slices, general_tracking.py:554-564
                idx = ([fo("FUT_ROOT_HEIGHT")] + [fo("FUT_ROLL_PITCH") + j for j in range(2)] + [fo("FUT_BASE_LIN_VEL") + j for j in range(3)]
                       + [fo("FUT_BASE_ANG_VEL") + 2] + [fo("FUT_DOF_POS") + j for j in range(D)] + [fo("FUT_LOCAL_KEY_POS") + j for j in range(3 * Kb)])
            else:
                f = OBS_GATHERS_V2[key][0]
                idx = list(range(fo(f), fo(f) + fdim[f]))
            if (key.startswith("future_") or key == "next_step_ref_motion") and not S:
                raise _lib.PbhcError(f"observation {key!r} needs obs.future_num_steps > 0")
            if len(idx) != dims[key] * mult(key):
                raise _lib.PbhcError(f"obs_dims[{key}]={dims[key]} does not match the {len(idx)} values the env produces")
            return idx
        if key not in feats:
            raise NotImplementedError(f"observation {key!r} has no HIP implementation")
        f = feats[key]
        if dims[key] > fdim[f]:
            raise _lib.PbhcError(f"obs_dims[{key}]={dims[key]} exceeds the feature size {fdim[f]}")
        idx = list(range(feat_off[f], feat_off[f] + dims[key]))
        if key == "dof_vel" and mode == 1 and ob.get("masked_dof_vel", False):       # general_tracking.py:821-829
            for j in (4, 5, 10, 11):
                idx[j] = feat_off["ZERO"]
        return idx

    # ---- output maps: every output element as (dst, src, scale, noise)
    group_names = list(ob.obs_dict.keys())
    G = len(group_names) + 1                                   # + history write-back
    if G > K["PBHC_MAX_GROUPS"]:
        raise _lib.PbhcError("too many observation groups")
    maps = []           # (name, dst, src, scale, noise, clip, pitch)
    for gi, g in enumerate(group_names):
        keys = ob.obs_dict[g]
        dst, src, sc, ns = [], [], [], []
        pos = 0
        for key in sorted(keys):
            raw = key.endswith("_raw")
            k = key[:-4] if raw else key
            scale, noise = float(ob.obs_scales[k]), (0.0 if raw else float(ob.noise_scales[k]))
            if k in ob.obs_auxiliary:                      # _get_obs_history_* (motion_tracking.py:993-1015)
                a = ob.obs_auxiliary[k]
                for hk in sorted(a.keys()):
                    n = int(a[hk]) * dims[hk]
                    base = feat_off["HISTORY"] + hist_off[hk]
                    dst.extend(range(pos, pos + n)); src.extend(range(base, base + n)); sc.extend([scale] * n); ns.extend([noise] * n)
                    pos += n
            else:
                idx = key_sources(k)
                dst.extend(range(pos, pos + len(idx))); src.extend(idx); sc.extend([scale] * len(idx)); ns.extend([noise] * len(idx))
                pos += len(idx)
        assert pos == groups[g], (g, pos, groups[g])
        maps.append((g, dst, src, sc, ns, 1, groups[g]))
    # history write-back: new[k][0] = parse(current k), new[k][t] = old[k][t-1]  (history_handler.py:40-44)
    dst, src, sc, ns = [], [], [], []
    for hk in hist_keys:
        cur = key_sources(hk)
        o0 = hist_off[hk]
        dst.extend(range(o0, o0 + len(cur))); src.extend(cur); sc.extend([float(ob.obs_scales[hk])] * len(cur)); ns.extend([float(ob.noise_scales[hk])] * len(cur))
        n_old = (hist_len[hk] - 1) * dims[hk]
        base = feat_off["HISTORY"] + o0
        dst.extend(range(o0 + dims[hk], o0 + dims[hk] + n_old)); src.extend(range(base, base + n_old)); sc.extend([1.0] * n_old); ns.extend([0.0] * n_old)
    if not src:
        dst, src, sc, ns = [0], [feat_off["ZERO"]], [1.0], [0.0]
    maps.append(("__history__", dst, src, sc, ns, 0, c.hist_dim))
    c.num_groups = len(maps)
    L.group_names = [m[0] for m in maps]
    L.map_tensors = []
    # compact form (16 bits per element, staged in LDS by the kernel): possible when every group writes its row in order and has
    # at most PBHC_MAX_SEGS distinct (scale, noise) pairs
    seg_tables = []
    compact = PACKED_MAPS and off <= 4096
    for (g, dst, src, sc, ns, clip, pitch) in maps:
        pairs = sorted(set(zip(sc, ns)))
        seg_tables.append(pairs)
        if dst != list(range(len(dst))) or len(pairs) > K["PBHC_MAX_SEGS"]:
            compact = False
    lds_off = 0
    image = []
    for i, (g, dst, src, sc, ns, clip, pitch) in enumerate(maps):
        if not src:                                            # a group made only of old history
            dst, src, sc, ns = [0], [feat_off["ZERO"]], [1.0], [0.0]
            # harmless: rewrites element 0 ... only valid if element 0 is not early-written; guard below
            raise _lib.PbhcError(f"observation group {g} has no non-history element")
        td = torch.tensor(dst, dtype=torch.int32, device=device)
        ts = torch.tensor(src, dtype=torch.int32, device=device)
        tsc = torch.tensor(sc, dtype=torch.float32, device=device)
        tn = torch.tensor(ns, dtype=torch.float32, device=device)
        L.map_tensors.append((td, ts, tsc, tn))
        c.groups[i].dim = len(src)
        c.groups[i].clip = clip
        c.groups[i].pitch = pitch
        c.groups[i].dst = td.data_ptr()
        c.groups[i].src = ts.data_ptr()
        c.groups[i].scale = tsc.data_ptr()
        c.groups[i].noise = tn.data_ptr()
        # the same map as runs of consecutive (dst, src) with one scale / noise / readiness class (PbhcObsRun): what the config-specialised
        # kernel unrolls into straight-line code
        runs = []
        for j in range(len(src)):
            late_j = int(feat_class[src[j]]) == 2
            if runs and runs[-1][0] + runs[-1][2] == dst[j] and runs[-1][1] + runs[-1][2] == src[j] and runs[-1][3] == late_j \
                    and runs[-1][4] == sc[j] and runs[-1][5] == ns[j] and src[j] != feat_off["HISTORY"]:        # (no run straddles the history block)
                runs[-1][2] += 1
            else:
                runs.append([dst[j], src[j], 1, late_j, sc[j], ns[j]])
        if len(runs) <= K["PBHC_MAX_RUNS"]:
            c.groups[i].num_runs = len(runs)
            for r_, (d_, s_, n_, l_, a_, b_) in enumerate(runs):
                R = c.groups[i].runs[r_]
                R.dst, R.src, R.len, R.late, R.scale, R.noise = int(d_), int(s_), int(n_), int(l_), float(a_), float(b_)
        else:
            c.groups[i].num_runs = -1
        if compact:
            pairs = seg_tables[i]
            seg_of = {p: k for k, p in enumerate(pairs)}
            pk = np.array([s_ | (seg_of[(a_, b_)] << 12) for s_, a_, b_ in zip(src, sc, ns)] + [0] * (len(src) % 2), dtype=np.uint16)
            tabs = np.zeros(32, dtype=np.float32)
            for k, (a_, b_) in enumerate(pairs):
                tabs[k], tabs[16 + k] = a_, b_
            if len(src) >= 65536:
                raise _lib.PbhcError("observation group too wide for the compact maps")
            # The kernel writes a row in element PAIRS (one 8-byte store), in two passes by readiness of the pair's sources (feat_class):
            # "early" = everything but the post-reset features (history, DR, per-env scalars, reference / future targets, difference
            # features), written while the dynamics chain still runs; "late" = pairs that read a post-reset feature.  A pair that holds a
            # noisy element belongs to neither: both of its elements go to the noise list, which the kernel visits last.
            n_el = len(src)
            npair = (n_el + 1) // 2
            pair_noisy = [any(ns[j] != 0.0 for j in (2 * p_, 2 * p_ + 1) if j < n_el) for p_ in range(npair)]
            pair_late = [any(int(feat_class[src[j]]) == 2 for j in (2 * p_, 2 * p_ + 1) if j < n_el) for p_ in range(npair)]
            early = [p_ for p_ in range(npair) if not pair_noisy[p_] and not pair_late[p_]]
            late = [p_ for p_ in range(npair) if not pair_noisy[p_] and pair_late[p_]]
            plist = np.array(early + late + [0] * ((len(early) + len(late)) % 2), dtype=np.uint16)
            noisy_e = [j | (int(pk[j]) << 16) for j in range(n_el) if pair_noisy[j // 2] and not pair_late[j // 2]]
            noisy_l = [j | (int(pk[j]) << 16) for j in range(n_el) if pair_noisy[j // 2] and pair_late[j // 2]]
            if noisy_e:
                noisy_e += [noisy_e[-1]] * ((-len(noisy_e)) % 4)      # early list padded to a Philox quad (repeats rewrite the same value)
            noisy = np.array(noisy_e + noisy_l, dtype=np.uint32)
            hdr = np.array([len(noisy_e), len(noisy_l), len(early), len(late)], dtype=np.uint32)
            blk = np.concatenate([tabs.view(np.uint32), hdr, plist.view(np.uint32), noisy, pk.view(np.uint32)])
            image.append(blk)
            c.groups[i].dst = None                                  # identity
            c.groups[i].lds_off = lds_off
            c.groups[i].map_words = len(blk)
            lds_off += len(blk)
    # who writes which row: role 0 (dynamics waves, free once their reward phase is done) or role 1 (reference / observation waves).  Greedy by
    # width, role 0 handicapped by the work of its reward / reset phases; a row that reads future targets — produced by role 1 while role 0
    # already writes — stays with role 1.
    fut_lo = min([feat_off[n_] for n_ in feat_off if n_.startswith("FUT_")] or [1 << 30])
    fut_hi = max([feat_off[n_] + fdim[n_] for n_ in feat_off if n_.startswith("FUT_")] or [-1])
    # Round 4: v1 hands EVERY row to role 1 — with them the history block leaves the LDS feature row (step_lds_plan: a fifth workgroup per
    # CU), and role 0, the chain that sets a workgroup's duration, ends with its reward / reset phases.
    load = [float(_os.environ.get("PBHC_ROLE0_HANDICAP", "0.1" if mode == 1 else "1e9")) * sum(len(m[2]) for m in maps), 0.0]    # (env var: measurement aid)
    for i in sorted(range(len(maps)), key=lambda i_: -len(maps[i_][2])):
        reads_future = any(fut_lo <= s_ < fut_hi for s_ in maps[i][2])
        r_ = 1 if (reads_future or load[1] <= load[0]) else 0
        c.groups[i].role = r_
        load[r_] += len(maps[i][2])
    L.group_roles = [int(c.groups[i].role) for i in range(len(maps))]
    # ... of which the dynamics waves take the share that balances the two roles after bar2 (they idle for ~2.3 k cycles after their reward /
    # reset phases while the reference waves write 1 010 elements): whole runs that read no history (that block is staged by the reference
    # waves), largest first, marked in PbhcObsRun.late bit 1.  The specialised kernel honours the marks in the builds whose history block
    # lives outside the feature row (step_lds_plan); every other build lets the reference waves write all runs.
    # Measured (profiles/round4_k_env_step_variants.txt (h)): shares of 0.15 / 0.23 / 0.32 give 18.3-18.6 us against 18.2 at 4096 envs and nothing
    # at 32 768 — the launch's tail is the chip-wide store drain, not the reference waves' instruction stream — so the default share is 0.
    share = float(_os.environ.get("PBHC_ROW_HELP_SHARE", "0"))
    if mode == 0 and share > 0.0 and all(r_ == 1 for r_ in L.group_roles) and all(c.groups[i].num_runs > 0 for i in range(len(maps))):
        hoff_ = feat_off["HISTORY"]
        cand = sorted(((int(c.groups[i].runs[r_].len), i, r_) for i in range(len(maps)) for r_ in range(c.groups[i].num_runs)
                       if c.groups[i].runs[r_].src + c.groups[i].runs[r_].len <= hoff_), reverse=True)
        budget = share * sum(len(m[2]) for m in maps)
        for n_, i, r_ in cand:
            if n_ <= budget:
                c.groups[i].runs[r_].late |= 2
                budget -= n_
    L.helper_elements = sum(int(c.groups[i].runs[r_].len) for i in range(len(maps)) for r_ in range(max(int(c.groups[i].num_runs), 0)) if c.groups[i].runs[r_].late & 2)
    c.map_lds_words = lds_off if compact else 0
    if compact:
        L.map_image = torch.from_numpy(np.concatenate(image).view(np.int32).copy()).to(device)
        assert L.map_image.numel() == lds_off
        c.map_image = L.map_image.data_ptr()
    c.clip_observations = float(ec.normalization.clip_observations)
    c.has_contact_mask = int(bool(motion_lib.has_contact_mask))
    if "teleop_contact_mask" in L.reward_names and not motion_lib.has_contact_mask:
        raise AttributeError("teleop_contact_mask reward needs a motion file with contact_mask")   # reference raises too (motion_tracking.py:1156)
    c.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    # ---- initial globals
    g = np.zeros(K["PBHC_NUM_GLOBALS"], dtype=np.float64)
    for i, k in enumerate(SIGMA_KEYS):
        v = float(rw.reward_tracking_sigma.get(k, 1.0)) if "reward_tracking_sigma" in rw else 1.0
        g[K["PBHC_G_SIGMA"] + i] = v
        g[K["PBHC_G_EMA"] + i] = v
    g[K["PBHC_G_PENALTY_SCALE"]] = float(rw.reward_initial_penalty_scale) if rw.reward_penalty_curriculum else 1.0
    g[K["PBHC_G_AVG_EP_LEN"]] = 0.0
    g[K["PBHC_G_MOTION_FAR_THR"]] = float(tc.terminate_when_motion_far_initial_threshold if (T.terminate_when_motion_far and tc.terminate_when_motion_far_curriculum)
                                           else ec.termination_scales.termination_motion_far_threshold)
    g[K["PBHC_G_SOFT_POS_VAL"]] = float(lc.soft_dof_pos_initial_limit)
    g[K["PBHC_G_SOFT_VEL_VAL"]] = float(lc.soft_dof_vel_initial_limit)
    g[K["PBHC_G_SOFT_TAU_VAL"]] = float(lc.soft_torque_initial_limit)
    g[K["PBHC_G_NOISE_CURRICULUM"]] = float(ob.noise_initial_value) if ob.get("add_noise_currculum", False) else 1.0
    if "noise_process" in ob and ob.noise_process.get("enable", False):
        raise NotImplementedError("obs.noise_process")
    L.globals0 = g
    return c, L
