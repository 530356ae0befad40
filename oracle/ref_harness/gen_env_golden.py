"""Golden traces of the reference's LeggedRobotMotionTracking.step (v1 / MHPPO env) on CPU.

Runs the unmodified reference env on the ReplayFakeSim for a few steps with scripted edge cases
and records, per step, every input the restatement needs (actions, replay frame, the torque-noise
uniforms, the values the reference sampled on resets) and every output / state it must reproduce.
"""
import os

import numpy as np
import torch

from oracle.ref_harness import gen_golden as G
from oracle.skeleton import parse_mjcf
from oracle.motion_lib import MotionLib
from pbhc_amd.utils import safe_pkl
from pbhc_amd.utils.config import load_config

FAKE = "oracle.ref_harness.fake_sim.ReplayFakeSim"


def make_cfg(path, N, motion_file=None, extra=None):
    ov = {"num_envs": N, "headless": True, "simulator._target_": FAKE}
    if motion_file:
        ov["robot.motion.motion_file"] = motion_file
    ov.update(extra or {})
    cfg = load_config(path, ov, now="golden")
    for k in list(cfg.obs.noise_scales.keys()):
        cfg.obs.noise_scales[k] = 0.0
    return cfg


def build_env(cfg, seed=0):
    from humanoidverse.utils.helpers import pre_process_config
    from humanoidverse.envs.motion_tracking.motion_tracking import LeggedRobotMotionTracking

    pre_process_config(cfg)
    torch.manual_seed(seed)
    np.random.seed(seed)
    return LeggedRobotMotionTracking(config=cfg.env.config, device="cpu")


def oracle_motion_lib(cfg):
    m = cfg.robot.motion
    skel = parse_mjcf(str(m.asset.assetRoot) + "/" + m.asset.assetFileName, [dict(e) for e in m.extend_config])
    clip = next(iter(safe_pkl.load(m.motion_file).values()))
    return skel, clip, MotionLib(skel, [clip])


def snapshot(env):
    """Every piece of persistent per-env / global state the step reads or writes."""
    s = env.simulator
    d = dict(
        root_states=s.robot_root_states, dof_pos=s.dof_pos, dof_vel=s.dof_vel, contact_forces=s.contact_forces,
        actions=env.actions, last_actions=env.last_actions, actions_after_delay=env.actions_after_delay,
        action_queue=env.action_queue, action_delay_idx=env.action_delay_idx,
        last_dof_pos=env.last_dof_pos, last_dof_vel=env.last_dof_vel, torques=env.torques,
        feet_air_time=env.feet_air_time, contacts=env.contacts, contacts_filt=env.contacts_filt,
        last_contacts=env.last_contacts.float(), last_contacts_filt=env.last_contacts_filt.float(),
        kp_scale=env._kp_scale, kd_scale=env._kd_scale, rfi_lim_scale=env._rfi_lim_scale, rao_scale=env._rao_scale,
        episode_length_buf=env.episode_length_buf, last_episode_length_buf=env.last_episode_length_buf,
        motion_start_times=env.motion_start_times, motion_len=env.motion_len, end_time_ratio_buf=env.end_time_ratio_buf,
        reset_buf=env.reset_buf, time_out_buf=env.time_out_buf,
        reward_penalty_scale=np.float64(getattr(env, "reward_penalty_scale", env.config.rewards.reward_initial_penalty_scale)), average_episode_length=np.float64(float(env.average_episode_length)),
        motion_far_threshold=np.float64(env.terminate_when_motion_far_threshold), common_step_counter=np.int64(env.common_step_counter),
    )
    for k, v in env.episode_sums.items():
        d["sum__" + k] = v
    for k, v in env.history_handler.history.items():
        d["hist__" + k] = v
    if hasattr(env, "_reward_error_ema"):
        for k, v in env._reward_error_ema.items():
            d["ema__" + k] = np.float64(v)
    for k, v in env.config.rewards.reward_tracking_sigma.items():
        d["sigma__" + k] = np.float64(v)
    return {k: G.T(v).copy() for k, v in d.items()}


def make_replay(env, ml, T, seed, script=True):
    """state_k = ref((ep_len0 + k + 1) dt + start0) + noise, plus scripted edge cases."""
    g = torch.Generator().manual_seed(seed)
    N, D, B = env.num_envs, env.num_dof, env.num_bodies
    root = torch.zeros(T, N, 13)
    qp = torch.zeros(T, N, D)
    qv = torch.zeros(T, N, D)
    cf = torch.zeros(T, N, B, 3)
    feet = env.feet_indices
    for k in range(T):
        t = (env.episode_length_buf + k + 1).float() * env.dt + env.motion_start_times
        ref = ml.get_motion_state(torch.zeros(N, dtype=torch.long), t, offset=env.env_origins)
        root[k, :, 0:3] = ref["root_pos"] + 0.02 * torch.randn(N, 3, generator=g)
        dq = torch.randn(N, 4, generator=g) * 0.02
        dq[:, 3] = 1.0
        q = ref["root_rot"] + dq * 0.0
        from oracle import rotations as R
        small = R.normalize(torch.cat([0.02 * torch.randn(N, 3, generator=g), torch.ones(N, 1)], -1))
        root[k, :, 3:7] = R.normalize(R.quat_mul(small, q))
        root[k, :, 7:10] = ref["root_vel"] + 0.1 * torch.randn(N, 3, generator=g)
        root[k, :, 10:13] = ref["root_ang_vel"] + 0.1 * torch.randn(N, 3, generator=g)
        qp[k] = ref["dof_pos"] + 0.02 * torch.randn(N, D, generator=g)
        qv[k] = ref["dof_vel"] + 0.1 * torch.randn(N, D, generator=g)
        if "contact_mask" in ref:
            on = (ref["contact_mask"] > 0.5).float()
        else:
            on = (ref["rg_pos_t"][:, feet, 2] < 0.06).float()
        on = torch.where(torch.rand(N, 2, generator=g) < 0.15, 1 - on, on)      # flicker -> air-time logic
        cf[k, :, feet, 2] = on * (300.0 + 50.0 * torch.randn(N, 2, generator=g))
        cf[k, :, feet, 0:2] = on.unsqueeze(-1) * 20.0 * torch.randn(N, 2, 2, generator=g)
    if script and N >= 12:
        small_tilt = lambda a: torch.tensor([np.sin(a / 2), 0.0, 0.0, np.cos(a / 2)], dtype=torch.float)
        root[2:, 3, 3:7] = R.quat_mul(small_tilt(1.2).expand(T - 2, 4), root[2:, 3, 3:7])   # gravity termination at step 2+
        root[3, 4, 0:3] += torch.tensor([2.5, 0.0, 0.0])                                     # motion_far at step 3
        cf[1:4, 5, env.penalised_contact_indices[:3], :] = 5.0                               # collision
        cf[2, 5, feet[0]] = torch.tensor([400.0, 0.0, 30.0])                                  # stumble + big force
        cf[4, 6, feet[1]] = torch.tensor([0.0, 0.0, 900.0])                                   # contact force penalty
        qp[1:5, 6, 3] = 3.1                                                                   # beyond soft+hard limit (knee)
        qp[1:5, 6, 5] = -0.5
        qv[2, 7, :] = 40.0                                                                    # beyond vel limits
        qv[3, 7, 2] = -33.0
    return root, qp, qv, cf


def run_trace(cfg_path, tag, N, T, motion_file=None, extra=None, seed=0):
    cfg = make_cfg(cfg_path, N, motion_file, extra)
    env = build_env(cfg, seed)
    skel, clip, ml = oracle_motion_lib(cfg)
    D = env.num_dof
    torch.manual_seed(seed + 1)
    env.reset_all()
    # ---- scripted initial conditions (public attributes of the reference env) ----------------
    if N >= 12:
        L = float(env.motion_len[0])
        env.motion_start_times[1] = L - 3.5 * env.dt          # motion end -> time-out
        env.episode_length_buf[2] = int(env.max_episode_length) - 2
        env.motion_start_times[2] = 0.0
        env.motion_start_times[8] = 0.0                       # t = exactly on frames / blend edge
        env.episode_length_buf[8] = 0
        env.action_delay_idx[9] = 2
        env.action_delay_idx[10] = 0
    env._kick_motion_res_counter = -1
    root, qp, qv, cf = make_replay(env, ml, T, seed + 2)
    env.simulator.set_replay(root, qp, qv, cf, start_frame=0)
    g = torch.Generator().manual_seed(seed + 3)
    actions = 0.6 * torch.randn(T, N, D, generator=g)
    actions[1, 0, 0] = 250.0
    actions[2, 0, 1] = -250.0

    # record the uniforms of the torque RFI noise without touching the reference: reseed, draw, reseed
    u_rfi = []
    orig_ct = env._compute_torques
    call = {"n": 0}

    def wrapped(a):
        s = 1000 + call["n"]
        call["n"] += 1
        torch.manual_seed(s)
        u = torch.rand(N, D)
        torch.manual_seed(s)
        u_rfi.append(u)
        return orig_ct(a)

    env._compute_torques = wrapped

    out = {"state0__" + k: v for k, v in snapshot(env).items()}
    out.update(
        env_origins=G.T(env.env_origins), default_dof_pos=G.T(env.default_dof_pos[0]), ref_init_yaw=G.T(env.ref_init_rpy[0, 2]),
        base_com_bias=G.T(env.simulator._base_com_bias), link_mass_scale=G.T(env.simulator._link_mass_scale),
        friction_coeffs=G.T(env.simulator.friction_coeffs), p_gains=G.T(env.p_gains), d_gains=G.T(env.d_gains),
        reward_names=np.array(env.reward_names), feet_indices=G.T(env.feet_indices),
        penalised_contact_indices=G.T(env.penalised_contact_indices), dt=np.float64(env.dt),
        max_episode_length=np.float64(env.max_episode_length),
        replay_root=G.T(root), replay_dof_pos=G.T(qp), replay_dof_vel=G.T(qv), replay_contact=G.T(cf), actions_in=G.T(actions),
        clip_pose_aa=clip["pose_aa"], clip_root_trans_offset=clip["root_trans_offset"], clip_fps=np.int64(clip["fps"]),
    )
    if "contact_mask" in clip:
        out["clip_contact_mask"] = clip["contact_mask"]
    per = {}

    def rec(k, v):
        per.setdefault(k, []).append(G.T(v).copy())

    for k in range(T):
        torch.manual_seed(seed + 100 + k)
        obs, rew, reset, extras = env.step({"actions": actions[k].clone()})
        rec("u_rfi", u_rfi[-1])
        for ok, ov in obs.items():
            rec("obs__" + ok, ov)
        rec("rew_buf", rew); rec("reset_buf_out", reset); rec("time_outs", extras["time_outs"] if "time_outs" in extras else env.time_out_buf)
        rec("ref_body_pos_extend", extras["ref_body_pos_extend"]); rec("ref_body_rot_extend", extras["ref_body_rot_extend"])
        for name in ["dif_global_body_pos", "dif_global_body_rot", "dif_global_body_vel", "dif_global_body_ang_vel", "dif_joint_angles",
                     "dif_joint_velocities", "_rigid_body_pos_extend", "_rigid_body_rot_extend", "_rigid_body_vel_extend",
                     "_rigid_body_ang_vel_extend", "base_lin_vel", "base_ang_vel", "projected_gravity", "rpy",
                     "_obs_dif_local_rigid_body_pos", "_obs_local_ref_rigid_body_pos", "_obs_vr_3point_pos", "_ref_motion_phase", "relyaw"]:
            rec("x__" + name, getattr(env, name))
        for lk, lv in env.log_dict.items():
            rec("log__" + lk, torch.as_tensor(lv, dtype=torch.float32))
        for sk, sv in snapshot(env).items():
            rec("state__" + sk, sv)
    for k, v in per.items():
        out["step__" + k] = np.stack(v)
    G.save(f"env_v1_{tag}.npz", **out)
    return cfg


def dump_fixture_config(cfg_path, name, motion_file=None, extra=None):
    """A pruned copy of a composed reference config (data, the reference's config schema) for use
    on the GPU box where /root/reference does not exist."""
    import yaml
    from pbhc_amd.utils.config import load_unresolved, set_by_path

    c = load_unresolved(cfg_path)
    for k, v in (extra or {}).items():
        set_by_path(c, k, v)
    if motion_file:
        c["robot"]["motion"]["motion_file"] = motion_file
    m = c["robot"]["motion"]
    # data fixtures that travel with the repo: the clip (arrays only) and the parsed skeleton tables
    from pbhc_amd.motion_lib import load_motion_file, save_motion_npz
    from pbhc_amd.skeleton import Skeleton
    src = m["motion_file"]
    clip_name = os.path.splitext(os.path.basename(src))[0] + ".npz"
    os.makedirs(os.path.join(G.GOLD, "clips"), exist_ok=True)
    save_motion_npz(os.path.join(G.GOLD, "clips", clip_name), [("clip0", cc) for _, cc in load_motion_file(src)])
    sk = Skeleton.from_mjcf(os.path.join(m["asset"]["assetRoot"], m["asset"]["assetFileName"]), [dict(e) for e in m["extend_config"]])
    sk_name = "skeleton_" + os.path.splitext(m["asset"]["assetFileName"])[0] + ".json"
    sk.to_json(os.path.join(G.GOLD, sk_name))
    m["motion_file"] = "tests/golden/clips/" + clip_name
    m["asset"]["assetRoot"] = "tests/golden"
    m["asset"]["assetFileName"] = sk_name
    for k in ["visualization", "smpl_pose_modifier", "joint_matches", "limb_weight_group"]:
        m.pop(k, None)
    c.pop("eval_overrides", None)

    def plain(n):
        if isinstance(n, dict):
            return {k: plain(v) for k, v in n.items()}
        if isinstance(n, list):
            return [plain(v) for v in n]
        return n

    os.makedirs(os.path.join(G.GOLD, "configs"), exist_ok=True)
    with open(os.path.join(G.GOLD, "configs", name), "w") as f:
        yaml.safe_dump(plain(c), f, sort_keys=False, default_flow_style=None, width=160)
    print("wrote config fixture", name)


V1_CFG = "example/pretrained_horse_stance_pose/config.yaml"
WALK_EXTRA = {"rewards.reward_scales.teleop_contact_mask": 0}


def main(traces=True):
    if traces:
        run_trace(V1_CFG, "horse", N=32, T=12)
        run_trace(V1_CFG, "walk", N=16, T=8, motion_file="motion_data/g1_walk_45cms_23dof.pkl", extra=WALK_EXTRA, seed=7)
    dump_fixture_config(V1_CFG, "v1_g1_23dof_horse_stance.yaml")
    dump_fixture_config(V1_CFG, "v1_g1_23dof_walk.yaml", motion_file="motion_data/g1_walk_45cms_23dof.pkl", extra=WALK_EXTRA)
