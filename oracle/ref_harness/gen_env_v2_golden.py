"""Golden traces of the reference's LeggedRobotGeneralTracking.step (v2 / KungfuBot2 env) on CPU.

Two configurations, both built from files the reference ships:
  student23 : logs/MotionTracking/phuma_student/config.yaml as composed by the reference (23 DoF, 877-dim actor obs)
  teacher29 : the same tree with the `obs` node of config/obs/motion_tracking/obs_ppo_teacher.yaml and the `robot` node of
              config/robot/robot_base.yaml + robot/g1/g1_29dof_general.yaml (Hydra's defaults-list merge done by hand) and the
              shipped 29-DoF clip
Same recording scheme as gen_env_golden.py (inputs the restatement needs + every output/state it must reproduce).
"""
import copy
import os

import numpy as np
import torch
import yaml

from oracle.ref_harness import gen_golden as G
from oracle.ref_harness import gen_env_golden as G1
from oracle import rotations as R
from pbhc_amd.utils.config import load_unresolved, resolve, set_by_path, _wrap

REF = "/root/reference"
STUDENT_CFG = "logs/MotionTracking/phuma_student/config.yaml"
CLIP29 = "motion_data/g1_rig_Skeleton_Sequence_converted_processed_g1_29dof_rev_1_0.pkl"


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = copy.deepcopy(v)
    return a


def _yaml(rel):
    with open(os.path.join(REF, "humanoidverse/config", rel)) as f:
        return yaml.safe_load(f)


def unresolved_tree(variant):
    c = load_unresolved(os.path.join(REF, STUDENT_CFG))
    if variant == "teacher29":
        robot = _merge(_yaml("robot/robot_base.yaml")["robot"], _yaml("robot/g1/g1_29dof_general.yaml")["robot"])
        c["robot"] = _wrap(robot)
        c["obs"] = _wrap(_yaml("obs/motion_tracking/obs_ppo_teacher.yaml")["obs"])
        c["robot"]["motion"]["motion_file"] = CLIP29
    return c


COMMON = {
    "headless": True, "simulator._target_": G1.FAKE, "domain_rand.push_robots": False,
    "algo.config.teacher_model_path": None, "algo.config.dagger_only": False,
}


def make_cfg(variant, N, extra=None):
    c = unresolved_tree(variant)
    ov = dict(COMMON, num_envs=N)
    ov.update(extra or {})
    for k, v in ov.items():
        set_by_path(c, k, v)
    cfg = resolve(c, now="golden")
    for k in list(cfg.obs.noise_scales.keys()):
        cfg.obs.noise_scales[k] = 0.0
    return cfg


def build_env(cfg, seed=0):
    from humanoidverse.utils.helpers import pre_process_config
    from humanoidverse.envs.motion_tracking.general_tracking import LeggedRobotGeneralTracking

    pre_process_config(cfg)
    torch.manual_seed(seed)
    np.random.seed(seed)
    return LeggedRobotGeneralTracking(config=cfg.env.config, device="cpu")


X_NAMES = ["dif_global_body_pos", "dif_global_body_rot", "dif_global_body_vel", "dif_global_body_ang_vel", "dif_joint_angles",
           "dif_joint_velocities", "dif_root_velocity", "dif_root_rot", "dif_root_height", "_rigid_body_pos_extend", "_rigid_body_rot_extend",
           "_rigid_body_vel_extend", "_rigid_body_ang_vel_extend", "base_lin_vel", "base_ang_vel", "projected_gravity", "rpy",
           "_obs_dif_local_key_body_pos", "_obs_local_ref_key_body_pos", "_ref_motion_phase", "body_pos_relative_w", "body_quat_relative_w",
           "dif_local_body_pos", "dif_local_body_rot", "_obs_local_body_rot", "_obs_local_body_pos", "_obs_anchor_ref_rot", "_obs_anchor_ref_pos",
           "dif_anchor_body_pos", "dif_anchor_pos_z", "dif_anchor_ori", "obs_next_step_mimic_buf", "obs_future_motion_dof_pos",
           "obs_future_motion_local_ref_key_body_pos", "obs_future_motion_base_ang_vel"]


def run_trace(variant, tag, N, T, extra=None, seed=0):
    cfg = make_cfg(variant, N, extra)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        env = build_env(cfg, seed)
        skel, clip, ml = G1.oracle_motion_lib(cfg)
    finally:
        os.chdir(cwd)
    D = env.num_dof
    torch.manual_seed(seed + 1)
    env.reset_all()
    if N >= 12:
        L = float(env.motion_len[0])
        env.motion_start_times[1] = L - 3.5 * env.dt          # motion end -> time-out
        env.motion_start_times[2] = L - 40.5 * env.dt         # future targets run past the clip end -> clamped frames
        env.motion_start_times[8] = 0.0                       # t exactly on frames
        env.episode_length_buf[8] = 0
        env.action_delay_idx[9] = 2
        env.action_delay_idx[10] = 0
    env._kick_motion_res_counter = -1
    root, qp, qv, cf = G1.make_replay(env, ml, T, seed + 2, script=False)
    if N >= 12:
        feet = env.feet_indices
        tilt = lambda a: torch.tensor([0.0, np.sin(a / 2), 0.0, np.cos(a / 2)], dtype=torch.float)
        root[2:, 3, 3:7] = R.quat_mul(tilt(1.7).expand(T - 2, 4), root[2:, 3, 3:7])       # ref_ori termination from step 2
        root[3, 4, 2] += 0.4                                                              # ref_pos_z termination at step 3
        qp[4, 5, 3] += 1.4                                                                # knee bend: ankle z off -> body_z termination
        root[1:3, 6, 3:7] = R.quat_mul(torch.tensor([0.0, 0.0, np.sin(0.9), np.cos(0.9)], dtype=torch.float).expand(2, 4), root[1:3, 6, 3:7])   # big yaw error only: no termination
        cf[1:4, 5, env.penalised_contact_indices[:3], :] = 5.0
        qp[1:5, 7, 3] = 3.1
        qv[2, 7, :] = 40.0
    env.simulator.set_replay(root, qp, qv, cf, start_frame=0)
    g = torch.Generator().manual_seed(seed + 3)
    actions = 0.6 * torch.randn(T, N, D, generator=g)
    actions[1, 0, 0] = 250.0
    actions[2, 0, 1] = -250.0

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
    out = {"state0__" + k: v for k, v in G1.snapshot(env).items()}
    out.update(
        env_origins=G.T(env.env_origins), default_dof_pos=G.T(env.default_dof_pos[0]), ref_init_yaw=G.T(env.ref_init_rpy[0, 2]),
        base_com_bias=G.T(env.simulator._base_com_bias), link_mass_scale=G.T(env.simulator._link_mass_scale), base_mass_scale=G.T(env.simulator._base_mass_scale),
        friction_coeffs=G.T(env.simulator.friction_coeffs), p_gains=G.T(env.p_gains), d_gains=G.T(env.d_gains),
        reward_names=np.array(env.reward_names), feet_indices=G.T(env.feet_indices), key_body_id=np.array(env.key_body_id), anchor_index=np.int64(env.anchor_index),
        penalised_contact_indices=G.T(env.penalised_contact_indices), dt=np.float64(env.dt), max_episode_length=np.float64(env.max_episode_length),
        replay_root=G.T(root), replay_dof_pos=G.T(qp), replay_dof_vel=G.T(qv), replay_contact=G.T(cf), actions_in=G.T(actions),
        clip_pose_aa=clip["pose_aa"], clip_root_trans_offset=clip["root_trans_offset"], clip_fps=np.int64(clip["fps"]),
    )
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
        for name in X_NAMES:
            rec("x__" + name, getattr(env, name))
        for lk, lv in env.log_dict.items():
            rec("log__" + lk, torch.as_tensor(lv, dtype=torch.float32))
        for sk, sv in G1.snapshot(env).items():
            rec("state__" + sk, sv)
    for k, v in per.items():
        out["step__" + k] = np.stack(v)
    G.save(f"env_v2_{tag}.npz", **out)
    return cfg


def dump_fixture_config(variant, name, extra=None):
    """Composed v2 config (data, reference schema) + clip arrays + skeleton tables for the GPU box."""
    from pbhc_amd.motion_lib import load_motion_file, save_motion_npz
    from pbhc_amd.skeleton import Skeleton

    dump_fixture_tree(unresolved_tree(variant), name, extra)


def dump_fixture_tree(c, name, extra=None):
    from pbhc_amd.motion_lib import load_motion_file, save_motion_npz
    from pbhc_amd.skeleton import Skeleton

    for k, v in dict(COMMON, **(extra or {})).items():
        if k != "simulator._target_":
            set_by_path(c, k, v)
    m = c["robot"]["motion"]
    src = os.path.join(REF, m["motion_file"])
    clip_name = os.path.splitext(os.path.basename(src))[0] + ".npz"
    os.makedirs(os.path.join(G.GOLD, "clips"), exist_ok=True)
    save_motion_npz(os.path.join(G.GOLD, "clips", clip_name), [("clip0", cc) for _, cc in load_motion_file(src)])
    sk = Skeleton.from_mjcf(os.path.join(REF, m["asset"]["assetRoot"], m["asset"]["assetFileName"]), [dict(e) for e in m["extend_config"]])
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

    with open(os.path.join(G.GOLD, "configs", name), "w") as f:
        yaml.safe_dump(plain(c), f, sort_keys=False, default_flow_style=None, width=160)
    print("wrote config fixture", name)


def main(traces=True):
    if traces:
        run_trace("student23", "student23", N=16, T=8, seed=11)
        run_trace("teacher29", "teacher29", N=16, T=8, seed=12)
    dump_fixture_config("student23", "v2_g1_23dof_student.yaml")
    dump_fixture_config("teacher29", "v2_g1_29dof_teacher.yaml")


if __name__ == "__main__":
    main()
