"""What the REFERENCE derives from its config trees — observation dims / slices, dt, the scaled reward table, gains, limits, body index
lists, history shapes — recorded for the five composed configs of tests/golden/configs, with the reference's config resolved by
`oracle/ref_harness/ref_config.py` (an implementation independent of the product's resolver) and the numbers read off the live reference
env object (`pre_process_config` helpers.py:82-126, `_prepare_reward_function` legged_robot_base.py:167-233, `_init_buffers` :39-110,
`_init_domain_rand_buffers`, motion-tracking body lists motion_tracking.py:98-170).

    python oracle/ref_harness/gen_config_golden.py      (build container only; writes tests/golden/config_derived.json)

tests/test_config_derived.py then checks that the product's `load_config` + `envs/env_config.build` reproduce them from the fixture yamls.
"""
import copy
import json
import os

import numpy as np
import torch
import yaml

from oracle.ref_harness import gen_golden as G          # noqa: F401  (sys.path, shims, cwd = /root/reference)
from oracle.ref_harness import ref_config as RC

REF = "/root/reference"
FAKE = "oracle.ref_harness.fake_sim.ReplayFakeSim"
V1_CFG = "example/pretrained_horse_stance_pose/config.yaml"
STUDENT_CFG = "logs/MotionTracking/phuma_student/config.yaml"
CLIP29 = "motion_data/g1_rig_Skeleton_Sequence_converted_processed_g1_29dof_rev_1_0.pkl"
COMMON = {"headless": True, "simulator._target_": FAKE, "num_envs": 8}
V2_COMMON = dict(COMMON, **{"domain_rand.push_robots": False, "algo.config.teacher_model_path": None, "algo.config.dagger_only": False})


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = copy.deepcopy(v)
    return a


def _cfg_yaml(rel):
    with open(os.path.join(REF, "humanoidverse/config", rel)) as f:
        return yaml.safe_load(f)


def _tree(variant):
    """the unresolved tree of each fixture, from the files the reference ships (the Hydra defaults-list merge of the two teacher variants
    done by hand, as oracle/ref_harness/gen_env_v2_golden.py / gen_distill_golden.py do)"""
    if variant in ("horse", "walk"):
        with open(os.path.join(REF, V1_CFG)) as f:
            return yaml.safe_load(f)
    with open(os.path.join(REF, STUDENT_CFG)) as f:
        c = yaml.safe_load(f)
    if variant == "teacher29":
        c["robot"] = _merge(_cfg_yaml("robot/robot_base.yaml")["robot"], _cfg_yaml("robot/g1/g1_29dof_general.yaml")["robot"])
        c["obs"] = _cfg_yaml("obs/motion_tracking/obs_ppo_teacher.yaml")["obs"]
        c["robot"]["motion"]["motion_file"] = CLIP29
    elif variant == "teacher23":
        c["obs"] = _cfg_yaml("obs/motion_tracking/obs_ppo_teacher.yaml")["obs"]
    return c


CASES = {
    "v1_g1_23dof_horse_stance.yaml": ("horse", COMMON),
    "v1_g1_23dof_walk.yaml": ("walk", dict(COMMON, **{"robot.motion.motion_file": "motion_data/g1_walk_45cms_23dof.pkl", "rewards.reward_scales.teleop_contact_mask": 0})),
    "v2_g1_23dof_student.yaml": ("student23", V2_COMMON),
    "v2_g1_23dof_teacher.yaml": ("teacher23", V2_COMMON),
    "v2_g1_29dof_teacher.yaml": ("teacher29", V2_COMMON),
}


def _resolved(variant, overrides):
    tmp = os.path.join(G.REPO, "oracle", "_ref", f"cfg_{variant}.yaml")
    os.makedirs(os.path.dirname(tmp), exist_ok=True)
    with open(tmp, "w") as f:
        yaml.safe_dump(_tree(variant), f, sort_keys=False)
    return RC.load(tmp, overrides, now="golden")


def _list(x):
    if torch.is_tensor(x):
        return x.detach().cpu().double().numpy().tolist()
    if isinstance(x, np.ndarray):
        return x.astype(np.float64).tolist()
    if isinstance(x, (list, tuple)):
        return [_list(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    return x


def derive(variant, overrides):
    from humanoidverse.utils.helpers import pre_process_config

    cfg = _resolved(variant, overrides)
    pre_process_config(cfg)
    torch.manual_seed(0)
    np.random.seed(0)
    if variant in ("horse", "walk"):
        from humanoidverse.envs.motion_tracking.motion_tracking import LeggedRobotMotionTracking as Env
    else:
        from humanoidverse.envs.motion_tracking.general_tracking import LeggedRobotGeneralTracking as Env
    env = Env(config=cfg.env.config, device="cpu")
    ec = env.config
    out = dict(
        algo_obs_dim_dict={k: int(v) for k, v in ec.robot.algo_obs_dim_dict.items()},
        obs_dims={k: int(v) for k, v in ec.obs.obs_dims.items()},
        obs_slices={g: {k: [int(a), int(b)] for k, (a, b) in d.items()} for g, d in ec.obs.post_compute_config["obs_slices"].items()},
        dt=float(env.dt), max_episode_length=float(env.max_episode_length), max_episode_length_s=float(env.max_episode_length_s),
        num_dof=int(env.num_dof), num_bodies=int(env.num_bodies), dim_actions=int(env.dim_actions),
        reward_scales_dt={k: float(v) for k, v in env.reward_scales.items()},          # zero scales dropped, x dt (legged_robot_base.py:173-181)
        reward_scale_order=list(env.reward_scales.keys()),                             # (the file is written with sorted keys)
        reward_names=list(env.reward_names),
        use_vec_reward=bool(ec.get("use_vec_reward", False)),
        p_gains=_list(env.p_gains), d_gains=_list(env.d_gains), default_dof_pos=_list(env.default_dof_pos[0]),
        torque_limits=_list(env.torque_limits), dof_vel_limits=_list(env.dof_vel_limits), dof_pos_limits=_list(env.dof_pos_limits),
        # legged_robot_base.py:805-808: a scalar scales every joint, a per-joint table goes through self.action_scales (:99-100)
        action_scale=(_list(env.action_scales) if not isinstance(ec.robot.control.action_scale, (int, float))
                      else [float(ec.robot.control.action_scale)] * int(env.dim_actions)),
        action_clip_value=float(ec.robot.control.action_clip_value),
        feet_indices=_list(env.feet_indices), penalised_contact_indices=_list(env.penalised_contact_indices),
        termination_contact_indices=_list(env.termination_contact_indices),
        upper_body_id=_list(env.upper_body_id), lower_body_id=_list(env.lower_body_id), motion_tracking_id=_list(env.motion_tracking_id),
        history={k: list(v.shape[1:]) for k, v in env.history_handler.history.items()},
        num_extend_bodies=int(len(ec.robot.motion.extend_config)),
    )
    if hasattr(env, "key_body_id"):
        out["key_body_id"] = _list(env.key_body_id)
    if hasattr(env, "tar_obs_steps"):
        out["tar_obs_steps"] = _list(env.tar_obs_steps)
    if hasattr(env, "anchor_index"):
        out["anchor_index"] = int(env.anchor_index)
    return out


def main():
    rec = {}
    for name, (variant, ov) in CASES.items():
        rec[name] = derive(variant, ov)
        print(name, "actor/critic dims", rec[name]["algo_obs_dim_dict"], "terms", len(rec[name]["reward_scales_dt"]))
    path = os.path.join(G.GOLD, "config_derived.json")
    with open(path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
