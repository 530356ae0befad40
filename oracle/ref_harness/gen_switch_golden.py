"""Reference traces for config switches no shipped yaml enables (small: 8 envs x 4 steps of the unmodified reference's
LeggedRobotMotionTracking.step on the walk clip, recorded exactly like gen_env_golden.py's traces):
  env_v1_walk_ctrlV.npz / env_v1_walk_ctrlT.npz    robot.control.control_type "V" / "T" (legged_robot_base.py:809-817)
  env_v1_walk_feetori.npz                          reward terms feet_heading_alignment(_contact), penalty_feet_ori(_contact) (:1030-1079) weighted
  env_v1_walk_softlim.npz (16 envs x 5 steps)      rewards.reward_limit.reward_limits_curriculum with min != max (:902-939): the three soft-limit
                                                   fractions widen at every step that resets an env (two scripted time-outs), the limit
                                                   penalties of the following steps are computed against the moved values

  env_v1_walk_termnoise.npz (16 envs x 5 steps)    termination.terminate_by_contact / terminate_by_low_height (:434-447; 5 N on terminating bodies
                                                   of one env, a base-height floor of 0.772 m) and obs.add_noise_currculum (:591-592,637-646: the
                                                   multiplier moves at every step that resets an env; noise scales are zero in the traces, the
                                                   logged multiplier is what is pinned)

    PYTHONPATH=/root/repo python oracle/ref_harness/gen_switch_golden.py      (build container only)
"""
from oracle.ref_harness import gen_env_golden as G1

WALK = "motion_data/g1_walk_45cms_23dof.pkl"


FEET_ORI = {"rewards.reward_scales.feet_heading_alignment": -0.5, "rewards.reward_scales.feet_heading_alignment_contact": -0.3,
            "rewards.reward_scales.penalty_feet_ori": -0.2, "rewards.reward_scales.penalty_feet_ori_contact": -0.4}


_LC = "rewards.reward_limit.reward_limits_curriculum."
SOFT_LIMITS = {_LC + "soft_dof_pos_curriculum": True, _LC + "soft_dof_vel_curriculum": True, _LC + "soft_torque_curriculum": True}
for _pre, _init, _lo, _hi, _deg in (("soft_dof_pos", 0.5, 0.4, 0.56, 0.05), ("soft_dof_vel", 0.3, 0.2, 0.9, 0.1), ("soft_torque", 0.1, 0.05, 0.9, 0.2)):
    SOFT_LIMITS.update({_LC + _pre + "_initial_limit": _init, _LC + _pre + "_min_limit": _lo, _LC + _pre + "_max_limit": _hi, _LC + _pre + "_curriculum_degree": _deg,
                        _LC + _pre + "_curriculum_level_down_threshold": 40, _LC + _pre + "_curriculum_level_up_threshold": 42})


TERM_NOISE = {"env.config.termination.terminate_by_contact": True, "env.config.termination.terminate_by_low_height": True,
              "env.config.termination_scales.termination_min_base_height": 0.772, "obs.add_noise_currculum": True, "obs.soft_dof_pos_curriculum_degree": 0.1}


def main(which=("walk_ctrlV", "walk_ctrlT", "walk_feetori", "walk_softlim", "walk_termnoise")):
    for tag, ct in (("walk_ctrlV", "V"), ("walk_ctrlT", "T")):
        if tag in which:
            G1.run_trace(G1.V1_CFG, tag, N=8, T=4, motion_file=WALK, extra=dict(G1.WALK_EXTRA, **{"robot.control.control_type": ct}), seed=21)
    if "walk_feetori" in which:
        G1.run_trace(G1.V1_CFG, "walk_feetori", N=8, T=4, motion_file=WALK, extra=dict(G1.WALK_EXTRA, **FEET_ORI), seed=22)
    if "walk_termnoise" in which:
        G1.run_trace(G1.V1_CFG, "walk_termnoise", N=16, T=5, motion_file=WALK, extra=dict(G1.WALK_EXTRA, **TERM_NOISE), seed=24)
    if "walk_softlim" in which:
        G1.run_trace(G1.V1_CFG, "walk_softlim", N=16, T=5, motion_file=WALK, extra=dict(G1.WALK_EXTRA, **SOFT_LIMITS), seed=23)


if __name__ == "__main__":
    import sys

    main(tuple(sys.argv[1:]) or ("walk_ctrlV", "walk_ctrlT", "walk_feetori", "walk_softlim", "walk_termnoise"))
