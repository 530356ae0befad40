"""Reference traces for config switches no shipped yaml enables (small: 8 envs x 4 steps of the unmodified reference's
LeggedRobotMotionTracking.step on the walk clip, recorded exactly like gen_env_golden.py's traces):
  env_v1_walk_ctrlV.npz / env_v1_walk_ctrlT.npz    robot.control.control_type "V" / "T" (legged_robot_base.py:809-817)

    PYTHONPATH=/root/repo python oracle/ref_harness/gen_switch_golden.py      (build container only)
"""
from oracle.ref_harness import gen_env_golden as G1

WALK = "motion_data/g1_walk_45cms_23dof.pkl"


def main():
    for tag, ct in (("walk_ctrlV", "V"), ("walk_ctrlT", "T")):
        G1.run_trace(G1.V1_CFG, tag, N=8, T=4, motion_file=WALK, extra=dict(G1.WALK_EXTRA, **{"robot.control.control_type": ct}), seed=21)


if __name__ == "__main__":
    main()
