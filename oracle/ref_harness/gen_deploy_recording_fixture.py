"""Fixture: a sim2sim rollout the reference recorded with its own deploy stack (MuJoCo robot + exported student policy).

Source (data the reference ships): logs/MotionTracking/phuma_student/motions/None_URCI_MujocoRobot_20260128_213834/
0_pid0_frame141_20260128_213841.pkl — written by deploy/urcirobot.py:655-699 (`TrySaveMotionStep` / `TrySaveMotionFile`) during a
run of motion_data/g1_ue_walk_23dof.pkl: per control step the 877-wide `actor_obs` the policy was fed (assembled by
deploy/urcirobot.py:326-374,864-949), the action it returned and the MuJoCo state.  The pickle is read with the static, non-executing
parser `pbhc_amd.utils.safe_pkl`; only arrays are kept.

Row r of the file holds: actor_obs built at the START of control step r (timer = r: previous action, reference motion at (r+1)*dt),
action a_r, and the robot state after a_r was applied — sampled one physics sub-step before the state the next observation is built
from (deploy/mujoco.py:480-523: GetState runs before each mj_step), which is why the tests rebuild the robot state from the recorded
observation itself rather than from these state columns.

A second source, logs/sim_to_sim_logs/sim_to_sim_log_20260130_205829.pkl (same deploy stack, same policy, clip
motion_data/g1_walk_45cms_23dof.pkl, 204 consecutive control steps from timer 0), records per step the robot state the observation WAS
built from (base quaternion, body-frame angular velocity, joint positions / velocities), the action, `actor_obs` (877),
`future_motion_targets` (600) and `prop_history` (740, identical to the history slice of actor_obs and therefore dropped here).

Run: python oracle/ref_harness/gen_deploy_recording_fixture.py  ->  tests/golden/deploy_student23_recording.npz, deploy_sim2sim_log_walk.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pbhc_amd.utils import safe_pkl  # noqa: E402

SRC = "/root/reference/logs/MotionTracking/phuma_student/motions/None_URCI_MujocoRobot_20260128_213834/0_pid0_frame141_20260128_213841.pkl"
SRC2 = "/root/reference/logs/sim_to_sim_logs/sim_to_sim_log_20260130_205829.pkl"
KEEP2 = ["robot_quat_xyzw", "robot_ang_vel", "robot_dof_pos", "robot_dof_vel", "action", "actor_obs", "future_motion_targets"]
KEEP = ["actor_obs", "action", "root_trans_offset", "root_rot", "root_lin_vel", "root_ang_vel", "dof", "dof_vel", "motion_times"]

if __name__ == "__main__":
    rec = list(safe_pkl.load(SRC).values())[0]
    out = {k: np.asarray(rec[k], dtype=np.float32) for k in KEEP}
    out["fps"] = np.float32(rec["fps"])
    dst = os.path.join(ROOT, "tests", "golden", "deploy_student23_recording.npz")
    np.savez_compressed(dst, **out)
    print(dst, {k: v.shape for k, v in out.items()}, os.path.getsize(dst))

    d = safe_pkl.load(SRC2)
    logs = d["logs"]
    assert [r["timer"] for r in logs] == list(range(len(logs)))
    out = {k: np.stack([np.asarray(r[k], dtype=np.float32).reshape(-1) for r in logs]) for k in KEEP2}
    hist = np.stack([np.asarray(r["prop_history"], dtype=np.float32).reshape(-1) for r in logs])
    assert np.array_equal(hist, out["actor_obs"][:, 78:818])
    out["motion_time"] = np.asarray([r["motion_time"] for r in logs], dtype=np.float64)
    out["dt"] = np.float64(d["config"]["dt"])
    dst = os.path.join(ROOT, "tests", "golden", "deploy_sim2sim_log_walk.npz")
    np.savez_compressed(dst, **out)
    print(dst, {k: v.shape for k, v in out.items()}, os.path.getsize(dst))
