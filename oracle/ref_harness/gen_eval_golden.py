"""Golden for the offline evaluation metrics: the reference's own measure_traj.py functions (unmodified, imported from
/root/reference) run on a rollout the reference's deploy stack RECORDED (logs/MotionTracking/phuma_student/motions/.../0_pid0_frame714_...pkl:
MuJoCo robot + exported student policy on g1_ue_walk) and on the clip it tracked (motion_data/g1_ue_walk_23dof.pkl).

    python oracle/ref_harness/gen_eval_golden.py      (build container only)

FK = the reference's MotionLibRobotWJX exactly as measure_traj.get_motionlib_data builds it (its OmegaConf.load of the robot yaml is replaced
by the harness's motion config: containers only); blend_motion / eval_accuracy / eval_smoothness are the reference's.  The fixture keeps the
recorded arrays the product needs (pose_aa, root_trans_offset, motion_times, terminate), the reference's FK outputs and every metric."""
import os

import numpy as np
import torch

from oracle.ref_harness import gen_golden as G

ROLLOUT = "logs/MotionTracking/phuma_student/motions/None_URCI_MujocoRobot_20260128_173245/0_pid0_frame714_20260128_173306.pkl"
CLIP = "motion_data/g1_ue_walk_23dof.pkl"


def main():
    import humanoidverse.measure_traj as MT
    from humanoidverse.utils.motion_lib.motion_lib_robot_WJX import MotionLibRobotWJX

    def tables(path):
        ml = MotionLibRobotWJX(G.motion_cfg(G.ROBOTS["g1_23dof"], path), num_envs=1, device="cpu")
        return ml.load_motions(random_sample=False)[0]

    appendix = MT.get_appendix_motion_data(ROLLOUT)
    pol = tables(ROLLOUT)
    ref_pre = tables(CLIP)
    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):            # blend_motion prints every key
        ref = MT.blend_motion(ref_pre, appendix["motion_times"])
    traj = {"pol": pol, "ref": ref, "appendix": appendix}
    out = {}
    for per_frame in (False, True):
        tag = "perframe__" if per_frame else "persec__"
        for k, v in MT.eval_accuracy(traj, per_frame).items():
            out[tag + k] = np.float64(v)
        for k, v in MT.eval_smoothness(traj, per_frame).items():
            out[tag + k] = np.float64(v)
    raw = G.safe_pkl.load(ROLLOUT)["motion0"]
    G.save("eval_metrics_student23.npz",
           rollout_file=np.array(ROLLOUT), clip_file=np.array(CLIP),
           pose_aa=raw["pose_aa"].astype(np.float32), root_trans_offset=raw["root_trans_offset"].astype(np.float32), fps=np.float64(raw["fps"]),
           motion_times=np.asarray(raw["motion_times"], np.float32), terminate=np.asarray(raw["terminate"]),
           pol__global_translation=pol["global_translation"], pol__dof_pos=pol["dof_pos"],
           refpre__global_translation=ref_pre["global_translation"], refpre__dof_pos=ref_pre["dof_pos"], refpre__fps=np.float64(ref_pre["fps"]),
           ref__global_translation=ref["global_translation"], ref__dof_pos=ref["dof_pos"],
           ref__global_rotation_extend=ref["global_rotation_extend"][:, :3], refpre__global_rotation_extend=ref_pre["global_rotation_extend"][:, :3],
           **out)


if __name__ == "__main__":
    main()
