"""Generate tests/golden/*.npz by running the UNMODIFIED reference (from /root/reference) on CPU.

Build-container only:  `python oracle/ref_harness/gen_golden.py [part ...]`
The reference's sources are imported where they lie; nothing of them is copied.  Absent
non-arithmetic third-party modules are replaced by the stand-ins in oracle/ref_harness/shims;
motion .pkl files are read through the static (non-executing) reader pbhc_amd.utils.safe_pkl,
patched in for joblib.load.  The simulator is oracle.ref_harness.fake_sim.ReplayFakeSim.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "shims"), REF, os.path.join(REF, "humanoidverse", "isaac_utils"), REPO]

import numpy as np  # noqa: E402
import torch  # noqa: E402

# torch.utils.tensorboard needs the absent `tensorboard` package: give it an inert module
_tb = types.ModuleType("torch.utils.tensorboard")


class _SW:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        return lambda *a, **k: None


_tb.SummaryWriter = _SW
sys.modules["torch.utils.tensorboard"] = _tb

import joblib  # noqa: E402
from pbhc_amd.utils import safe_pkl  # noqa: E402
from pbhc_amd.utils.config import load_config  # noqa: E402

joblib.load = lambda f, *a, **k: safe_pkl.load(f)  # never unpickle reference-shipped files

from easydict import EasyDict  # noqa: E402  (shim)

GOLD = os.path.join(REPO, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
os.chdir(REF)
torch.set_num_threads(4)


def T(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def save(name, **arrs):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: T(v) for k, v in arrs.items()})
    print("wrote", path, f"{os.path.getsize(path)/1024:.1f} KiB")


# ------------------------------------------------------------------------------------------
def part_rotations():
    import isaac_utils.rotations as RR
    import humanoidverse.utils.torch_utils as TU

    g = torch.Generator().manual_seed(11)
    M = 257
    q = torch.randn(M, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    p = torch.randn(M, 4, generator=g)
    p = p / p.norm(dim=-1, keepdim=True)
    # edge cases: identity, w=-1, nearly equal pairs, antipodal pairs, gimbal (sinp = +-1)
    q[0] = torch.tensor([0, 0, 0, 1.0]); p[0] = q[0]
    q[1] = torch.tensor([0, 0, 0, -1.0])
    p[2] = q[2]
    p[3] = -q[3]
    p[4] = q[4] + 1e-4 * torch.randn(4, generator=g); p[4] /= p[4].norm()
    s = float(np.sqrt(0.5))
    q[5] = torch.tensor([0, s, 0, s]); q[6] = torch.tensor([0, -s, 0, s])
    v = torch.randn(M, 3, generator=g)
    t = torch.rand(M, 1, generator=g)
    t[7] = 0.0; t[8] = 1.0
    aa = torch.randn(M, 3, generator=g)
    aa[0] = 0.0; aa[1] = torch.tensor([1e-8, 0, 0]); aa[2] = torch.tensor([0, 0, 3.1]);
    yaw = torch.randn(M, 1, generator=g)
    mats = RR.quaternion_to_matrix(RR.xyzw_to_wxyz(q))
    ang, axis = RR.quat_to_angle_axis(q.clone())
    ang2, axis2 = RR.quat_angle_axis(q.clone(), w_last=True)
    save(
        "rotations.npz",
        q=q, p=p, v=v, t=t, aa=aa, yaw=yaw,
        my_quat_rotate=RR.my_quat_rotate(q, v),
        quat_rotate=RR.quat_rotate(q, v, w_last=True),
        quat_rotate_inverse=RR.quat_rotate_inverse(q, v, w_last=True),
        tu_quat_rotate_inverse=TU.quat_rotate_inverse(q, v),
        quat_apply=RR.quat_apply(q, v, w_last=True),
        quat_mul=RR.quat_mul(q, p, w_last=True),
        quat_conjugate=RR.quat_conjugate(q, w_last=True),
        slerp=RR.slerp(q, p, t),
        calc_heading=RR.calc_heading(q),
        calc_heading_quat=RR.calc_heading_quat(q, w_last=True),
        calc_heading_quat_inv=RR.calc_heading_quat_inv(q, w_last=True),
        calc_yaw_heading_quat_inv=RR.calc_yaw_heading_quat_inv(yaw),
        get_euler_xyz_in_tensor=RR.get_euler_xyz_in_tensor(q),
        quat_to_angle_axis_angle=ang, quat_to_angle_axis_axis=axis,
        quat_angle_axis_angle=ang2, quat_angle_axis_axis=axis2,
        quat_from_angle_axis=RR.quat_from_angle_axis(yaw[:, 0], v, w_last=True),
        axis_angle_to_quaternion=RR.axis_angle_to_quaternion(aa),
        quaternion_to_matrix=mats,
        matrix_to_quaternion=RR.matrix_to_quaternion(mats),
    )


# ------------------------------------------------------------------------------------------
EXT = [
    dict(joint_name="left_hand_link", parent_name="left_elbow_link", pos=[0.25, 0.0, 0.0], rot=[1.0, 0.0, 0.0, 0.0]),
    dict(joint_name="right_hand_link", parent_name="right_elbow_link", pos=[0.25, 0.0, 0.0], rot=[1.0, 0.0, 0.0, 0.0]),
    dict(joint_name="head_link", parent_name="torso_link", pos=[0.0, 0.0, 0.42], rot=[1.0, 0.0, 0.0, 0.0]),
]
ROBOTS = {
    "g1_23dof": "g1_23dof_lock_wrist_fitmotionONLY.xml",
    "g1_29dof": "g1_29dof_rev_1_0.xml",
}


def motion_cfg(xml, motion_file=None):
    return EasyDict(dict(asset=dict(assetRoot="description/robots/g1/", assetFileName=xml), extend_config=EXT, motion_file=motion_file, step_dt=0.02))


def part_skeleton_fk():
    from humanoidverse.utils.motion_lib.torch_humanoid_batch import Humanoid_Batch

    clips = {
        "g1_23dof": ("example/motion_data/Horse-stance_pose.pkl", None),
        "g1_29dof": ("motion_data/g1_rig_Skeleton_Sequence_converted_processed_g1_29dof_rev_1_0.pkl", 96),
    }
    for name, xml in ROBOTS.items():
        hb = Humanoid_Batch(motion_cfg(xml))
        out = dict(
            body_names=np.array(hb.body_names_augment), parents=hb._parents, offsets=hb._offsets[0],
            local_rot_wxyz=hb._local_rotation[0], dof_axis=hb.dof_axis, num_bodies=np.int64(hb.num_bodies),
        )
        f, nf = clips[name]
        clip = next(iter(safe_pkl.load(f).values()))
        pose = torch.from_numpy(clip["pose_aa"][:nf]).clone()
        trans = torch.from_numpy(clip["root_trans_offset"][:nf]).clone()
        dt = 1 / clip["fps"]
        r = hb.fk_batch(pose[None], trans[None], return_full=True, dt=dt)
        out.update(
            clip_file=np.array(f), clip_frames=np.int64(pose.shape[0]), fps=np.int64(clip["fps"]),
            pose_aa=pose, root_trans_offset=trans,
            gts_t=r.global_translation_extend[0], grs_t=r.global_rotation_extend[0],
            gvs_t=r.global_velocity_extend[0], gavs_t=r.global_angular_velocity_extend[0],
            dof_pos=r.dof_pos[0], dof_vel=r.dof_vels[0], local_rot=r.local_rotation[0],
            gts=r.global_translation[0], gvs=r.global_velocity[0], gavs=r.global_angular_velocity[0],
        )
        # sim-FK anchor: pose_aa rebuilt from (root rotvec, axis*q) as the reference does when it
        # saves rollouts (motion_tracking.py:919) must give the clip's own body poses back.
        save(f"skeleton_fk_{name}.npz", **out)


def part_motion_state():
    from humanoidverse.utils.motion_lib.motion_lib_robot_WJX import MotionLibRobotWJX
    from humanoidverse.utils.motion_lib.motion_lib_robot import MotionLibRobot

    N = 24
    for tag, cls, f in [
        ("wjx_horse", MotionLibRobotWJX, "example/motion_data/Horse-stance_pose.pkl"),
        ("origin_walk", MotionLibRobot, "motion_data/g1_walk_45cms_23dof.pkl"),
    ]:
        ml = cls(motion_cfg(ROBOTS["g1_23dof"], f), num_envs=N, device="cpu")
        torch.manual_seed(5)
        ml.load_motions(random_sample=True)
        L = float(ml._motion_lengths[0])
        dt = float(ml._motion_dt[0])
        g = torch.Generator().manual_seed(3)
        times = torch.rand(N, generator=g) * L
        times[0] = -0.3; times[1] = 0.0; times[2] = L; times[3] = L + 0.5; times[4] = 7 * dt
        times[5] = 7 * dt - 1e-6; times[6] = L - 1e-5; times[7] = 0.5 * dt
        ids = torch.arange(N)
        offset = torch.randn(N, 3, generator=g)
        res = ml.get_motion_state(ids, times, offset=offset)
        keys = ["root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos_t", "rg_rot_t", "body_vel_t", "body_ang_vel_t", "rg_pos", "rb_rot", "body_vel", "body_ang_vel"]
        if "contact_mask" in res:
            keys.append("contact_mask")
        clip = next(iter(safe_pkl.load(f).values()))
        extra = dict(pose_aa=clip["pose_aa"], root_trans_offset=clip["root_trans_offset"], fps=np.int64(clip["fps"]))
        if "contact_mask" in clip:
            extra["clip_contact_mask"] = clip["contact_mask"]
        save(f"motion_state_{tag}.npz", **extra, clip_file=np.array(f), times=times, offset=offset, motion_len=np.float32(L),
             motion_dt=np.float32(dt), num_frames=np.int64(ml._motion_num_frames[0]), **{k: res[k] for k in keys})


def part_target_heading():
    """load_motions(target_heading=...) of the reference (motion_lib_base.py:445-456): every clip re-based to face the target heading."""
    from humanoidverse.utils.motion_lib.motion_lib_robot import MotionLibRobot

    f = "motion_data/g1_walk_45cms_23dof.pkl"
    th = np.array([0.0, 0.0, np.sin(0.35), np.cos(0.35)])                    # 0.7 rad about z, xyzw
    ml = MotionLibRobot(motion_cfg(ROBOTS["g1_23dof"], f), num_envs=2, device="cpu")
    ml.load_motions(random_sample=False, target_heading=th)
    F = int(ml._motion_num_frames[0])
    clip = next(iter(safe_pkl.load(f).values()))
    save("motion_target_heading_walk.npz", clip_file=np.array(f), target_heading=th, pose_aa=clip["pose_aa"], root_trans_offset=clip["root_trans_offset"],
         fps=np.int64(clip["fps"]), gts_t=ml.gts_t[:F], grs_t=ml.grs_t[:F], gvs_t=ml.gvs_t[:F], gavs_t=ml.gavs_t[:F], dof_pos=ml.dof_pos[:F])


PARTS = dict(rotations=part_rotations, skeleton_fk=part_skeleton_fk, motion_state=part_motion_state, target_heading=part_target_heading)

if __name__ == "__main__":
    try:
        from oracle.ref_harness import gen_env_golden, gen_ppo_golden  # noqa: F401

        PARTS["env"] = gen_env_golden.main
        PARTS["ppo"] = gen_ppo_golden.main
    except ImportError as e:  # parts are added as the build progresses
        print("note:", e)
    which = sys.argv[1:] or list(PARTS)
    for w in which:
        print("==", w)
        PARTS[w]()
