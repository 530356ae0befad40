"""SURVEY §8 f4 (metrics half): pbhc_amd.eval against the reference's own measure_traj.py functions, run unmodified by
oracle/ref_harness/gen_eval_golden.py on a rollout the reference's deploy stack recorded (MuJoCo robot + exported student policy on
g1_ue_walk) and on the clip it tracked: tests/golden/eval_metrics_student23.npz.  CPU part: the metric arithmetic on the reference's FK
outputs; the FK itself (HIP) is covered by the GPU test below."""
import os

import numpy as np
import pytest
import torch

from pbhc_amd.eval import metrics as M
from tests.helpers import GOLDEN


def _g():
    return dict(np.load(os.path.join(GOLDEN, "eval_metrics_student23.npz")))


def _check(got, g, tag, rel=2e-5):
    keys = [k[len(tag):] for k in g if k.startswith(tag)]
    assert set(got.keys()) == set(keys), (sorted(got.keys()), sorted(keys))
    for k in keys:
        ref = float(g[tag + k])
        assert abs(float(got[k]) - ref) <= rel * abs(ref) + 1e-9, (tag + k, float(got[k]), ref)


def test_blend_and_metrics_match_the_reference_tools():
    g = _g()
    t = lambda a: torch.from_numpy(np.asarray(a))
    pre = dict(global_translation=t(g["refpre__global_translation"]), dof_pos=t(g["refpre__dof_pos"]),
               global_rotation_extend=t(g["refpre__global_rotation_extend"]), fps=float(g["refpre__fps"]))
    ref = M.blend_motion(pre, g["motion_times"])
    assert torch.allclose(ref["global_translation"], t(g["ref__global_translation"]), atol=2e-6)
    assert torch.allclose(ref["dof_pos"], t(g["ref__dof_pos"]), atol=2e-6)
    assert torch.allclose(ref["global_rotation_extend"], t(g["ref__global_rotation_extend"]), atol=2e-5)      # slerp of the rotation tables
    pol = dict(global_translation=t(g["pol__global_translation"]), dof_pos=t(g["pol__dof_pos"]))
    traj = {"pol": pol, "ref": ref, "appendix": {"fps": float(g["fps"])}}
    for per_frame, tag in ((False, "persec__"), (True, "perframe__")):
        got = dict(M.eval_accuracy(traj, per_frame))
        got.update(M.eval_smoothness(traj, per_frame))
        _check(got, g, tag)


def test_first_termination_ratio():
    # ratio_eps.py: mean index of the first termination flag, L for an episode without one
    arr = np.zeros((4, 10))
    arr[0, 3] = 1; arr[0, 7] = 1
    arr[1, 0] = 1
    arr[3, 9] = 1
    length, ratio = M.first_termination_ratio(arr)
    assert length == (3 + 0 + 10 + 9) / 4 and ratio == length / 10
    g = _g()
    length, ratio = M.first_termination_ratio(g["terminate"].reshape(1, -1))
    assert 0 <= length <= g["terminate"].size and abs(ratio - length / g["terminate"].size) < 1e-12


@pytest.mark.gpu
def test_rollout_metrics_end_to_end_on_the_gpu():
    """pose_aa / root_trans_offset of the recorded rollout and of the clip -> HIP FK -> blend -> metrics == the reference's numbers."""
    from pbhc_amd.motion_lib import load_motion_file
    from pbhc_amd.skeleton import Skeleton

    g = _g()
    sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"))
    pol = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=float(g["fps"]), motion_times=g["motion_times"], terminate=g["terminate"])
    clip = load_motion_file(os.path.join(GOLDEN, "clips", "g1_ue_walk_23dof.npz"))[0][1]
    traj = M.load_traj_data(sk, pol, clip, device="cuda:0")
    assert torch.allclose(traj["pol"]["global_translation"].cpu(), torch.from_numpy(g["pol__global_translation"]), atol=5e-6)
    assert torch.allclose(traj["pol"]["dof_pos"].cpu(), torch.from_numpy(g["pol__dof_pos"]), atol=2e-6)
    assert torch.allclose(traj["ref"]["global_translation"].cpu(), torch.from_numpy(g["ref__global_translation"]), atol=5e-6)
    for per_frame, tag in ((False, "persec__"), (True, "perframe__")):
        got = dict(M.eval_accuracy(traj, per_frame))
        got.update(M.eval_smoothness(traj, per_frame))
        _check(got, g, tag, rel=2e-3)                        # second / third finite differences x fps^2..3 amplify the FK's 1e-6
    # the batch form of sample_eps.py on a one-episode batch reproduces the per-frame numbers x 1e3
    saved = {k: np.asarray(pol[k])[None] for k in ("pose_aa", "root_trans_offset", "motion_times", "terminate")}
    saved["dof"] = g["pol__dof_pos"][None]
    res = M.eval_batch_traj(sk, saved, clip, motion_len=g["pose_aa"].shape[0], device="cuda:0")
    assert abs(res["accuracy"]["E_gmpbpe"]["mean"] - 1e3 * float(g["perframe__E_gmpbpe"])) < 2e-3 * 1e3 * float(g["perframe__E_gmpbpe"])
    assert res["accuracy"]["E_gmpbpe"]["std"] == 0.0 and len(res["_raw"]) == 1
