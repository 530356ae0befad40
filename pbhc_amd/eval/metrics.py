"""Tracking-accuracy / smoothness / survival metrics of recorded rollouts.

Same quantities, names, units and aggregation as the reference's offline tools:
  * `trajectory_tables`   = get_motionlib_data (measure_traj.py:50-68): FK of a recorded trajectory through the motion library — here the
                            HIP load-time FK (`pbhc_motion_build`), one launch set per trajectory instead of a Python loop over bodies;
  * `blend_motion`        = measure_traj.py:70-123: the reference clip re-sampled at the rollout's own time stamps (lerp, slerp for
                            rotations).  NB its clip length is num_frames / fps (not (num_frames - 1) / fps as in the motion library);
  * `eval_accuracy`       = measure_traj.py:147-222: E_gmpbpe, E_mpbpe, E_mpjpe, E_mpjve, E_mpjae, E_pbve, E_pbae, E_root_acc, E_root_vel
                            (+ E_contact_acc when both sides carry a contact mask);
  * `eval_smoothness`     = measure_traj.py:224-287: L2 norms of the finite-difference velocity / acceleration / jerk of bodies and joints;
  * `eval_batch_traj`     = sample_eps.py:21-97 / ratio_eps.py: per-episode metrics x 1e3 with per-frame differences, mean and std over episodes;
  * `first_termination_ratio` = ratio_eps.py (`calculate_average_first_one`): mean index of the first termination flag and its share of
                            the episode length.
The metric arithmetic is plain tensor code (any device); only the FK needs the GPU library (there is no CPU FK in the product).
"""
from __future__ import annotations

import numpy as np
import torch


def trajectory_tables(skeleton, motion, device="cuda:0"):
    """motion: dict with root_trans_offset [F,3], pose_aa [F,Bx,3], fps (a rollout saved by the env / deploy stack, or a reference clip).
    -> dict(global_translation [F,B,3], global_translation_extend [F,Bx,3], global_rotation_extend [F,Bx,4], dof_pos [F,D], fps)."""
    from ..motion_lib import MotionLib

    ml = MotionLib(skeleton, [dict(pose_aa=np.asarray(motion["pose_aa"], np.float32), root_trans_offset=np.asarray(motion["root_trans_offset"], np.float32),
                                   fps=int(round(float(motion["fps"]))))], 1, device)
    D, Bx, B = skeleton.num_dof, skeleton.num_bodies_ext, skeleton.num_bodies
    rows = ml.frames
    F = rows.shape[0]
    o = 2 * D + 2
    pos = rows[:, o:o + 3 * Bx].view(F, Bx, 3)
    out = dict(global_translation=pos[:, :B].contiguous(), global_translation_extend=pos.contiguous(),
               global_rotation_extend=rows[:, o + 3 * Bx:o + 7 * Bx].view(F, Bx, 4).contiguous(), dof_pos=rows[:, :D].contiguous(),
               fps=int(round(float(motion["fps"]))))
    if "contact_mask" in motion:
        out["contact_mask"] = torch.as_tensor(np.asarray(motion["contact_mask"], np.float32), device=rows.device)
    return out


def _slerp(q0, q1, t):
    # isaac_utils/rotations.py:210-232 (xyzw), as the reference's blend uses it
    cos_half = (q0 * q1).sum(-1, keepdim=True)
    neg = cos_half < 0
    q1 = torch.where(neg, -q1, q1)
    cos_half = cos_half.abs()
    half = torch.acos(cos_half)
    sin_half = torch.sqrt(1.0 - cos_half * cos_half)
    ra = torch.sin((1 - t) * half) / sin_half
    rb = torch.sin(t * half) / sin_half
    out = ra * q0 + rb * q1
    out = torch.where(sin_half.abs() < 0.001, 0.5 * q0 + 0.5 * q1, out)
    return torch.where(cos_half.abs() >= 1, q0, out)


def blend_motion(preblend, times):
    """preblend: tables of the reference clip (trajectory_tables); times [T]: the rollout's motion times.  measure_traj.py:70-123."""
    fps = preblend["fps"]
    nf = preblend["dof_pos"].shape[0]
    length = nf / fps                                        # sic: num_frames / fps
    dt = 1.0 / fps
    times = torch.as_tensor(times, dtype=torch.float32, device=preblend["dof_pos"].device).clone()
    phase = torch.clip(times / length, 0.0, 1.0)
    times[times < 0] = 0
    i0 = (phase * (nf - 1)).long()
    i1 = torch.clamp(i0 + 1, max=nf - 1)
    blend = torch.clip((times - i0 * dt) / dt, 0.0, 1.0)
    out = {}
    for k, v in preblend.items():
        if k == "fps":
            out[k] = v
            continue
        a, b = v[i0], v[i1]
        w = blend.reshape(-1, *([1] * (a.dim() - 1)))
        out[k] = _slerp(a, b, w) if "rotation" in k else w * b + (1 - w) * a
    return out


def _mean_norm(x):
    return torch.norm(x, dim=-1).mean(dim=-1).mean()


def eval_accuracy(traj, delta_per_frame=False):
    """traj: dict(pol=tables of the rollout, ref=blended reference tables, appendix=dict(fps=...)).  measure_traj.py:147-222."""
    pol, ref = traj["pol"], traj["ref"]
    delta = 1 if delta_per_frame else traj["appendix"]["fps"]
    d = lambda x: (x[1:] - x[:-1]) * delta
    pg, rg = pol["global_translation"], ref["global_translation"]
    res = {
        "E_gmpbpe": _mean_norm(pg - rg),
        "E_mpbpe": _mean_norm((pg - pg[..., 0:1, :]) - (rg - rg[..., 0:1, :])),
        "E_mpjpe": _mean_norm(pol["dof_pos"] - ref["dof_pos"]),
    }
    pdv, rdv = d(pol["dof_pos"]), d(ref["dof_pos"])
    res["E_mpjve"] = _mean_norm(pdv - rdv)
    res["E_mpjae"] = _mean_norm(d(pdv) - d(rdv))
    pv, rv = d(pg), d(rg)
    pa, ra = d(pv), d(rv)
    res["E_pbve"] = _mean_norm(pv - rv)
    res["E_pbae"] = _mean_norm(pa - ra)
    res["E_root_acc"] = _mean_norm(pa[..., 0:1, :] - ra[..., 0:1, :])
    res["E_root_vel"] = _mean_norm(pv[..., 0:1, :] - rv[..., 0:1, :])
    if "contact_mask" in pol and "contact_mask" in ref:
        res["E_contact_acc"] = torch.mean((pol["contact_mask"] - ref["contact_mask"]).abs(), dim=-1).mean()
    return res


def eval_smoothness(traj, delta_per_frame=False):
    """measure_traj.py:224-287"""
    pol, ref = traj["pol"], traj["ref"]
    delta = 1 if delta_per_frame else traj["appendix"]["fps"]
    d = lambda x: (x[1:] - x[:-1]) * delta
    out = {}
    for tag, t in (("", pol), ("ref_", ref)):
        v = d(t["global_translation"]); a = d(v); j = d(a)
        dv = d(t["dof_pos"]); da = d(dv); dj = d(da)
        out.update({f"L2_{tag}vel": _mean_norm(v), f"L2_{tag}acc": _mean_norm(a), f"L2_{tag}jerk": _mean_norm(j),
                    f"L2_{tag}dof_vel": _mean_norm(dv), f"L2_{tag}dof_acc": _mean_norm(da), f"L2_{tag}dof_jerk": _mean_norm(dj)})
    return out


def load_traj_data(skeleton, pol_motion, ref_motion, device="cuda:0"):
    """measure_traj.py:125-145: rollout + reference clip -> the `traj` dict of the metric functions."""
    appendix = {"motion_times": torch.as_tensor(np.asarray(pol_motion["motion_times"], np.float32)).reshape(-1), "fps": pol_motion["fps"]}
    for k in ("action", "actor_obs", "terminate"):
        if k in pol_motion:
            appendix[k] = torch.as_tensor(np.asarray(pol_motion[k]))
    pol = trajectory_tables(skeleton, pol_motion, device)
    ref = blend_motion(trajectory_tables(skeleton, ref_motion, device), appendix["motion_times"])
    return {"pol": pol, "ref": ref, "appendix": appendix}


def eval_batch_traj(skeleton, saved_motion_dict, ref_motion, motion_len=None, device="cuda:0"):
    """sample_eps.py:21-97: `saved_motion_dict` holds [N, L, ...] arrays of N recorded episodes (dof, pose_aa, root_trans_offset, motion_times,
    terminate, ...).  Per episode: accuracy + smoothness with per-frame differences, x 1e3; then mean / std over the episodes."""
    N, L = saved_motion_dict["dof"].shape[0], saved_motion_dict["dof"].shape[1]
    if motion_len is not None:
        assert L == motion_len, f"Motion length {L} does not match the expected length {motion_len}"
    ref_pre = trajectory_tables(skeleton, ref_motion, device)
    total = {"_raw": []}
    ref = None
    for i in range(N):
        ep = {k: np.asarray(v)[i] for k, v in saved_motion_dict.items()}
        ep["fps"] = 50                                      # sample_eps.py:40
        times = torch.as_tensor(np.asarray(ep["motion_times"], np.float32)).reshape(-1)
        if i == 0:                                          # the reference blends once, at episode 0's time stamps (sample_eps.py:47-48)
            ref = blend_motion(ref_pre, times)
        traj = {"pol": trajectory_tables(skeleton, ep, device), "ref": ref, "appendix": {"fps": 50, "motion_times": times}}
        total["_raw"].append({"accuracy": {k: float(v) * 1e3 for k, v in eval_accuracy(traj, True).items()},
                              "smoothness": {k: float(v) * 1e3 for k, v in eval_smoothness(traj, True).items()}})
    for part in ("accuracy", "smoothness"):
        agg = {}
        for key in total["_raw"][0][part]:
            arr = np.array([total["_raw"][i][part][key] for i in range(N)])
            agg[key] = {"mean": float(np.mean(arr)), "std": float(np.std(arr))}
        total[part] = agg
    return total


def first_termination_ratio(terminate):
    """ratio_eps.py `calculate_average_first_one`: terminate [N, L] (0/1) -> (mean index of the first 1, L where an episode has none;
    that mean / L)."""
    arr = np.asarray(terminate)
    arr = arr.reshape(arr.shape[0], -1)
    first = np.argmax(arr, axis=1)
    first[np.max(arr, axis=1) == 0] = arr.shape[1]
    length = float(np.mean(first))
    return length, length / arr.shape[1]
