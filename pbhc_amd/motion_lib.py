"""Reference-motion library on the GPU: load-time FK tables + per-step phase lookup.

Drop-in for the parts of MotionLibBase / MotionLibRobot / MotionLibRobotWJX the env uses
(reference: humanoidverse/utils/motion_lib/motion_lib_base.py:123-259,261-391,486-513;
motion_lib_robot_WJX.py:146-293).  Differences by design:
* FK + filtered velocities run once per UNIQUE clip in HIP (`pbhc_motion_build`), not once per
  env slot in Python; slots map to clips through `slot_clip`.
* all per-frame quantities live in ONE packed row per frame
  `[dof_pos D | dof_vel D | contact 2 | pos Bx*3 | rot Bx*4 | vel Bx*3 | ang Bx*3]`
  so a lookup reads two contiguous rows.
Motion files: the reference's joblib .pkl (read by the static, non-executing parser
`pbhc_amd.utils.safe_pkl`) or an .npz with the same arrays.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .utils import safe_pkl


def load_motion_file(path):
    """-> list of (name, clip dict) with root_trans_offset [F,3], pose_aa [F,Bx,3], fps, optional contact_mask [F,2]."""
    if os.path.isdir(path):
        out = []
        for f in sorted(os.listdir(path)):
            if f.endswith((".pkl", ".npz")):
                out.extend(load_motion_file(os.path.join(path, f)))
        return out
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        names = sorted({k.split("::")[0] for k in z.files})
        clips = []
        for n in names:
            c = {k.split("::")[1]: z[k] for k in z.files if k.startswith(n + "::")}
            c["fps"] = int(c["fps"])
            clips.append((n, c))
        return clips
    data = safe_pkl.load(path)
    return [(k, v) for k, v in data.items()]


def save_motion_npz(path, clips):
    arrs = {}
    for name, c in clips:
        for k in ("root_trans_offset", "pose_aa", "contact_mask"):
            if k in c:
                arrs[f"{name}::{k}"] = np.asarray(c[k])
        arrs[f"{name}::fps"] = np.int64(c["fps"])
    np.savez_compressed(path, **arrs)


def rebase_heading(clip, target_heading):
    """motion_lib_base.py:445-456 on the host, in float64 like the reference (scipy Rotation): heading_delta = target * inv(heading of the first
    root rotation); root pose_aa <- heading_delta * root rotation, root translation <- translation @ heading_delta^T."""
    from scipy.spatial.transform import Rotation as sRot

    pose = np.array(clip["pose_aa"], dtype=np.float32, copy=True)
    trans = np.asarray(clip["root_trans_offset"])
    q0 = sRot.from_rotvec(pose[0, 0].astype(np.float64)).as_quat()                     # xyzw
    # calc_heading_quat_inv (isaac_utils/rotations.py:296-306) in fp32 like the reference's torch call: rotate the x axis, atan2, -heading about z
    q = q0.astype(np.float32)
    w = q[3]
    v = np.array([1.0, 0.0, 0.0], np.float32)
    rot = v * (2.0 * w * w - 1.0) + np.cross(q[:3], v) * w * 2.0 + q[:3] * np.dot(q[:3], v) * 2.0
    heading = np.float32(np.arctan2(rot[1], rot[0]))
    hinv = np.array([0.0, 0.0, np.sin(-heading / 2.0), np.cos(-heading / 2.0)], dtype=np.float32)
    delta = sRot.from_quat(np.asarray(target_heading, dtype=np.float64)) * sRot.from_quat(hinv.astype(np.float64))
    pose[:, 0] = (delta * sRot.from_rotvec(pose[:, 0].astype(np.float64))).as_rotvec().astype(np.float32)
    out = dict(clip)
    out["pose_aa"] = pose
    out["root_trans_offset"] = (trans.astype(np.float64) @ delta.as_matrix().squeeze().T).astype(np.float32)
    return out


class MotionLib:
    def __init__(self, skeleton, clips, num_envs, device, max_len=-1):
        """clips: list of clip dicts (see load_motion_file).  max_len > 0: `load_motions` gives every env slot its own random crop of at most
        `max_len` frames of its clip (motion_lib_base.py:420-428) — the tables are then per SLOT (num_envs entries of max_len rows, allocated
        once so that the step kernel's pointers stay valid) instead of per unique clip."""
        self.skeleton = skeleton
        self.device = torch.device(device)
        self.num_envs = num_envs
        self._csk = skeleton.to_c()
        Bx, D = skeleton.num_bodies_ext, skeleton.num_dof
        self.row = 2 * D + 2 + 13 * Bx
        self.has_contact_mask = all("contact_mask" in c for c in clips)
        self._contact_size = 2
        self.max_len = int(max_len)
        self.generator = None              # torch.Generator for the sampling draws (None: the global one, as the reference)
        self._num_unique_motions = len(clips)
        # sampling hooks of the reference (setup_constants, motion_lib_base.py:109-118): per-unique-clip tensors a curriculum may write
        self._sampling_prob = torch.ones(len(clips), device=self.device) / len(clips)
        self._termination_history = torch.zeros(len(clips), device=self.device)
        self._success_rate = torch.zeros(len(clips), device=self.device)
        self._sampling_history = torch.zeros(len(clips), device=self.device)
        self.slot_clip = torch.zeros(num_envs, dtype=torch.long, device=self.device)
        self.table = _lib.PbhcMotionTable()
        if self.max_len > 0 and any(np.asarray(c["pose_aa"]).shape[0] >= self.max_len for c in clips):
            self._init_slot_tables(clips)
            return
        self.max_len = -1
        self._raw = None
        self.slot_table = self.slot_clip                 # the kernel's motion ids: unique-clip tables, slot -> clip
        self._clips = clips                              # kept for load_motions(target_heading=...) (host arrays)
        self._build_unique_tables(clips)

    def _build_unique_tables(self, clips, into=None):
        """FK + filtered velocities of EVERY unique clip in one launch set (`pbhc_motion_build_batch`): the clips' frames are concatenated on the
        host, uploaded once, and the kernels stop velocities / the Gaussian filter at clip boundaries — no per-clip launch, copy or
        synchronisation (the reference runs a Python FK per env slot).  `into`: rebuild in place (same row count)."""
        Bx = self.skeleton.num_bodies_ext
        nframes, dts = [], []
        for c in clips:
            if np.asarray(c["pose_aa"]).shape[1] < Bx:
                raise _lib.PbhcError(f"pose_aa has {np.asarray(c['pose_aa']).shape[1]} bodies, skeleton needs {Bx}")
            F = np.asarray(c["pose_aa"]).shape[0]
            if self.has_contact_mask and tuple(np.asarray(c["contact_mask"]).shape) != (F, 2):
                raise _lib.PbhcError(f"contact mask shape {tuple(np.asarray(c['contact_mask']).shape)} is not supported")
            nframes.append(F); dts.append(1.0 / int(c["fps"]))
        starts = np.concatenate([[0], np.cumsum(nframes)]).astype(np.int32)
        total = int(starts[-1])
        pose = np.concatenate([np.asarray(c["pose_aa"], dtype=np.float32)[:, :Bx] for c in clips], axis=0)
        trans = np.concatenate([np.asarray(c["root_trans_offset"], dtype=np.float32) for c in clips], axis=0)
        dev = self.device
        d_pose = torch.from_numpy(np.ascontiguousarray(pose)).to(dev)
        d_trans = torch.from_numpy(np.ascontiguousarray(trans)).to(dev)
        d_contact = None
        if self.has_contact_mask:
            d_contact = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(c["contact_mask"], dtype=np.float32) for c in clips], axis=0))).to(dev)
        d_fc = torch.from_numpy(np.repeat(np.arange(len(clips), dtype=np.int32), nframes)).to(dev)
        d_start = torch.from_numpy(starts).to(dev)
        d_dt = torch.tensor(dts, dtype=torch.float32, device=dev)
        out = torch.empty(total, self.row, device=dev) if into is None else into
        assert out.shape[0] == total
        scratch = torch.empty(total * Bx * 14, device=dev)
        _lib.check(_lib.lib().pbhc_motion_build_batch(C.byref(self._csk), _lib.ptr(d_pose), _lib.ptr(d_trans), _lib.ptr(d_contact), total, len(clips),
                                                      _lib.ptr(d_fc), _lib.ptr(d_start), _lib.ptr(d_dt), _lib.ptr(out), _lib.ptr(scratch),
                                                      _lib.current_stream()), "pbhc_motion_build_batch")
        torch.cuda.current_stream().synchronize()          # ONE synchronisation: the staging tensors above die with this scope
        if into is not None:
            return
        self.frames = out
        self.length_starts = d_start[:-1].contiguous()
        self.num_frames = torch.tensor(nframes, dtype=torch.int32, device=dev)
        self._motion_dt = d_dt
        self._motion_lengths = torch.tensor([dt * (F - 1) for dt, F in zip(dts, nframes)], dtype=torch.float32, device=dev)
        self._fill_table(len(clips), int(nframes[0]), dts[0], dts[0] * (nframes[0] - 1))

    def _fill_table(self, num_entries, f0, dt0, len0):
        self.table.frames = self.frames.data_ptr()
        self.table.row = self.row
        self.table.num_motions = num_entries
        self.table.length_starts = self.length_starts.data_ptr()
        self.table.num_frames = self.num_frames.data_ptr()
        self.table.motion_dt = self._motion_dt.data_ptr()
        self.table.motion_len = self._motion_lengths.data_ptr()
        self.table.single_num_frames = int(f0)
        self.table.single_dt = float(torch.tensor(dt0, dtype=torch.float32))
        self.table.single_len = float(torch.tensor(len0, dtype=torch.float32))

    # ---- per-slot crops (max_len) -----------------------------------------------------------
    def _init_slot_tables(self, clips):
        N, K, dev = self.num_envs, self.max_len, self.device
        Bx = self.skeleton.num_bodies_ext
        self._raw = []
        for c in clips:
            pose = torch.as_tensor(np.asarray(c["pose_aa"], dtype=np.float32)[:, :Bx]).contiguous().to(dev)
            if pose.shape[1] != Bx:
                raise _lib.PbhcError(f"pose_aa has {pose.shape[1]} bodies, skeleton needs {Bx}")
            trans = torch.as_tensor(np.asarray(c["root_trans_offset"], dtype=np.float32)).contiguous().to(dev)
            contact = torch.as_tensor(np.asarray(c["contact_mask"], dtype=np.float32)).contiguous().to(dev) if self.has_contact_mask else None
            self._raw.append((pose, trans, contact, int(c["fps"])))
        self.frames = torch.zeros(N * K, self.row, device=dev)
        self.length_starts = (torch.arange(N, dtype=torch.int32, device=dev) * K).contiguous()
        self.num_frames = torch.ones(N, dtype=torch.int32, device=dev)
        self._motion_dt = torch.ones(N, dtype=torch.float32, device=dev)
        self._motion_lengths = torch.zeros(N, dtype=torch.float32, device=dev)
        self.slot_table = torch.arange(N, dtype=torch.long, device=dev)          # slot i reads table entry i
        self._scratch = torch.empty(K * Bx * 14, device=dev)
        self.crop_starts = torch.zeros(N, dtype=torch.long)
        self._fill_table(max(N, 2), 1, 1.0, 0.0)          # >= 2: the kernel reads the per-entry arrays, never the by-value single-clip meta

    def _build_slot_crops(self, crop_starts=None):
        """FK + filtered velocities of every slot's crop, written in place (the reference crops BEFORE FK, so the velocity filter sees the
        crop's own edges, motion_lib_base.py:420-434)."""
        import random

        rnd = getattr(self, "pyrand", None) or random
        N, K = self.num_envs, self.max_len
        clip_of = self.slot_clip.tolist()
        nf, dts, lens = [], [], []
        st = _lib.current_stream()
        for i in range(N):
            pose, trans, contact, fps = self._raw[clip_of[i]]
            F = pose.shape[0]
            if F < K:
                a, b = 0, F
            else:
                a = int(crop_starts[i]) if crop_starts is not None else rnd.randint(0, F - K)
                b = a + K
            self.crop_starts[i] = a
            n = b - a
            _lib.check(_lib.lib().pbhc_motion_build(C.byref(self._csk), _lib.ptr(pose[a:b]), _lib.ptr(trans[a:b]), _lib.ptr(contact[a:b]) if contact is not None else None,
                                                    n, 1.0 / fps, _lib.ptr(self.frames[i * K:i * K + n]), _lib.ptr(self._scratch), st), "pbhc_motion_build")
            nf.append(n); dts.append(1.0 / fps); lens.append(1.0 / fps * (n - 1))
        self.num_frames.copy_(torch.tensor(nf, dtype=torch.int32))
        self._motion_dt.copy_(torch.tensor(dts, dtype=torch.float32))
        self._motion_lengths.copy_(torch.tensor(lens, dtype=torch.float32))

    @classmethod
    def from_config(cls, mcfg, skeleton, num_envs, device, max_len=-1):
        """mcfg = config.robot.motion (motion_file = .pkl / .npz / directory); max_len: see __init__."""
        path = str(mcfg.motion_file)
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(_lib.ROOT, path)
        clips = [c for _, c in load_motion_file(path)]
        if mcfg.get("motion_lib_type", "origin") == "WJX" and len(clips) != 1:
            raise _lib.PbhcError("Not Allowed to load more than one motion!")     # motion_lib_robot_WJX.py:202
        return cls(skeleton, clips, num_envs, device, max_len=max_len)

    # ---- slot -> clip assignment (load_motions, motion_lib_base.py:293-305) -----------------
    def load_motions(self, random_sample=True, start_idx=0, max_len=-1, target_heading=None, sampling_prob=None, crop_starts=None):
        """Slot -> clip assignment; the per-clip FK tables were built once at construction (the reference re-runs FK for every slot here).
        With a library built for `max_len` crops, every slot's crop is re-drawn and its table rebuilt (crop_starts: fixed starts, tests).
        `target_heading` (xyzw quaternion): every clip is yawed so that its first frame faces that heading (root rotation and translation,
        motion_lib_base.py:445-456) and the tables are rebuilt in place."""
        if target_heading is not None:
            if self.max_len > 0:
                raise NotImplementedError("load_motions(target_heading) with max_len crops")
            self._build_unique_tables([rebase_heading(c, target_heading) for c in self._clips], into=self.frames)
        if max_len != -1 and max_len != self.max_len and (self.max_len != -1 or any(int(f) >= max_len for f in self.num_frames.tolist())):
            raise _lib.PbhcError(f"load_motions(max_len={max_len}): the library was built with max_len={self.max_len} (pass robot.motion.motion_max_len at construction)")
        # in place: the step kernel holds the pointer of this tensor
        if random_sample:
            prob = self._sampling_prob if sampling_prob is None else sampling_prob
            self.slot_clip.copy_(torch.multinomial(prob, num_samples=self.num_envs, replacement=True, generator=self.generator))
        else:
            self.slot_clip.copy_(torch.remainder(torch.arange(self.num_envs, device=self.device) + start_idx, self._num_unique_motions))
        self._curr_motion_ids = self.slot_clip
        if self.max_len > 0:
            self._build_slot_crops(crop_starts)
        return self.slot_clip

    def get_motion_length(self, slot_ids=None):
        if slot_ids is None:
            return self._motion_lengths[self.slot_table]
        return self._motion_lengths[self.slot_table[slot_ids]]

    def sample_time(self, slot_ids):
        # motion_lib_base.py:486-495
        phase = torch.rand(slot_ids.shape, device=self.device, generator=self.generator)
        return phase * self.get_motion_length(slot_ids)

    def get_motion_state(self, slot_ids, motion_times, offset=None):
        """motion_lib_base.py:123-259; returns the reference's dict keys (views of one packed buffer)."""
        n = slot_ids.shape[0]
        D, Bx, B = self.skeleton.num_dof, self.skeleton.num_bodies_ext, self.skeleton.num_bodies
        ids = self.slot_table[slot_ids].contiguous()
        times = motion_times.to(torch.float32).contiguous()
        off = None if offset is None else offset.to(torch.float32).contiguous()
        out = torch.empty(n, self.row, device=self.device)
        _lib.check(_lib.lib().pbhc_motion_state(C.byref(self.table), Bx, D, _lib.ptr(ids), _lib.ptr(times), _lib.ptr(off), n,
                                                _lib.ptr(out), _lib.current_stream()), "pbhc_motion_state")
        o_pos = 2 * D + 2
        o_rot, o_vel, o_ang = o_pos + 3 * Bx, o_pos + 7 * Bx, o_pos + 10 * Bx
        pos = out[:, o_pos:o_rot].view(n, Bx, 3)
        rot = out[:, o_rot:o_vel].view(n, Bx, 4)
        vel = out[:, o_vel:o_ang].view(n, Bx, 3)
        ang = out[:, o_ang:].view(n, Bx, 3)
        res = dict(root_pos=pos[:, 0], root_rot=rot[:, 0], dof_pos=out[:, :D], root_vel=vel[:, 0], root_ang_vel=ang[:, 0],
                   dof_vel=out[:, D:2 * D], rg_pos=pos[:, :B], rb_rot=rot[:, :B], body_vel=vel[:, :B], body_ang_vel=ang[:, :B],
                   rg_pos_t=pos, rg_rot_t=rot, body_vel_t=vel, body_ang_vel_t=ang)
        if self.has_contact_mask:
            res["contact_mask"] = out[:, 2 * D:2 * D + 2]
        return res
