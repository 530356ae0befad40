"""Reference-motion library: load-time tables + per-step phase lookup.  Test infrastructure.

Restates MotionLibBase.load_motions / get_motion_state / _calc_frame_blend / sample_time
(reference: humanoidverse/utils/motion_lib/motion_lib_base.py:123-259,261-391,486-513 and the
single-motion variant motion_lib_robot_WJX.py:146-293,394-520).  The reference runs one FK per
env slot and concatenates; slots that share a clip hold identical rows, so this restatement keeps
one table per unique clip and maps slot -> clip.
"""
import torch

from . import rotations as R
from .fk import motion_fk


class MotionLib:
    def __init__(self, skel, clips):
        """clips: list of dicts with root_trans_offset [F,3], pose_aa [F,Bx,3], fps, optional contact_mask [F,2]."""
        self.skel = skel
        self.tables = []
        starts, nframes, dts, lens = [], [], [], []
        s = 0
        for c in clips:
            fps = int(c["fps"])
            dt = 1.0 / fps
            t = motion_fk(skel, c["pose_aa"], c["root_trans_offset"], dt)
            F = t["gts_t"].shape[0]
            if "contact_mask" in c:
                t["contact"] = torch.as_tensor(c["contact_mask"]).float()
            self.tables.append(t)
            starts.append(s); nframes.append(F); dts.append(dt); lens.append(dt * (F - 1))
            s += F
        self.has_contact_mask = all("contact" in t for t in self.tables)
        keys = ["gts_t", "grs_t", "gvs_t", "gavs_t", "dof_pos", "dof_vel"] + (["contact"] if self.has_contact_mask else [])
        self.cat = {k: torch.cat([t[k] for t in self.tables], dim=0) for k in keys}
        self.length_starts = torch.tensor(starts, dtype=torch.long)
        self.num_frames = torch.tensor(nframes, dtype=torch.long)
        self.motion_dt = torch.tensor(dts, dtype=torch.float32)
        self.motion_len = torch.tensor(lens, dtype=torch.float32)

    def get_motion_length(self, ids):
        return self.motion_len[ids]

    @staticmethod
    def calc_frame_blend(time, length, num_frames, dt):
        # reference: motion_lib_base.py:503-513
        time = time.clone()
        phase = torch.clip(time / length, 0.0, 1.0)
        time[time < 0] = 0
        f0 = (phase * (num_frames - 1)).long()
        f1 = torch.min(f0 + 1, num_frames - 1)
        blend = torch.clip((time - f0 * dt) / dt, 0.0, 1.0)
        return f0, f1, blend

    def get_motion_state(self, ids, times, offset=None):
        # reference: motion_lib_base.py:123-259
        f0, f1, blend = self.calc_frame_blend(times, self.motion_len[ids], self.num_frames[ids], self.motion_dt[ids])
        f0 = f0 + self.length_starts[ids]
        f1 = f1 + self.length_starts[ids]
        b = blend.unsqueeze(-1)
        be = b.unsqueeze(-1)
        c = self.cat
        lerp2 = lambda k: (1.0 - b) * c[k][f0] + b * c[k][f1]
        lerp3 = lambda k: (1.0 - be) * c[k][f0] + be * c[k][f1]
        pos = lerp3("gts_t")
        if offset is not None:
            pos = pos + offset[..., None, :]
        rot = R.slerp(c["grs_t"][f0], c["grs_t"][f1], be)
        vel = lerp3("gvs_t")
        ang = lerp3("gavs_t")
        out = dict(
            root_pos=pos[..., 0, :].clone(), root_rot=rot[..., 0, :].clone(),
            root_vel=vel[..., 0, :].clone(), root_ang_vel=ang[..., 0, :].clone(),
            dof_pos=lerp2("dof_pos"), dof_vel=lerp2("dof_vel"),
            rg_pos_t=pos, rg_rot_t=rot, body_vel_t=vel, body_ang_vel_t=ang,
        )
        if self.has_contact_mask:
            out["contact_mask"] = lerp2("contact")
        return out
