"""LeggedRobotGeneralTracking — drop-in for the reference's KungfuBot2 (general tracking) env.

Same constructor / methods / attributes as the reference class
(reference: humanoidverse/envs/motion_tracking/general_tracking.py:29-107, on legged_robot_base.py and base_task.py), selected by
`env._target_: pbhc_amd.envs.general_tracking.LeggedRobotGeneralTracking`.  The whole `step()` is the MODE-1 instantiation of the
fused HIP kernel (true quaternion differences, anchor-relative frames, 20-step future targets, key-body rewards, the three
anchor / body-z terminations); see pbhc_amd/envs/motion_tracking.py for the shared host logic.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from . import env_config
from .motion_tracking import LeggedRobotMotionTracking

K = _lib.K


class LeggedRobotGeneralTracking(LeggedRobotMotionTracking):
    TRACKING_MODE = 1

    def __init__(self, config, device):
        super().__init__(config, device)
        L = self.layout
        self.key_body_id = L.key
        self.anchor_index = int(self._c.anchor_index)
        self.num_key_bodies = len(L.key)
        self.tar_obs_steps = torch.tensor(getattr(L, "future_steps", []), dtype=torch.long, device=self.device)     # PPO reads len(env.tar_obs_steps)
        self.curr_motion_ids = self._motion_lib.slot_clip
        self.num_motions = self._motion_lib._num_unique_motions
        self.motion_start_idx = 0

    def _load_motions_initial(self):
        self._motion_lib.load_motions(random_sample=False, max_len=self.max_len)                 # general_tracking.py:57-61: init not random sample

    def next_task(self):
        self.motion_start_idx += self.num_envs
        if self.motion_start_idx >= self.num_motions:
            self.motion_start_idx = 0
        self._motion_lib.load_motions(random_sample=False, start_idx=self.motion_start_idx, max_len=self.max_len)
        self.curr_motion_ids = self._motion_lib.slot_clip
        self.reset_all()

    def read_log(self):
        out = super().read_log()
        g = self.globals.cpu().numpy()
        L0 = K["PBHC_G_LOG"]
        extra = {
            "key_body_diff_norm": g[L0 + K["PBHC_L_KEY_BODY_DIFF_NORM"]], "local_upper_body_diff_norm": g[L0 + K["PBHC_L_LOCAL_UPPER_BODY_DIFF_NORM"]],
            "local_lower_body_diff_norm": g[L0 + K["PBHC_L_LOCAL_LOWER_BODY_DIFF_NORM"]], "local_vr_3point_diff_norm": g[L0 + K["PBHC_L_LOCAL_VR_3POINT_DIFF_NORM"]],
            "local_key_body_diff_norm": g[L0 + K["PBHC_L_LOCAL_KEY_BODY_DIFF_NORM"]], "terminate_by_ref_pos_z": g[L0 + K["PBHC_L_TERM_REF_POS_Z"]],
            "terminate_by_ref_ori": g[L0 + K["PBHC_L_TERM_REF_ORI"]], "terminate_by_body_z": g[L0 + K["PBHC_L_TERM_BODY_Z"]],
        }
        out.update(extra)
        self.log_dict.update({k: torch.tensor(float(v)) for k, v in extra.items()})
        return out
