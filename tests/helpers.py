"""Shared helpers for the parity tests (fixtures -> oracle objects)."""
import os

import numpy as np
import torch

from pbhc_amd.utils.config import load_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def skel_from_golden(robot="g1_23dof"):
    g = dict(np.load(os.path.join(GOLDEN, f"skeleton_fk_{robot}.npz")))
    return dict(parents=g["parents"], offsets=g["offsets"], local_rot_wxyz=g["local_rot_wxyz"],
                dof_axis=g["dof_axis"].astype(np.float32), num_bodies=int(g["num_bodies"]),
                body_names_ext=[str(x) for x in g["body_names"]],
                body_names=[str(x) for x in g["body_names"]][: int(g["num_bodies"])])


def load_env_golden(tag):
    return dict(np.load(os.path.join(GOLDEN, f"env_v1_{tag}.npz")))


def clip_from_env_golden(g):
    clip = dict(pose_aa=g["clip_pose_aa"], root_trans_offset=g["clip_root_trans_offset"], fps=int(g["clip_fps"]))
    if "clip_contact_mask" in g:
        clip["contact_mask"] = g["clip_contact_mask"]
    return clip


def fixture_config(name, num_envs):
    cfg = load_config(os.path.join(GOLDEN, "configs", name), {"num_envs": num_envs}, now="test")
    for k in list(cfg.obs.noise_scales.keys()):
        cfg.obs.noise_scales[k] = 0.0
    return cfg


def state_dict_from_golden(g, prefix="state0__", step=None):
    out = {}
    for k in g:
        if k.startswith(prefix):
            v = g[k]
            out[k[len(prefix):]] = v if step is None else v[step]
    return out
