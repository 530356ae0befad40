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


def fixture_config(name, num_envs, overrides=None):
    cfg = load_config(os.path.join(GOLDEN, "configs", name), dict({"num_envs": num_envs}, **(overrides or {})), now="test")
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


# the narrowed layer sizes the ppo_v2 golden was generated with (oracle/ref_harness/gen_ppo_v2_golden.py)
PPO_V2_NARROW = {
    "algo.config.module_dict.actor.layer_config.hidden_dims": [64, 48, 32],
    "algo.config.module_dict.critic.layer_config.hidden_dims": [64, 48, 32],
    "algo.config.module_dict.actor.motion_encoder.hidden_dim": 12,
    "algo.config.module_dict.actor.motion_encoder.output_dim": 16,
    "algo.config.module_dict.actor.history_encoder.hidden_dim": 10,
    "algo.config.module_dict.actor.history_encoder.output_dim": 8,
    "algo.config.module_dict.actor.priv_encoder.layer_config.hidden_dims": [16],
    "algo.config.num_steps_per_env": 8,
}


# ---- HIP-side helpers (GPU tests, smoke, bench) ---------------------------------------------------
STUB = "pbhc_amd.simulator.replay_stub.ReplaySimStub"


def build_hip_env(cfgname, num_envs, device="cuda:0", noise_off=True, overrides=None, general=False):
    if general:
        from pbhc_amd.envs.general_tracking import LeggedRobotGeneralTracking as LeggedRobotMotionTracking
    else:
        from pbhc_amd.envs.motion_tracking import LeggedRobotMotionTracking

    ov = {"num_envs": num_envs, "simulator._target_": STUB}
    ov.update(overrides or {})
    cfg = load_config(os.path.join(GOLDEN, "configs", cfgname), ov, now="test")
    if noise_off:
        for k in list(cfg.obs.noise_scales.keys()):
            cfg.obs.noise_scales[k] = 0.0
    env = LeggedRobotMotionTracking(cfg.env.config, device)
    return cfg, env


def load_state_into_hip_env(env, st, g=None):
    """st: state dict with the names of oracle/ref_harness/gen_env_golden.snapshot."""
    from pbhc_amd.envs.env_config import SIGMA_KEYS
    from pbhc_amd import _lib

    K = _lib.K
    dev = env.device
    t = lambda k, dt=torch.float32: torch.as_tensor(np.asarray(st[k])).to(dev, dt)
    s = env.simulator
    s.robot_root_states.copy_(t("root_states"))
    s.dof_pos.copy_(t("dof_pos")); s.dof_vel.copy_(t("dof_vel"))
    s.contact_forces.copy_(t("contact_forces"))
    for name in ["actions", "last_actions", "actions_after_delay", "action_queue", "last_dof_pos", "last_dof_vel", "torques", "feet_air_time",
                 "contacts", "contacts_filt", "last_contacts", "last_contacts_filt", "motion_start_times", "motion_len", "end_time_ratio_buf"]:
        getattr(env, name).copy_(t(name))
    env._kp_scale.copy_(t("kp_scale")); env._kd_scale.copy_(t("kd_scale"))
    env._rfi_lim_scale.copy_(t("rfi_lim_scale")); env._rao_scale.copy_(t("rao_scale"))
    env.action_delay_idx.copy_(t("action_delay_idx", torch.long))
    env._episode_length_buf.copy_(t("episode_length_buf", torch.long))
    env.last_episode_length_buf.copy_(t("last_episode_length_buf", torch.long))
    for name, col in env.episode_sums.items():
        col.copy_(t("sum__" + name))
    for name, view in env.history.items():
        view.copy_(t("hist__" + name))
    gl = env.globals
    for i, k in enumerate(SIGMA_KEYS):
        if "sigma__" + k in st:
            gl[K["PBHC_G_SIGMA"] + i] = float(st["sigma__" + k])
            gl[K["PBHC_G_EMA"] + i] = float(st.get("ema__" + k, st["sigma__" + k]))
    gl[K["PBHC_G_PENALTY_SCALE"]] = float(st["reward_penalty_scale"])
    gl[K["PBHC_G_AVG_EP_LEN"]] = float(st["average_episode_length"])
    gl[K["PBHC_G_MOTION_FAR_THR"]] = float(st["motion_far_threshold"])
    if g is not None:
        # soft-limit curricula: the trace logs the fraction each step's reward pass used; the first one is the state the trace starts from
        # (the reference's reset_all() before the trace already moved the initial values once)
        for slot, lk in (("PBHC_G_SOFT_POS_VAL", "soft_dof_pos"), ("PBHC_G_SOFT_VEL_VAL", "soft_dof_vel"), ("PBHC_G_SOFT_TAU_VAL", "soft_torque")):
            if "step__log__" + lk + "_curriculum_value" in g:
                gl[K[slot]] = float(g["step__log__" + lk + "_curriculum_value"][0])
        if "step__log__current_noise_curriculum_value" in g:                 # likewise for the observation-noise multiplier
            gl[K["PBHC_G_NOISE_CURRICULUM"]] = float(g["step__log__current_noise_curriculum_value"][0])
        env.env_origins.copy_(torch.from_numpy(g["env_origins"]).to(dev))
        s._base_com_bias.copy_(torch.from_numpy(g["base_com_bias"]).to(dev))
        s._link_mass_scale.copy_(torch.from_numpy(g["link_mass_scale"]).to(dev))
        s.friction_coeffs.copy_(torch.from_numpy(g["friction_coeffs"]).to(dev))
        if "base_mass_scale" in g:
            s._base_mass_scale.copy_(torch.from_numpy(g["base_mass_scale"]).to(dev))


def synth_replay(ml, skel, N, T, start_times, ep_len, dt, origins, seed, feet, has_contact=True, ids=None):
    """Synthetic replay window on the CPU (oracle motion lib): state_k = ref((ep+k+1)dt+start) + noise."""
    from oracle import rotations as R

    g = torch.Generator().manual_seed(seed)
    D, B = skel["dof_axis"].shape[0], skel["num_bodies"]
    root = torch.zeros(T, N, 13); qp = torch.zeros(T, N, D); qv = torch.zeros(T, N, D); cf = torch.zeros(T, N, B, 3)
    ids = torch.zeros(N, dtype=torch.long) if ids is None else ids
    for k in range(T):
        t = (ep_len + k + 1).float() * dt + start_times
        ref = ml.get_motion_state(ids, t, offset=origins)
        root[k, :, 0:3] = ref["root_pos"] + 0.02 * torch.randn(N, 3, generator=g)
        small = R.normalize(torch.cat([0.02 * torch.randn(N, 3, generator=g), torch.ones(N, 1)], -1))
        root[k, :, 3:7] = R.normalize(R.quat_mul(small, ref["root_rot"]))
        root[k, :, 7:10] = ref["root_vel"] + 0.1 * torch.randn(N, 3, generator=g)
        root[k, :, 10:13] = ref["root_ang_vel"] + 0.1 * torch.randn(N, 3, generator=g)
        qp[k] = ref["dof_pos"] + 0.02 * torch.randn(N, D, generator=g)
        qv[k] = ref["dof_vel"] + 0.1 * torch.randn(N, D, generator=g)
        on = (ref["contact_mask"] > 0.5).float() if "contact_mask" in ref else (ref["rg_pos_t"][:, feet, 2] < 0.06).float()
        on = torch.where(torch.rand(N, 2, generator=g) < 0.1, 1 - on, on)
        cf[k, :, feet, 2] = on * (300.0 + 50.0 * torch.randn(N, 2, generator=g))
        cf[k, :, feet, 0:2] = on.unsqueeze(-1) * 20.0 * torch.randn(N, 2, 2, generator=g)
    return root, qp, qv, cf


# ---- the reference's recorded sim2sim rollout (tests/golden/deploy_student23_recording.npz) -----------------------------------
# actor_obs layout of the shipped student policy: sorted keys of obs_dict.actor_obs (deploy/urcirobot.py:349-354)
DEPLOY_LAYOUT = [("actions", 23), ("anchor_ref_rot", 6), ("base_ang_vel", 3), ("dof_pos", 23), ("dof_vel", 23), ("history_actor", 740),
                 ("next_step_ref_motion", 57), ("roll_pitch", 2)]


def deploy_layout_slices():
    out, o = {}, 0
    for k, w in DEPLOY_LAYOUT:
        out[k] = slice(o, o + w)
        o += w
    return out


def deploy_recording_states(rec, oml, default_dof_pos, obs_scales, dt):
    """Robot states the recorded observations were built from, rows 1..T-1 -> (root [T-1,13], dof_pos, dof_vel [T-1,D]).

    The state columns of the recording lag the observation by one physics sub-step (see gen_deploy_recording_fixture.py), so the state
    is recovered from the observation row itself: joint positions / velocities and the body-frame angular velocity by undoing offset and
    scale, and the base orientation from `anchor_ref_rot` = first two columns of R_robot^T R_ref (deploy/urcirobot.py:466-475; the
    robot starts on the clip's first frame, so the deploy stack's initial-yaw alignment is the identity to 1e-7) with R_ref from the
    oracle motion library at (row+1)*dt.  What this leaves to be checked, independently of the inputs: `roll_pitch` (Euler convention
    + anchor formula + library root rotation), `next_step_ref_motion` (library interpolation, FK, anchor-local key bodies, local root
    velocity), the key-major newest-first history and the whole layout / scaling."""
    from oracle import rotations as R

    sl = deploy_layout_slices()
    A = torch.from_numpy(rec["actor_obs"])
    T = A.shape[0]
    ids = torch.zeros(1, dtype=torch.long)
    roots, qs, qds = [], [], []
    for r in range(1, T):
        row = A[r]
        ref = oml.get_motion_state(ids, torch.tensor([(r + 1) * dt], dtype=torch.float32))
        Rref = R.quaternion_to_matrix_wxyz(R.xyzw_to_wxyz(ref["root_rot"]))[0]
        c = row[sl["anchor_ref_rot"]].reshape(3, 2)
        M = torch.cat([c, torch.linalg.cross(c[:, 0], c[:, 1])[:, None]], dim=1)
        q = R.wxyz_to_xyzw(R.matrix_to_quaternion_wxyz((Rref @ M.T)[None]))[0]
        q = q / q.norm()
        w_world = R.quat_rotate(q[None], (row[sl["base_ang_vel"]] / obs_scales["base_ang_vel"])[None])[0]
        roots.append(torch.cat([torch.from_numpy(rec["root_trans_offset"][r - 1]), q, torch.from_numpy(rec["root_lin_vel"][r - 1]), w_world]))
        qs.append(row[sl["dof_pos"]] / obs_scales["dof_pos"] + default_dof_pos)
        qds.append(row[sl["dof_vel"]] / obs_scales["dof_vel"])
    return torch.stack(roots), torch.stack(qs), torch.stack(qds)


def fresh_episode_state(orc):
    """Oracle-format state dict of an episode that starts at clip time 0 with zero actions and an empty history."""
    st = {k: v.clone() for k, v in orc.s.items()}
    for k in ["actions", "last_actions", "actions_after_delay", "action_queue", "last_dof_vel", "torques", "motion_start_times"]:
        st[k] = torch.zeros_like(st[k])
    for k in ["episode_length_buf", "last_episode_length_buf", "action_delay_idx"]:
        st[k] = torch.zeros_like(st[k])
    st["motion_len"] = orc.ml.motion_len[orc.motion_ids].clone()
    flat = {k: v.numpy() for k, v in st.items()}
    for k in orc.sums:
        flat["sum__" + k] = np.zeros(orc.N, np.float32)
    for k in orc.hist:
        flat["hist__" + k] = np.zeros(tuple(orc.hist[k].shape), np.float32)
    for k in orc.sigma:
        flat["sigma__" + k] = orc.sigma[k]
    flat.update(reward_penalty_scale=1.0, average_episode_length=0.0, motion_far_threshold=1.5)
    return flat


# ---- the step kernel's observation noise, restated on the host (csrc/pbhc_math.h philox4x32 + csrc/pbhc_env_step.h obs_noise_u) -------------
# The reference draws torch.rand_like per observation group and step (helpers.py:128-152): its stream cannot be reproduced, so noise-on
# parity is checked against the KERNEL's own generator: Philox4x32-7 keyed (seed, env, step counter, 16, 0) -> first word -> re-keyed by
# (row, element) -> bijective finaliser -> 24-bit uniform.
def _philox4x32_7_word0(seed, env_ids, step_ctr):
    M = np.uint64(0xFFFFFFFF)
    k0 = np.full(env_ids.shape, seed & 0xFFFFFFFF, np.uint64)
    k1 = np.full(env_ids.shape, (seed >> 32) & 0xFFFFFFFF, np.uint64)
    c0, c1 = env_ids.astype(np.uint64), np.full(env_ids.shape, step_ctr, np.uint64)
    c2, c3 = np.full(env_ids.shape, 16, np.uint64), np.zeros(env_ids.shape, np.uint64)
    for _ in range(7):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n1, n2, n3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M, p0 & M
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return c0.astype(np.uint64)


def expected_obs_noise(seed, env_ids, step_ctr, group_index, noise, scale, noise_cur=1.0):
    """[len(env_ids), dim] additive term (2u - 1) * noise[j] * noise_cur * scale[j] the step kernel puts on element j of observation row
    `group_index` (position in env.layout.group_names) at RNG step counter `step_ctr`."""
    M = np.uint64(0xFFFFFFFF)
    p0 = _philox4x32_7_word0(int(seed), np.asarray(env_ids, np.int64), int(step_ctr))[:, None]            # [n, 1]
    j = np.arange(len(noise), dtype=np.uint64)[None, :]
    x = (p0 ^ ((np.uint64(16 + group_index) * np.uint64(0x9E3779B9) + j * np.uint64(0x85EBCA6B)) & M)) & M
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    u = (x >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy((u * np.float32(2.0) - np.float32(1.0)) * (np.asarray(noise, np.float32) * np.float32(noise_cur))[None, :] * np.asarray(scale, np.float32)[None, :])


# overrides of the reference trace env_v1_walk_softlim.npz (oracle/ref_harness/gen_switch_golden.py: SOFT_LIMITS)
_LC = "rewards.reward_limit.reward_limits_curriculum."
SOFT_LIMIT_OVERRIDES = {_LC + "soft_dof_pos_curriculum": True, _LC + "soft_dof_vel_curriculum": True, _LC + "soft_torque_curriculum": True}
for _pre, _init, _lo, _hi, _deg in (("soft_dof_pos", 0.5, 0.4, 0.56, 0.05), ("soft_dof_vel", 0.3, 0.2, 0.9, 0.1), ("soft_torque", 0.1, 0.05, 0.9, 0.2)):
    SOFT_LIMIT_OVERRIDES.update({_LC + _pre + "_initial_limit": _init, _LC + _pre + "_min_limit": _lo, _LC + _pre + "_max_limit": _hi, _LC + _pre + "_curriculum_degree": _deg,
                                 _LC + _pre + "_curriculum_level_down_threshold": 40, _LC + _pre + "_curriculum_level_up_threshold": 42})

# overrides of the reference trace env_v1_walk_termnoise.npz (oracle/ref_harness/gen_switch_golden.py: TERM_NOISE)
TERM_NOISE_OVERRIDES = {"env.config.termination.terminate_by_contact": True, "env.config.termination.terminate_by_low_height": True,
                        "env.config.termination_scales.termination_min_base_height": 0.772, "obs.add_noise_currculum": True,
                        "obs.soft_dof_pos_curriculum_degree": 0.1}
