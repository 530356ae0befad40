"""GPU parity tests: the HIP path (through the C ABI of libpbhc_hip.so) against the oracle and the
reference-generated golden traces.  Tolerances are fp32: rtol = atol = 2e-5 on observations /
states, 1e-4 on exp()-amplified reward terms and velocity tables (different libm, tree reductions)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.helpers import (GOLDEN, SOFT_LIMIT_OVERRIDES, TERM_NOISE_OVERRIDES, build_hip_env, clip_from_env_golden, fixture_config, load_env_golden, load_state_into_hip_env,
                           skel_from_golden, state_dict_from_golden, synth_replay)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# Two quantities of the reference are discontinuous / ill-conditioned in fp32.  The HIP load-time FK already follows the reference's op
# order (3x3 matrix chain, plain left-to-right products exactly like torch's CPU matmul, matrix_to_quaternion candidates): table
# ROTATIONS agree to 2.4e-7 = 2 ulp (tools/tolerance_probe.py, profiles/round2_tolerance_probe.txt).  What is left comes from the
# 1-ulp differences between glibc's and the GPU's sinf / cosf / acosf, amplified by the reference's own formulas:
#  * table angular velocities are axis * acos(2w^2-1) / dt of a near-identity quaternion (torch_humanoid_batch.py:282-290):
#    d(omega) = 4 dw / (omega dt^2), so k ulps of w move a slow body's omega by k * 4 * 6e-8 / (omega dt^2).  Bound PER ELEMENT by that
#    conditioning with k = 4 ulps (`angvel_tol`; measured: 0 of 17010 table elements above it, the worst 5.9e-3 at |omega| < 0.05 rad/s,
#    1.8e-4 above 1 rad/s) — not a blanket tolerance;
#  * slerp (rotations.py:210-232) switches to 0.5*(q0+q1) when sin(half angle) < 1e-3, and it forms that sine as sqrt(1 - cos^2) with
#    cos within 5e-7 of 1: one ulp of the dot product moves sin^2 by 12 %, so pairs with sin(half angle) in [0.7e-3, 1.4e-3] can take
#    either branch (a jump of up to |t-0.5|*|q1-q0| ~ 1e-3).  Where the pair is known (`slerp_flip_zone`) the loose bound applies ONLY
#    inside that zone and the plain one everywhere else; where only the outputs are at hand (env traces) SLERP keeps the two-tier form
#    (`hard` = worst case, `frac` = share of elements allowed above `tol`; measured share 0.35 % on the near-static Horse-stance clip).
SLERP = dict(hard=1.2e-3, frac=0.02)
ANGVEL = dict(hard=1e-2, frac=0.02, tol_override=2e-3)     # env traces only (root angular velocity written by a reset: no frame pair at hand)


def slerp_scale_bound(q0, q1, k=4.0):
    """[..., 1] relative bound on the SCALE of a slerp output of the pair (q0, q1) — what a vector rotated by it inherits twice over.
    The reference's slerp (rotations.py:210-232) divides sin(t a) by s = sqrt(1 - c^2) with c the fp32 dot product: k ulps of c move
    1 - c^2 by 2 k 6e-8, i.e. s by k 6e-8 / s^2 relative, while a = acos(c) moves with c itself — so for slowly turning frame pairs the
    result is q (1 +- k 6e-8 / s^2), not unit, and nothing downstream renormalises it (s = 2.5e-2, a body turning at 1.5 rad/s sampled at
    30 fps: 4e-4).  Below s = 1.4e-3 the fall-back jump of `slerp_jump_bound` takes over."""
    q0, q1 = torch.as_tensor(q0).double(), torch.as_tensor(q1).double()
    q1 = torch.where(((q0 * q1).sum(-1, keepdim=True) < 0), -q1, q1)
    c = (q0 * q1).sum(-1, keepdim=True).clamp(max=1.0)
    sh = torch.sqrt(1.0 - c * c)
    jump = 0.5 * (q1 - q0).norm(dim=-1, keepdim=True) + 2e-5
    return torch.where(sh < 1.4e-3, jump, (k * 6e-8 / (sh * sh).clamp(min=1e-12)).clamp(max=2e-2)).float()


def angvel_tol(ref, dt, k=16.0, base=5e-5, norm=None):
    """elementwise bound for table / looked-up angular velocities: base + k * eps / (dt^2 |omega|), |omega| floored at 1e-3 rad per frame.
    `norm` [..., 1]: the speed that conditions the element when it is not the element's own (see table_speed_floor)."""
    ref = torch.as_tensor(ref).float()
    wn = (ref.norm(dim=-1, keepdim=True) if norm is None else torch.as_tensor(norm).float()).clamp(min=1e-3 / dt)
    return (base + k * 6e-8 / (dt * dt * wn)).expand_as(ref)


def table_speed_floor(gavs_t, radius=8):
    """[F, Bx, 1]: the slowest |omega| within the reference's Gaussian filter window (sigma 2, truncate 4 -> 8 frames either side): a table
    entry is a weighted mean of the RAW per-frame values around it, and the slow ones among them carry the error"""
    n = torch.as_tensor(gavs_t).float().norm(dim=-1)                       # [F, Bx]
    F = n.shape[0]
    idx = (torch.arange(F)[:, None] + torch.arange(-radius, radius + 1)[None, :]).clamp(0, F - 1)       # [F, 2r+1]
    return n[idx].min(dim=1).values.unsqueeze(-1)


def slerp_jump_bound(q0, q1, base=5e-5):
    """[..., 1] bound on a slerp output of the pair (q0, q1): the reference's slerp (rotations.py:210-232) has two fall-backs, 0.5*(q0+q1)
    when sin(half angle) < 1e-3 and q0 when cos >= 1, and it forms that sine as sqrt(1 - cos^2) with cos within 5e-7 of 1 — an ulp of the
    dot product moves sin^2 by 12 % there — so a pair with sin(half angle) below 1.4e-3 may take another branch than the reference did: a
    jump of at most 0.5 * |q1 - q0| (between the interpolated and the averaged / first quaternion).  Above it: the plain bound."""
    q0, q1 = torch.as_tensor(q0).double(), torch.as_tensor(q1).double()
    q1 = torch.where(((q0 * q1).sum(-1, keepdim=True) < 0), -q1, q1)
    c = (q0 * q1).sum(-1, keepdim=True).clamp(max=1.0)
    sh = torch.sqrt(1.0 - c * c)
    jump = 0.5 * (q1 - q0).norm(dim=-1, keepdim=True) + 2e-5
    return torch.where(sh < 1.4e-3, base + jump, torch.full_like(jump, base)).float()


def trace_slerp_bounds(oml, ep_before, start_before, ep_after, start_after, reset, dt, base):
    """Per-element bounds for the two slerp-made quantities of a REFERENCE TRACE step (round 3 gave them a two-tier blanket: up to 2 % of the
    elements 40x off), from the frame pair each env's lookup blended — recoverable from the trace's own state: the step looks the reference
    up at (episode_length + 2) dt + start with the values BEFORE the step (motion_tracking.py:554,588), a reset at (0 + 1) dt + its new start
    (motion_tracking.py:477-507).  -> (bodies [N, Bx, 1] for ref_body_rot_extend, root [N, 1] for the root rotation a reset wrote): `base`
    outside the flip zone / scale conditioning of THAT pair (slerp_scale_bound), the pair's own bound inside."""
    c = oml.cat["grs_t"]
    z = torch.zeros_like(ep_before)
    t1 = (ep_before + 2).float() * dt + start_before
    f0, f1, _ = oml.calc_frame_blend(t1, oml.motion_len[z], oml.num_frames[z], oml.motion_dt[z])
    bodies = base + slerp_scale_bound(c[f0], c[f1])                             # [N, Bx, 1]
    t2 = (ep_after * 0 + 1).float() * dt + start_after
    g0, g1, _ = oml.calc_frame_blend(t2, oml.motion_len[z], oml.num_frames[z], oml.motion_dt[z])
    root = torch.where(reset.bool()[:, None], base + slerp_scale_bound(c[g0][:, 0], c[g1][:, 0]), torch.full((len(z), 1), base))
    return bodies, root


def _probe(line):
    """PBHC_TOLERANCE_PROBE=<file>: append one line per bounded quantity (profiles/round4_tolerance_probe.txt is a copy of one run)"""
    path = os.environ.get("PBHC_TOLERANCE_PROBE")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def close(a, b, tol, what, rtol=None, hard=None, frac=0.0, tol_override=None):
    tol = tol if tol_override is None else tol_override
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    if torch.is_tensor(tol):                                  # elementwise absolute bound (conditioned tolerances)
        tol = tol.detach().float().cpu()
        assert tol.shape == b.shape, (what, tol.shape, b.shape)
        rtol = 0.0 if rtol is None else rtol
    lim = tol + (tol if rtol is None else rtol) * b.abs()
    bad = err > lim
    if hard is not None:
        assert float(err.max()) <= hard, (what, "hard bound", float(err.max()))
        assert float(bad.float().mean()) <= frac, (what, float(err.max()), float(bad.float().mean()))
        return
    if torch.is_tensor(tol) and os.environ.get("PBHC_TOLERANCE_PROBE"):
        _probe(f"{what}: worst residual / bound {float((err / lim.clamp(min=1e-12)).max()):.3f}, max residual {float(err.max()):.2e}, elements above 3e-5 "
               f"{int((err > 3e-5 + 3e-5 * b.abs()).sum())} of {err.numel()}, loosest bound {float(tol.max()):.2e}")
    assert not bool(bad.any()), (what, float(err.max()), int(bad.sum()), bad.nonzero()[:5].tolist())


def _hip_motion_lib(clip, N=8):
    from pbhc_amd.motion_lib import MotionLib
    from pbhc_amd.skeleton import Skeleton

    sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"))
    return sk, MotionLib(sk, [clip], N, DEV)


def test_skeleton_json_matches_reference_tables():
    from pbhc_amd.skeleton import Skeleton

    sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"))
    g = skel_from_golden()
    assert sk.body_names_ext == g["body_names_ext"]
    assert np.array_equal(sk.parents, g["parents"]) and np.allclose(sk.offsets, g["offsets"]) and np.allclose(sk.local_rot_wxyz, g["local_rot_wxyz"])
    assert np.allclose(sk.dof_axis, g["dof_axis"])


def test_motion_build_matches_reference_fk():
    """pbhc_motion_build vs the reference's Humanoid_Batch.fk_batch outputs (golden)."""
    g = dict(np.load(os.path.join(GOLDEN, "skeleton_fk_g1_23dof.npz")))
    clip = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=int(g["fps"]))
    sk, ml = _hip_motion_lib(clip)
    D, Bx = sk.num_dof, sk.num_bodies_ext
    rows = ml.frames.cpu()
    F = rows.shape[0]
    o = 2 * D + 2
    close(rows[:, :D], g["dof_pos"], 2e-6, "dof_pos")
    close(rows[:, D:2 * D], g["dof_vel"], 2e-5, "dof_vel")
    close(rows[:, o:o + 3 * Bx].view(F, Bx, 3), g["gts_t"], 3e-6, "gts_t")
    rot = rows[:, o + 3 * Bx:o + 7 * Bx].view(F, Bx, 4)
    close(rot, g["grs_t"], 3e-6, "grs_t")
    close(rows[:, o + 7 * Bx:o + 10 * Bx].view(F, Bx, 3), g["gvs_t"], 5e-5, "gvs_t")
    # angular velocity = axis * acos(2w^2-1) / dt of a near-identity quaternion (torch_humanoid_batch.py:282-290): bounded element by
    # element by its conditioning (see angvel_tol above), plus the table-wide relative Frobenius error
    gav = rows[:, o + 10 * Bx:].view(F, Bx, 3)
    ref = torch.from_numpy(g["gavs_t"])
    close(gav, ref, angvel_tol(ref, 1.0 / int(g["fps"]), norm=table_speed_floor(ref)), "gavs_t")
    assert float((gav - ref).norm() / ref.norm()) < 1e-3


@pytest.mark.parametrize("tag", ["wjx_horse", "origin_walk"])
def test_motion_state_matches_reference(tag):
    g = dict(np.load(os.path.join(GOLDEN, f"motion_state_{tag}.npz")))
    clip = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=int(g["fps"]))
    if "clip_contact_mask" in g:
        clip["contact_mask"] = g["clip_contact_mask"]
    N = g["times"].shape[0]
    sk, ml = _hip_motion_lib(clip, N)
    res = ml.get_motion_state(torch.arange(N, device=DEV), torch.from_numpy(g["times"]).to(DEV), torch.from_numpy(g["offset"]).to(DEV))
    # the frame pair of every lookup (oracle tables = the reference's, tests/test_oracle_motion.py): where can the reference's slerp flip?
    from oracle.motion_lib import MotionLib as OML

    oml = OML(skel_from_golden(), [clip])
    times = torch.from_numpy(g["times"])
    ids = torch.zeros(N, dtype=torch.long)
    f0, f1, _ = oml.calc_frame_blend(times, oml.motion_len[ids], oml.num_frames[ids], oml.motion_dt[ids])
    rot_tol = slerp_jump_bound(oml.cat["grs_t"][f0], oml.cat["grs_t"][f1])          # [N, Bx, 1]
    speed = table_speed_floor(oml.cat["gavs_t"])                                    # [F, Bx, 1]
    speed = torch.minimum(speed[f0], speed[f1])                                     # [N, Bx, 1]
    dt_clip = float(oml.motion_dt[0])
    for k in ["root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos_t", "rg_rot_t", "body_vel_t", "body_ang_vel_t",
              "rg_pos", "rb_rot", "body_vel", "body_ang_vel"] + (["contact_mask"] if "contact_mask" in g else []):
        ref = torch.from_numpy(g[k])
        if "ang_vel" in k:
            # lerp of two table rows: the slower of the two frames' filter windows conditions the pair
            sp = speed[:, 0] if ref.dim() == 2 else speed[:, :ref.shape[1]]
            close(res[k], ref, angvel_tol(ref, dt_clip, norm=sp), f"{tag}:{k}")
        elif "rot" in k:
            rt = rot_tol[:, 0] if ref.dim() == 2 else rot_tol[:, :ref.shape[1]]
            close(res[k], ref, rt.expand_as(ref), f"{tag}:{k}", rtol=5e-5)
        else:
            close(res[k], ref, 5e-5, f"{tag}:{k}")


def test_library_batch_build_directory_mode_and_target_heading(tmp_path):
    """Motion-library ingestion: a DIRECTORY of clip files (motion_lib_base.py:63-107) built by ONE batched launch set — every
    clip's rows equal its own single-clip build bit for bit (velocities and the Gaussian filter stop at clip boundaries) — and
    load_motions(target_heading=...) (motion_lib_base.py:445-456) against the tables the reference built with that heading."""
    import bench
    from pbhc_amd import _lib
    from pbhc_amd.motion_lib import MotionLib, save_motion_npz
    from pbhc_amd.skeleton import Skeleton

    sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"))
    g = dict(np.load(os.path.join(GOLDEN, "motion_target_heading_walk.npz")))
    base = dict(pose_aa=g["pose_aa"], root_trans_offset=g["root_trans_offset"], fps=int(g["fps"]))
    clips = bench.synth_library(base, 5, seed=1)
    for i, c in enumerate(clips):                                       # one file per clip, names out of order: the directory is read sorted
        save_motion_npz(str(tmp_path / f"clip_{(7 * i) % 5}.npz"), [(f"m{i}", c)])
    mcfg = type("Cfg", (dict,), {"__getattr__": dict.get})(motion_file=str(tmp_path), motion_lib_type="origin")
    ml = MotionLib.from_config(mcfg, sk, 16, DEV)
    order = sorted(range(5), key=lambda i: f"clip_{(7 * i) % 5}.npz")
    assert ml._num_unique_motions == 5 and ml.num_frames.tolist() == [clips[i]["pose_aa"].shape[0] for i in order]
    for k, i in enumerate(order):
        single = MotionLib(sk, [clips[i]], 1, DEV)
        a = int(ml.length_starts[k])
        assert torch.equal(ml.frames[a:a + single.frames.shape[0]], single.frames), f"clip {i}: batched rows differ from its own build"
    # target heading: the library of ONE clip the reference fixture was made from
    one = MotionLib(sk, [base], 4, DEV)
    ptr = one.frames.data_ptr()
    one.load_motions(random_sample=False, target_heading=g["target_heading"])
    assert one.frames.data_ptr() == ptr and one.table.frames == ptr       # rebuilt in place: the step kernel's pointers stay valid
    rows = one.frames.cpu()
    D, Bx = sk.num_dof, sk.num_bodies_ext
    F = rows.shape[0]
    o = 2 * D + 2
    close(rows[:, o:o + 3 * Bx].view(F, Bx, 3), g["gts_t"], 5e-6, "target heading gts_t")
    close(rows[:, o + 3 * Bx:o + 7 * Bx].view(F, Bx, 4).abs(), np.abs(g["grs_t"]), 5e-6, "target heading grs_t")
    close(rows[:, :D], g["dof_pos"], 2e-6, "target heading dof_pos")
    close(rows[:, o + 7 * Bx:o + 10 * Bx].view(F, Bx, 3), g["gvs_t"], 1e-4, "target heading gvs_t")


def test_sim_fk_matches_oracle():
    from oracle.fk import sim_fk
    from pbhc_amd import _lib
    from pbhc_amd.skeleton import Skeleton

    sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"))
    osk = skel_from_golden()
    N, D, B = 1000, sk.num_dof, sk.num_bodies
    gen = torch.Generator().manual_seed(0)
    root = torch.randn(N, 13, generator=gen)
    root[:, 3:7] /= root[:, 3:7].norm(dim=-1, keepdim=True)
    q, qd = torch.randn(N, D, generator=gen), torch.randn(N, D, generator=gen)
    out = torch.zeros(N, B, 13, device=DEV)
    csk = sk.to_c()
    rg, qg, qdg = root.to(DEV), q.to(DEV).contiguous(), qd.to(DEV).contiguous()
    _lib.check(_lib.lib().pbhc_sim_fk(C.byref(csk), rg.data_ptr(), qg.data_ptr(), qdg.data_ptr(), 1, N, out.data_ptr(), _lib.current_stream()))
    p, r, v, w = sim_fk(osk, root, q, qd)
    close(out[..., 0:3], p, 1e-5, "pos"); close(out[..., 3:7], r, 1e-5, "rot"); close(out[..., 7:10], v, 2e-5, "vel"); close(out[..., 10:13], w, 2e-5, "ang")


CASES = [("horse", "v1_g1_23dof_horse_stance.yaml"), ("walk", "v1_g1_23dof_walk.yaml")]


@pytest.mark.parametrize("tag,cfgname,overrides", [("walk_ctrlV", "v1_g1_23dof_walk.yaml", {"robot.control.control_type": "V"}),
                                                   ("walk_ctrlT", "v1_g1_23dof_walk.yaml", {"robot.control.control_type": "T"}),
                                                   ("walk_feetori", "v1_g1_23dof_walk.yaml", {"rewards.reward_scales.feet_heading_alignment": -0.5, "rewards.reward_scales.feet_heading_alignment_contact": -0.3, "rewards.reward_scales.penalty_feet_ori": -0.2, "rewards.reward_scales.penalty_feet_ori_contact": -0.4}),
                                                   ("walk_softlim", "v1_g1_23dof_walk.yaml", SOFT_LIMIT_OVERRIDES),
                                                   ("walk_termnoise", "v1_g1_23dof_walk.yaml", TERM_NOISE_OVERRIDES)])
def test_env_step_matches_reference_trace_of_a_switch(tag, cfgname, overrides):
    """control types "V" and "T" (legged_robot_base.py:809-817), the four foot-orientation reward terms (:1030-1079), the soft-limit
    curricula with min != max (:902-939), terminate_by_contact / terminate_by_low_height (:434-447) and the observation-noise curriculum
    (:591-592,637-646) — no shipped yaml uses them: the reference's own traces with the switch on (oracle/ref_harness/gen_switch_golden.py)"""
    test_env_step_matches_reference_trace(tag, cfgname, overrides)
    if tag == "walk_softlim":                            # the fractions themselves, step by step (logged by the reference's reward pass)
        g = load_env_golden(tag)
        assert float(g["step__log__soft_dof_pos_curriculum_value"][-1]) > float(g["step__log__soft_dof_pos_curriculum_value"][0])


@pytest.mark.parametrize("tag,cfgname", CASES)
def test_env_step_matches_reference_trace(tag, cfgname, overrides=None):
    """The fused HIP step replays the reference's own trace (same inputs, injected draws)."""
    g = load_env_golden(tag)
    T, N, D = g["actions_in"].shape
    cfg, env = build_hip_env(cfgname, N, overrides=overrides)
    assert env.reward_names == list(g["reward_names"])
    load_state_into_hip_env(env, state_dict_from_golden(g), g)
    dev = env.device
    tg = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    env.simulator.set_replay(tg(g["replay_root"]), tg(g["replay_dof_pos"]), tg(g["replay_dof_vel"]), tg(g["replay_contact"]))
    K = __import__("pbhc_amd._lib", fromlist=["K"]).K
    from oracle.motion_lib import MotionLib as OML                 # (the frame pairs the trace's lookups blended: for the per-element slerp bounds)

    oml = OML(skel_from_golden(), [clip_from_env_golden(g)])
    for k in range(T):
        st = lambda name, dt=torch.float32: tg(g["step__state__" + name][k]).to(dt)
        env.set_injected_draws(u_rfi=tg(g["step__u_rfi"][k]), start_time=st("motion_start_times"), kp=st("kp_scale"), kd=st("kd_scale"),
                               rfi_lim=st("rfi_lim_scale"), rao=st("rao_scale"), delay=st("action_delay_idx", torch.long))
        obs, rew, reset, extras = env.step({"actions": tg(g["actions_in"][k])})
        torch.cuda.synchronize()
        w = f"{tag} step {k}: "
        assert torch.equal(reset.cpu(), torch.from_numpy(g["step__reset_buf_out"][k])), w + "reset_buf"
        assert torch.equal(extras["time_outs"].cpu(), torch.from_numpy(g["step__time_outs"][k])), w + "time_outs"
        # the two slerp-made quantities: bounded per element by the frame pair THIS env blended (trace_slerp_bounds), 2e-5 / 3e-5 elsewhere
        gs = lambda name, j: torch.from_numpy(g["state0__" + name] if j < 0 else g["step__state__" + name][j])
        rot_tol, root_tol = trace_slerp_bounds(oml, gs("episode_length_buf", k - 1), gs("motion_start_times", k - 1), gs("episode_length_buf", k),
                                               gs("motion_start_times", k), torch.from_numpy(g["step__reset_buf_out"][k]), float(env.dt), 2e-5)
        close(extras["ref_body_pos_extend"], g["step__ref_body_pos_extend"][k], 2e-5, w + "ref_body_pos_extend")
        close(extras["ref_body_rot_extend"], g["step__ref_body_rot_extend"][k], rot_tol.expand(-1, -1, 4), w + "ref_body_rot_extend", rtol=2e-5)
        close(env.simulator._rigid_body_pos, g["step__x___rigid_body_pos_extend"][k][:, :env.num_bodies], 2e-5, w + "body pos")
        close(rew, g["step__rew_buf"][k], 3e-5, w + "rew_buf", rtol=1e-4)
        for ok in ["actor_obs", "critic_obs"]:
            close(obs[ok], g["step__obs__" + ok][k], 3e-5, w + ok)
        for name in ["torques", "feet_air_time", "last_contacts", "actions", "last_actions", "action_queue", "motion_len", "end_time_ratio_buf",
                     "contacts_filt", "last_dof_vel", "motion_start_times"]:
            close(getattr(env, name), g["step__state__" + name][k], 3e-5, w + "state " + name)
        close(env.simulator.dof_pos, g["step__state__dof_pos"][k], 3e-5, w + "dof_pos")
        rs_tol = torch.full((N, 10), 3e-5)
        rs_tol[:, 3:7] = (root_tol + 1e-5).expand(-1, 4)                          # (a surviving env's root state is the replay frame itself)
        close(env.simulator.robot_root_states[:, :10], g["step__state__root_states"][k][:, :10], rs_tol, w + "root_states", rtol=3e-5)
        # a reset writes the looked-up root angular velocity of the reference motion: bounded by ITS conditioning (angvel_tol), everything else 3e-5
        ref_w = torch.from_numpy(g["step__state__root_states"][k][:, 10:])
        close(env.simulator.robot_root_states[:, 10:], ref_w, angvel_tol(ref_w, float(env._motion_lib._motion_dt[0]), k=32.0, base=3e-5), w + "root ang vel", rtol=3e-5)
        assert torch.equal(env.episode_length_buf.cpu(), torch.from_numpy(g["step__state__episode_length_buf"][k]))
        for name, col in env.episode_sums.items():
            close(col, g["step__state__sum__" + name][k], 3e-5, w + "sum " + name, rtol=1e-4)
        for name, view in env.history.items():
            close(view, g["step__state__hist__" + name][k], 3e-5, w + "hist " + name)
        gl = env.globals.cpu().numpy()
        from pbhc_amd.envs.env_config import SIGMA_KEYS
        for i, name in enumerate(SIGMA_KEYS):
            if "step__state__sigma__" + name not in g:
                continue
            ref = float(g["step__state__sigma__" + name][k])
            assert abs(gl[K["PBHC_G_SIGMA"] + i] - ref) <= 2e-6 * abs(ref), w + "sigma " + name
        assert abs(gl[K["PBHC_G_PENALTY_SCALE"]] - float(g["step__state__reward_penalty_scale"][k])) < 1e-9
        assert abs(gl[K["PBHC_G_AVG_EP_LEN"]] - float(g["step__state__average_episode_length"][k])) < 1e-5
        assert abs(gl[K["PBHC_G_MOTION_FAR_THR"]] - float(g["step__state__motion_far_threshold"][k])) < 1e-9
        log = env.read_log()
        for lk in ["terminate_by_gravity", "terminate_by_motion_far", "terminate_by_time_out", "upper_body_diff_norm", "joint_pos_diff_norm", "action_clip_frac"]:
            close(torch.tensor(log[lk]), g["step__log__" + lk][k], 1e-4, w + "log " + lk)
        for lk in ("terminate_by_contact", "terminate_by_low_height"):
            if "step__log__" + lk in g:
                close(torch.tensor(log[lk]), g["step__log__" + lk][k], 1e-4, w + "log " + lk)
        for lk in ("soft_dof_pos_curriculum_value", "soft_dof_vel_curriculum_value", "soft_torque_curriculum_value", "current_noise_curriculum_value"):
            # the reference logs the fraction its reward pass USED in a step; after our step k the global holds the one step k + 1 will use
            if "step__log__" + lk in g and k + 1 < g["step__log__" + lk].shape[0]:
                ref = float(g["step__log__" + lk][k + 1])
                assert abs(float(log[lk]) - ref) <= 2e-7 * abs(ref), w + lk


def test_env_step_matches_oracle_4096():
    """Full-size (4096 envs) HIP step vs the oracle on identical synthetic replay tensors."""
    _env_step_vs_oracle(4096, 4)


def test_env_step_matches_oracle_4096_walk_config():
    """BASELINE configs[1] at its full size: 4096 envs, G1 23-DoF, g1_walk_45cms (no contact mask -> teleop_contact_mask 0, 20 reward
    columns) — the bench's own workload — HIP step vs the oracle on identical synthetic replay tensors, resets and time-outs included."""
    _env_step_vs_oracle(4096, 4, tag="walk", cfgname="v1_g1_23dof_walk.yaml")


def test_env_step_matches_oracle_ragged_tail():
    """13 envs: the last workgroup is only partly filled (loads of its missing envs are clamped onto env N-1, nothing is stored for them),
    and the env-count-dependent reductions (log means, curricula) divide by 13."""
    _env_step_vs_oracle(13, 4)


def test_env_step_contact_and_low_height_terminations():
    """termination.terminate_by_contact / terminate_by_low_height (legged_robot_base.py:434-444; off in the shipped yamls): contact forces on
    `robot.terminate_after_contacts_on` bodies and a base-height floor above some envs' root height — reset flags, rewards, observations
    and the per-cause log means against the oracle."""
    ov = {"env.config.termination.terminate_by_contact": True, "env.config.termination.terminate_by_low_height": True,
          "env.config.termination_scales.termination_min_base_height": 0.74}
    _env_step_vs_oracle(512, 3, overrides=ov, contact_hits=True)


def test_env_step_noise_curriculum_follows_the_oracle():
    """obs.add_noise_currculum (legged_robot_base.py:119-124,591-592,1117-1126): the global noise multiplier starts at obs.noise_initial_value and
    moves by (1 -/+ degree) at every step that resets an env, keyed on the average episode length (device rule in k_env_finalize)."""
    ov = {"obs.add_noise_currculum": True, "obs.soft_dof_pos_curriculum_degree": 0.1}
    _env_step_vs_oracle(512, 4, overrides=ov, noise_curriculum=True)


def test_env_step_soft_limit_curricula_follow_the_oracle():
    """rewards.reward_limit.reward_limits_curriculum with min != max (legged_robot_base.py:902-939; the shipped yamls pin the value with
    min == max): the three soft-limit fractions move by (1 +/- degree) at every step that resets an env — episodes are short here, so the limits
    widen step by step until the clip — and the limit penalties are computed against the moving value (device rule in k_env_finalize)."""
    lc = "rewards.reward_limit.reward_limits_curriculum."
    ov = {lc + "soft_dof_pos_curriculum": True, lc + "soft_dof_vel_curriculum": True, lc + "soft_torque_curriculum": True}
    for pre, init, lo, hi, deg in (("soft_dof_pos", 0.5, 0.4, 0.56, 0.05), ("soft_dof_vel", 0.3, 0.2, 0.9, 0.1), ("soft_torque", 0.1, 0.05, 0.9, 0.2)):
        ov.update({lc + pre + "_initial_limit": init, lc + pre + "_min_limit": lo, lc + pre + "_max_limit": hi, lc + pre + "_curriculum_degree": deg,
                   lc + pre + "_curriculum_level_down_threshold": 40, lc + pre + "_curriculum_level_up_threshold": 42})
    _env_step_vs_oracle(512, 4, overrides=ov, soft_limits=True)


def test_env_step_close_to_limit_terminations():
    """termination.terminate_when_close_to_{dof_pos,dof_vel,torque}_limit (legged_robot_base.py:449-479; off in the shipped yamls): with a
    per-step probability an env terminates when a joint is beyond its termination limit — the three gates' uniforms are injected (open / closed
    per step and gate), the limits are tightened so that a part of the envs is beyond them; reset flags, rewards, observations and the per-cause
    log means against the oracle."""
    ov = {"env.config.termination.terminate_when_close_to_dof_pos_limit": True, "env.config.termination.terminate_when_close_to_dof_vel_limit": True,
          "env.config.termination.terminate_when_close_to_torque_limit": True,
          "env.config.termination_scales.termination_close_to_dof_pos_limit": 0.55, "env.config.termination_scales.termination_close_to_dof_vel_limit": 0.02,
          "env.config.termination_scales.termination_close_to_torque_limit": 0.35}
    _env_step_vs_oracle(512, 4, overrides=ov, gates=[[0.1, 0.9, 0.9], [0.9, 0.1, 0.9], [0.9, 0.9, 0.1], [0.2, 0.2, 0.2]])


def test_env_step_randomized_default_dof_pos_matches_oracle():
    """domain_rand.randomize_default_dof_pos (legged_robot_base.py:632-635; off in the shipped yamls): per-env default joint angles redrawn at
    every reset; the torques of the following steps and the dof_pos observation use them."""
    ov = {"domain_rand.randomize_default_dof_pos": True, "domain_rand.dof_pos_range": [-0.05, 0.05]}
    _env_step_vs_oracle(512, 4, overrides=ov, default_bias=True)


def test_observation_noise_scales_with_the_noise_curriculum():
    """With the curriculum on, every noisy element's amplitude is noise * scale * current_noise_curriculum_value (0.05 initially)."""
    N = 2048
    outs = []
    for noise_off in (True, False):
        torch.manual_seed(11)
        cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", N, noise_off=noise_off, overrides={"obs.add_noise_currculum": True})
        torch.manual_seed(12)
        env.reset_all()
        obs, _, _, _ = env.step({"actions": torch.zeros(N, env.num_dof, device=DEV)})
        torch.cuda.synchronize()
        outs.append(({k: v.clone() for k, v in obs.items()}, cfg, env.layout, float(env.globals[_lib_K()["PBHC_G_NOISE_CURRICULUM"]])))
    (clean, _, _, _), (noisy, cfg, L, cur) = outs
    ob = cfg.obs
    assert abs(cur - float(ob.noise_initial_value)) < 0.2 * float(ob.noise_initial_value) and cur < 0.5      # (one curriculum move at most)
    pos, seen = 0, 0
    for key in sorted(ob.obs_dict["actor_obs"]):
        d = L.obs_dims[key] if key in L.obs_dims else sum(L.obs_dims[k] * n for k, n in ob.obs_auxiliary[key].items())
        amp = float(ob.noise_scales[key]) * float(ob.obs_scales[key]) if key not in ob.obs_auxiliary else 0.0
        if amp > 0.0:
            diff = (noisy["actor_obs"][:, pos:pos + d] - clean["actor_obs"][:, pos:pos + d]).cpu()
            init = float(ob.noise_initial_value)
            assert 0.5 * amp * init < float(diff.abs().max()) <= amp * init * 1.000011 + 1e-7, (key, float(diff.abs().max()), amp * init)
            seen += 1
        pos += d
    assert seen >= 3


def _lib_K():
    from pbhc_amd import _lib
    return _lib.K


def _env_step_vs_oracle(N, T, tag="horse", cfgname="v1_g1_23dof_horse_stance.yaml", overrides=None, contact_hits=False, noise_curriculum=False, default_bias=False, gates=None,
                        redraw_steps=(), soft_limits=False):
    from oracle.env_v1 import MotionTrackingOracle
    from oracle.fk import sim_fk
    from oracle.motion_lib import MotionLib as OML

    cfg, env = build_hip_env(cfgname, N, overrides=overrides)
    g = load_env_golden(tag)
    skel = skel_from_golden()
    oml = OML(skel, [clip_from_env_golden(g)])
    ocfg = fixture_config(cfgname, N, overrides)
    dr = dict(base_com_bias=env.simulator._base_com_bias.cpu(), link_mass_scale=env.simulator._link_mass_scale.cpu(), friction_coeffs=env.simulator.friction_coeffs.cpu())
    orc = MotionTrackingOracle(ocfg, skel, oml, N, dr)
    orc.env_origins = env.env_origins.cpu()
    orc.ref_init_yaw = env.ref_init_yaw
    gen = torch.Generator().manual_seed(5)
    start = torch.rand(N, generator=gen) * float(oml.motion_len[0])
    ep = torch.randint(0, 50, (N,), generator=gen)
    start[:max(N // 64, 2)] = float(oml.motion_len[0]) - 0.03          # some motion-end time-outs
    st = {k: v.clone() for k, v in orc.s.items()}
    st["motion_start_times"] = start; st["episode_length_buf"] = ep; st["last_episode_length_buf"] = ep.clone()
    st["motion_len"] = torch.full((N,), float(oml.motion_len[0]))
    st["kp_scale"] = 0.9 + 0.2 * torch.rand(N, 23, generator=gen); st["kd_scale"] = 0.9 + 0.2 * torch.rand(N, 23, generator=gen)
    st["rfi_lim_scale"] = 0.5 + torch.rand(N, 23, generator=gen); st["rao_scale"] = 0.1 * (torch.rand(N, 23, generator=gen) - 0.5)
    st["action_delay_idx"] = torch.randint(0, 3, (N,), generator=gen)
    root, qp, qv, cf = synth_replay(oml, skel, N, T + 1, start, ep, orc.dt, orc.env_origins, 6, orc.feet)
    if contact_hits:                                     # every 9th env: 5 N on one of the terminating bodies, in a different step each
        assert len(orc.termination_contact) >= 2
        for k in range(1, T + 1):
            ids = torch.arange(k, N, 9 * T)
            cf[k, ids, orc.termination_contact[k % len(orc.termination_contact)], 1] = 5.0
        root[1:, ::7, 2] -= 0.08                         # every 7th env sinks below the base-height floor
    st["root_states"], st["dof_pos"], st["dof_vel"], st["contact_forces"] = root[0], qp[0], qv[0], cf[0]
    flat = {k: v.numpy() for k, v in st.items()}
    for k in orc.sums:
        flat["sum__" + k] = np.zeros(N, np.float32)
    for k in orc.hist:
        flat["hist__" + k] = (0.1 * torch.randn(orc.hist[k].shape, generator=gen)).numpy()
    for k in orc.sigma:
        flat["sigma__" + k] = orc.sigma[k]
    flat.update(reward_penalty_scale=0.1, average_episode_length=0.0, motion_far_threshold=1.5)
    orc.load_state(flat)
    load_state_into_hip_env(env, flat)
    tg = lambda a: a.contiguous().to(DEV)
    env.simulator.set_replay(tg(root[1:]), tg(qp[1:]), tg(qv[1:]), tg(cf[1:]))
    kept = []
    for k in range(T):
        act = 0.5 * torch.randn(N, 23, generator=gen)
        u = torch.rand(N, 23, generator=gen)
        samp = dict(motion_start_times=torch.rand(N, generator=gen) * float(oml.motion_len[0]), kp_scale=0.9 + 0.2 * torch.rand(N, 23, generator=gen),
                    kd_scale=0.9 + 0.2 * torch.rand(N, 23, generator=gen), rfi_lim_scale=0.5 + torch.rand(N, 23, generator=gen),
                    rao_scale=0.1 * (torch.rand(N, 23, generator=gen) - 0.5), action_delay_idx=torch.randint(0, 3, (N,), generator=gen))
        if default_bias:
            samp["dof_pos_bias"] = 0.1 * (torch.rand(N, 23, generator=gen) - 0.5)
        frame = dict(root=root[k + 1], dof_pos=qp[k + 1], dof_vel=qv[k + 1], contact=cf[k + 1])
        body = sim_fk(skel, frame["root"], frame["dof_pos"], frame["dof_vel"])
        redraw = k in redraw_steps
        o_obs, o_rew, o_reset, o_ex = orc.step(act, frame, body, u_rfi=u, reset_samples=samp, gate_u=gates[k] if gates else None,
                                               redraw_samples=samp if redraw else None)
        if redraw:                                       # reinit_epis_rand fires in THIS step (the schedule itself: tests/test_reinit_schedule.py)
            env._reinit.mean, env._reinit.counter = 1e12, float(env.common_step_counter + 1)
        env.set_injected_draws(u_rfi=tg(u), start_time=tg(samp["motion_start_times"]), kp=tg(samp["kp_scale"]), kd=tg(samp["kd_scale"]),
                               rfi_lim=tg(samp["rfi_lim_scale"]), rao=tg(samp["rao_scale"]), delay=tg(samp["action_delay_idx"]),
                               dof_pos_bias=tg(samp["dof_pos_bias"]) if default_bias else None,
                               gate_u=torch.tensor(gates[k], dtype=torch.float32, device=DEV) if gates else None)
        obs, rew, reset, extras = env.step({"actions": tg(act)})
        torch.cuda.synchronize()
        w = f"step {k}: "
        assert torch.equal(reset.cpu(), o_reset), w + f"reset mismatch {int((reset.cpu() != o_reset).sum())}"
        assert int(o_reset.sum()) > 0 or k > 0
        close(rew, o_rew, 3e-5, w + "rew", rtol=2e-4)
        for ok in o_obs:      # env origins reach 320 m: fp32 spacing there is 3e-5, and the position differences cancel at that magnitude
            close(obs[ok], o_obs[ok], 1e-4, w + ok)
        close(env.torques, orc.s["torques"], 3e-5, w + "torques", rtol=1e-5)
        for name, view in env.history.items():
            close(view, orc.hist[name], 3e-5, w + "hist " + name)
        if redraw_steps:                                 # the episodic DR state of EVERY env after a re-draw (and of the reset ones otherwise)
            for a_, b_ in ((env._kp_scale, "kp_scale"), (env._kd_scale, "kd_scale"), (env._rfi_lim_scale, "rfi_lim_scale"), (env._rao_scale, "rao_scale")):
                close(a_, orc.s[b_], 1e-7, w + b_)
            assert torch.equal(env.action_delay_idx.cpu(), orc.s["action_delay_idx"]), w + "action_delay_idx"
            close(env.action_queue, orc.s["action_queue"], 1e-7, w + "action_queue")
            if redraw:
                assert float((orc.s["kp_scale"] - samp["kp_scale"]).abs().max()) == 0.0 and float(orc.s["action_queue"].abs().max()) == 0.0
        if gates:
            log = env.read_log()
            cause = ("dof_pos_limit", "dof_vel_limit", "torque_limit")[k] if k < 3 else None
            for c_ in ("dof_pos_limit", "dof_vel_limit", "torque_limit"):
                close(torch.tensor(log["terminate_by_" + c_]), orc.log["terminate_by_" + c_], 1e-5, w + "log terminate_by_" + c_)
                if cause is not None:                                    # only the open gate may fire, and it does for a part of the envs
                    assert (float(orc.log["terminate_by_" + c_]) > 0) == (c_ == cause), (k, c_, float(orc.log["terminate_by_" + c_]))
        if default_bias and "default_dof_pos" in orc.s:
            close(env.default_dof_pos, orc.s["default_dof_pos"], 1e-7, w + "default_dof_pos")
            assert float((orc.s["default_dof_pos"] - orc.default_dof_pos).abs().max()) > 0.01
        if soft_limits:
            log = env.read_log()
            for name, val in (("soft_dof_pos", orc.soft_pos_val), ("soft_dof_vel", orc.soft_vel_val), ("soft_torque", orc.soft_tau_val)):
                close(torch.tensor(float(log[name + "_curriculum_value"])), torch.tensor(float(val)), 1e-12, w + name + " curriculum value", rtol=1e-12)
            if k == T - 1:                                # moved from the initial values (pos: into its clip), and the penalties saw it
                assert orc.soft_pos_val == 0.56 and 0.3 < orc.soft_vel_val < 0.9 and 0.1 < orc.soft_tau_val < 0.9
                for term in ("limits_dof_pos", "limits_dof_vel", "limits_torque"):
                    assert term in env.episode_sums, term
        if noise_curriculum:
            assert orc.noise_curriculum and 0.0 < orc.noise_cur < 0.05
            close(torch.tensor(float(env.read_log()["current_noise_curriculum_value"])), torch.tensor(orc.noise_cur), 1e-9, w + "noise curriculum value")
        if contact_hits:
            log = env.read_log()
            for cause in ("contact", "low_height", "gravity"):
                close(torch.tensor(log["terminate_by_" + cause]), orc.log["terminate_by_" + cause], 1e-5, w + "log terminate_by_" + cause)
            assert float(orc.log["terminate_by_contact"]) > 0 and float(orc.log["terminate_by_low_height"]) > 0
        # extras["episode"] (legged_robot_base.py:510-515): per reset env, gathered lazily on the device.  Step k's mapping is read one step
        # LATE (after step k+1 ran): env.step() materialises a mapping that somebody kept before it overwrites the buffers.
        if int(o_reset.sum()) > 0:
            ids = o_reset.nonzero().flatten()
            want = dict(orc.episode_extras)
            want = {n: v.clone() for n, v in want.items()}
            want["end_epis_length"] = orc.s["last_episode_length_buf"][ids].clone()
            kept.append((k, extras["episode"], want, ids))
    assert kept, "no step with a reset"
    for k, ep, want, ids in kept:
        assert set(ep.keys()) == {"rew_" + n for n in env.episode_sums} | {"end_epis_length"}
        assert ep["end_epis_length"].shape[0] == len(ids)
        for name, v in want.items():
            close(ep[name].float(), v.float(), 3e-5, f"step {k}: extras episode {name}", rtol=2e-4)


def test_reinit_epis_rand_redraws_every_env_inside_the_step():
    """domain_rand.reinit_epis_rand (legged_robot_base.py:390-395 -> :599-635): in the step where the schedule fires, the gain / torque-noise
    scales, the control delay and the action queue of EVERY env are re-drawn after that step's torques and before its observations (the
    privileged dr_kp / dr_kd / dr_ctrl_delay observations of the same step show the new values; the next step's torques use them)."""
    _env_step_vs_oracle(64, 4, redraw_steps=(1, 2))


def test_lazy_simulator_surface_and_reference_bodies_equal_the_stored_ones():
    """The fused step stores neither the simulator surface's rigid-body state / contact forces nor extras['ref_body_*_extend'] unless asked
    (PbhcStepIO: NULL); read afterwards they are re-derived from the replay frame / the step's reference time.  Both forms of the same
    step: the kernel's own stores (set_eager_outputs) against the lazy tensors — bit for bit where both sides run the same un-fused
    arithmetic (reference lerp / slerp, contact copy); the rigid-body chain to 2e-6 for poses and 1e-5 for twists: the stand-alone FK
    kernel WALKS every body's chain (renormalising the rotation at every joint), the specialised step kernel composes the same chain by
    pointer jumping (csrc/pbhc_env_step.h: fk_jump_wave — another association order, one renormalisation at the end), and a joint
    turning at 10 rad/s carries those few ulps into its angular velocity."""
    cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", 96)
    obs = env.reset_all()
    import bench

    env.simulator.set_replay(*bench.make_replay_on_device(env, 8, 3))
    env.set_eager_outputs(True)
    gen = torch.Generator(device=DEV).manual_seed(3)
    for k in range(5):
        if k == 2:
            env.episode_length_buf[::5] = 10 ** 6          # resets: the surface keeps the PRE-reset frame, the reference bodies the pre-reset time
        obs, rew, reset, extras = env.step({"actions": 0.3 * torch.randn(96, env.num_dof, device=DEV, generator=gen)})
        torch.cuda.synchronize()
        e = env._eager
        close(env.simulator._rigid_body_state[..., :7], e["rigid_body_state"][..., :7], 2e-6, f"step {k} lazy rigid-body pose", rtol=2e-6)
        close(env.simulator._rigid_body_state[..., 7:], e["rigid_body_state"][..., 7:], 1e-5, f"step {k} lazy rigid-body twist", rtol=1e-5)
        assert torch.equal(env.simulator.contact_forces, e["contact_forces"]), k
        assert torch.equal(extras["ref_body_pos_extend"], e["ref_body_pos_extend"]), k
        assert torch.equal(extras["ref_body_rot_extend"], e["ref_body_rot_extend"]), k
        close(env.simulator._rigid_body_pos, e["rigid_body_state"][..., 0:3], 2e-6, f"step {k} lazy rigid-body pos", rtol=2e-6)
        if k == 2:
            assert int(reset.sum()) >= 19


def test_gae_matches_reference_storage():
    from pbhc_amd import _lib

    g = {k: v for k, v in np.load(os.path.join(GOLDEN, "ppo_v1.npz")).items()}
    T, N, R = g["st__rewards"].shape
    tg = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rewards, values, dones = tg(g["st__rewards"]), tg(g["st__values"]), tg(g["st__dones"][..., 0])
    # last_values of the golden = critic(last_obs) with the initial weights: recompute on CPU via the oracle nets
    from oracle import ppo
    cp = {k[len("critic__"):]: torch.from_numpy(v) for k, v in g.items() if k.startswith("critic__")}
    last_values = ppo.mlp_forward(cp, "critic_module", torch.from_numpy(g["last_critic_obs"]))
    returns = torch.zeros(T, N, R, device=DEV); adv = torch.zeros(T, N, device=DEV)
    stats = torch.zeros(2 * ((T * N + 255) // 256) + 4, dtype=torch.float64, device=DEV)
    _lib.check(_lib.lib().pbhc_gae(rewards.data_ptr(), values.data_ptr(), dones.data_ptr(), tg(last_values.numpy()).data_ptr(), T, N, R, 0.99, 0.95,
                                   returns.data_ptr(), adv.data_ptr(), stats.data_ptr(), _lib.current_stream()))
    torch.cuda.synchronize()
    close(returns, g["st__returns"], 2e-5, "returns")
    close(adv.unsqueeze(-1), g["st__advantages"], 5e-5, "advantages")


def test_mhppo_update_matches_reference():
    """One _training_step (5 epochs x 4 minibatches) of pbhc_amd MHPPO on the reference's rollout buffer, initial weights and
    permutation reproduces the reference's updated weights, losses and learning rate (tests/golden/ppo_v1.npz)."""
    from pbhc_amd.agents.mh_ppo import MHPPO

    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v1.npz")).items()}
    N = g["st__actions"].shape[1]
    cfg, env = build_hip_env("v1_g1_23dof_horse_stance.yaml", N)
    hd = [int(x) for x in g["hidden_dims"]]
    cfg.algo.config.module_dict.actor.layer_config.hidden_dims = hd
    cfg.algo.config.module_dict.critic.layer_config.hidden_dims = hd
    algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
    algo.setup()
    algo.actor.load_state_dict({k[len("actor__"):]: v for k, v in g.items() if k.startswith("actor__")}, strict=True)
    algo.critic.load_state_dict({k[len("critic__"):]: v for k, v in g.items() if k.startswith("critic__")}, strict=True)
    for k in algo.storage.stored_keys:
        getattr(algo.storage, k).copy_(g["st__" + k].to(DEV))
    algo._train_mode()
    loss = algo._training_step(indices=g["perm"].to(DEV))
    torch.cuda.synchronize()
    for k in ["Value", "Surrogate", "Entropy"]:
        assert abs(float(loss[k]) - float(g["loss__" + k])) < 2e-4 * max(1.0, abs(float(g["loss__" + k]))), k
    assert abs(float(algo._lr_a) - float(g["lr_actor"])) < 1e-9 and abs(float(algo._lr_c) - float(g["lr_critic"])) < 1e-9
    # Adam moves a weight by ~lr per step whatever the gradient's size, so an element whose gradient sign is decided by fp32
    # rounding can drift by a few lr (lr: 1e-3 -> 1e-5 over the 20 steps): elementwise bound 2e-4, whole-tensor bound 1e-4 relative.
    for name, sd in (("actor", algo.actor.state_dict()), ("critic", algo.critic.state_dict())):
        for k, v in sd.items():
            ref = g[f"{name}1__" + k]
            close(v, ref, 2e-4, f"{name} {k}", rtol=2e-4)
            assert float((v.cpu() - ref).norm() / ref.norm().clamp(min=1e-6)) < 1e-4, (name, k)


def test_checkpoint_roundtrip_uses_reference_keys(tmp_path):
    from pbhc_amd.agents.mh_ppo import MHPPO

    cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", 16)
    algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
    algo.setup()
    p = str(tmp_path / "model_0.pt")
    algo.save(p, infos={"x": 1})
    d = torch.load(p, map_location="cpu", weights_only=False)
    assert set(d.keys()) == {"actor_model_state_dict", "critic_model_state_dict", "actor_optimizer_state_dict", "critic_optimizer_state_dict", "iter", "infos"}
    assert list(d["actor_model_state_dict"].keys()) == ["std"] + [f"actor_module.module.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    assert list(d["critic_model_state_dict"].keys()) == [f"critic_module.module.{i}.{w}" for i in (0, 2, 4, 6) for w in ("weight", "bias")]
    assert d["critic_model_state_dict"]["critic_module.module.6.weight"].shape[0] == env.num_rew_fn
    assert isinstance(d["actor_optimizer_state_dict"]["param_groups"][0]["lr"], float)
    assert algo.load(p) == {"x": 1}
    # what the reference's exporters do with `inference_model` (utils/inference_helpers.py:13-52): deepcopy the actor to the CPU, call
    # act_inference on an example observation (torch.onnx.export itself needs the `onnx` package, absent here) — and TorchScript tracing
    import copy

    actor_cpu = copy.deepcopy(algo.inference_model["actor"]).to("cpu")
    ex = algo.get_example_obs()["actor_obs"][:4].cpu()
    with torch.no_grad():
        ref_out = algo.inference_model["actor"].act_inference(ex.to(DEV)).cpu()
        assert torch.allclose(actor_cpu.act_inference(ex), ref_out, atol=1e-5)
        traced = torch.jit.trace(actor_cpu.actor_module, ex)
        assert torch.allclose(traced(ex), ref_out, atol=1e-5)
    # ... and the exporter itself: the file evaluates (numpy ONNX reader) to the GPU actor's action on the example observation
    from pbhc_amd.utils import inference_helpers as ih

    file = ih.export_policy_as_onnx(algo.inference_model, str(tmp_path), "model_0.onnx", algo.get_example_obs())
    assert ih.check_onnx(file, algo.inference_model, algo.get_example_obs(), atol=1e-5) <= 1e-5


def _rollouts_with_split(split, agent="v1", batched=False, fused_sample=True, rollout_graph=False, rollouts=3, train_between=False, new_replay_at=None):
    """three rollouts (eager, graph capture, graph replay) from the same seeds; returns the last rollout's buffer + env state.
    train_between: the timed loop of bench.py / learn() — `_training_step()` after every rollout (a fixed permutation per iteration), so the
    weights the captured policy launch reads change under the graph.  new_replay_at: `simulator.set_replay()` with another window before
    rollout number `new_replay_at` (a captured graph holds the old window's addresses)."""
    import os

    os.environ["PBHC_ROLLOUT_GRAPH"] = "1" if rollout_graph else "0"
    os.environ["PBHC_ROLLOUT_SPLIT"] = "1" if split else "0"
    os.environ["PBHC_CRITIC_BATCHED"] = "1" if batched else "0"
    os.environ["PBHC_FUSED_SAMPLE"] = "1" if fused_sample else "0"
    try:
        torch.manual_seed(11)
        np.random.seed(11)
        if agent == "v1":
            from pbhc_amd.agents.mh_ppo import MHPPO as Algo

            cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", 512, noise_off=False)
        else:
            from tests.test_gpu_parity_v2 import _v2_algo

            cfg, env, algo = _v2_algo(256, noise_off=False)
        if agent == "v1":
            algo = Algo(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
            algo.setup()
        algo._train_mode()
        obs = env.reset_all()
        if rollout_graph or rollouts > 3:                      # a replay window (the graph's steps read the device-side frame cursor)
            import bench

            env.simulator.set_replay(*bench.make_replay_on_device(env, rollouts * algo.num_steps_per_env + 2, seed=5))
        used = []
        gperm = torch.Generator().manual_seed(23)
        pflat_start = algo._pflat.clone() if train_between else None
        for r in range(rollouts):
            if new_replay_at is not None and r == new_replay_at:
                import bench

                env.simulator.set_replay(*bench.make_replay_on_device(env, (rollouts - r) * algo.num_steps_per_env + 3, seed=9))
            algo.storage.clear()
            obs = algo._rollout_step(obs)
            used.append(bool(getattr(algo, "_rollout_used_graph", False)))
            if train_between:
                n = algo.storage.num_envs * algo.storage.num_transitions_per_env
                algo._training_step(indices=torch.randperm(n, generator=gperm).to(DEV))
        torch.cuda.synchronize()
        st = algo.storage
        out = {k: getattr(st, k).clone() for k in st.stored_keys}
        out["_used_graph_each"] = torch.tensor(used)
        if train_between:
            for k in ("_pflat", "_mflat", "_vflat", "_lr", "_adam_step", "_gflat"):
                v = getattr(algo, k)
                if isinstance(v, (list, tuple)):                   # (ppo_mimic: one moment buffer per optimiser)
                    for i, vi in enumerate(v):
                        out[f"{k}{i}"] = vi.clone()
                else:
                    out[k] = v.clone()
            out["cur_reward_sum"], out["cur_episode_length"] = algo.cur_reward_sum.clone(), algo.cur_episode_length.clone()
            out["_pflat_start"] = pflat_start
        out["_used_graph"] = torch.tensor([bool(getattr(algo, "_rollout_used_graph", False))])
        out["common_step_counter"] = torch.tensor([env.common_step_counter])
        out["frame_cursor"] = env.simulator.frame_cursor.clone().cpu()
        out["globals"] = env.globals.clone()
        out["episode_sums"] = env._episode_sums.clone()
        out["ep_stats"] = algo._ep_stats.clone()
        if batched:
            out["_time_outs_seen"] = algo._time_outs.any().reshape(1)
        return out
    finally:
        os.environ.pop("PBHC_ROLLOUT_GRAPH", None)
        os.environ.pop("PBHC_ROLLOUT_SPLIT", None)
        os.environ.pop("PBHC_CRITIC_BATCHED", None)
        os.environ.pop("PBHC_FUSED_SAMPLE", None)


@pytest.mark.parametrize("agent", ["v1", "v2"])
def test_rollout_branch_stream_equals_one_stream(agent):
    """The rollout keeps the env step's reduction and the bootstrap kernel (and the v1 per-step critic) on a branch stream next to the
    step -> policy -> sampling chain; that is a schedule, not arithmetic: buffers, globals (sigma EMA, curricula, step counter) and episode
    statistics are bit-identical to the one-stream order (mh_ppo.py:270-342, ppo_mimic.py:371-438)."""
    a = _rollouts_with_split(True, agent)
    b = _rollouts_with_split(False, agent)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("agent", ["v1", "v2"])
def test_rollout_as_one_graph_equals_the_eager_loop(agent):
    """The agents' default rollout replays ONE hipGraph for the control steps of a rollout (MHPPO: policy stack + sampling, fused env step,
    its reduction, the done / episode-statistics kernel; ppo_mimic.PPO: encoders + packed actor / critic stacks, sampling, env step, reduction,
    bootstrap kernel; frame index from the device-side cursor): a schedule, not arithmetic — after five rollouts (eager, capture + replay,
    three replays) every buffer, the env's globals / episode sums, the episode statistics, the step counter and the replay cursor are
    bit-identical to the step-by-step loop (mh_ppo.py:270-342, ppo_mimic.py:369-440)."""
    a = _rollouts_with_split(True, agent, batched=True, fused_sample=True, rollout_graph=True, rollouts=5)
    b = _rollouts_with_split(True, agent, batched=True, fused_sample=True, rollout_graph=False, rollouts=5)
    a.pop("_time_outs_seen"); b.pop("_time_outs_seen")
    assert bool(a.pop("_used_graph")) and not bool(b.pop("_used_graph"))
    assert a.pop("_used_graph_each").tolist() == [False, True, True, True, True] and not b.pop("_used_graph_each").any()
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("agent", ["v1", "v2"])
def test_graph_rollout_update_graph_replay_equals_the_eager_loop(agent):
    """The path bench.py times and learn() runs (mh_ppo.py:222-247): rollout as ONE replayed hipGraph -> `_training_step` (20 optimiser steps:
    the flat weights the captured policy launch reads are rewritten in place, the policy stack is re-packed before the next replay, the Adam
    pass leaves the gradient buffer zeroed) -> replay of the SAME graph -> ...  Four iterations against the step-by-step loop with the same
    per-iteration permutation: every rollout-buffer key, the flat parameters, both Adam moments, the learning rates, the step counts, the
    env's globals / episode sums and the episode statistics are bit-identical."""
    kw = dict(batched=True, fused_sample=True, rollouts=4, train_between=True)
    a = _rollouts_with_split(True, agent, rollout_graph=True, **kw)
    b = _rollouts_with_split(True, agent, rollout_graph=False, **kw)
    assert a.pop("_used_graph_each").tolist() == [False, True, True, True] and not b.pop("_used_graph_each").any()
    a.pop("_used_graph"); b.pop("_used_graph")
    a.pop("_time_outs_seen"); b.pop("_time_outs_seen")
    assert not torch.equal(a["_pflat"], a["_pflat_start"])           # (the weights do move under the graph)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_graph_rollout_is_recaptured_after_a_new_replay_window():
    """`simulator.set_replay()` between two rollouts: the env picks the new window up lazily inside its next step, i.e. after MHPPO has looked
    its cached rollout graph up — the graph's key therefore carries the simulator's replay version (and whether the step kernel is the
    specialised one), a stale graph would replay the OLD window's frame addresses.  Graph vs eager, new window before the fourth of five
    rollouts: bit-identical."""
    kw = dict(batched=True, fused_sample=True, rollouts=5, new_replay_at=3)
    a = _rollouts_with_split(True, "v1", rollout_graph=True, **kw)
    b = _rollouts_with_split(True, "v1", rollout_graph=False, **kw)
    assert a.pop("_used_graph_each").tolist() == [False, True, True, True, True] and not b.pop("_used_graph_each").any()
    a.pop("_used_graph"); b.pop("_used_graph")
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("agent", ["v1", "v2"])
def test_rollout_fused_sampling_equals_sampling_kernel(agent):
    """The agents' default rollout samples in the policy kernel's last epilogue (`pbhc_mlp_fwd_sample` / `pbhc_mlp_fwd_cat`, keyed by a snapshot
    of the step counter + the step index) instead of launching `pbhc_policy_sample` on the live counter: the same Philox keys and arithmetic, so
    every action — and with it every observation, reward and env state of three rollouts — is bit-identical; the log-prob sums its columns in
    another order."""
    a = _rollouts_with_split(True, agent, batched=True, fused_sample=True)
    b = _rollouts_with_split(True, agent, batched=True, fused_sample=False)
    a.pop("_time_outs_seen"); b.pop("_time_outs_seen")
    for k in a:
        if k == "actions_log_prob":
            close(a[k], b[k], 2e-5, "fused sampling log-prob", rtol=1e-6)
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("agent", ["v1", "v2"])
def test_rollout_batched_critic_equals_per_step_critic(agent):
    """The agents' default rollout evaluates the critic ONCE over all T slabs after the loop and adds the time-out bootstrap
    (rewards += gamma * values * time_outs, mh_ppo.py:300-305, ppo_mimic.py:425-431) then — ppo_mimic.PPO on the motion embeddings the per-step
    forwards stored; against the per-step critic everything the critic does not feed is bit-identical, and values / rewards / returns /
    advantages agree to GEMM rounding (different tile shapes at 98 304 rows)."""
    a = _rollouts_with_split(True, agent, batched=True)
    b = _rollouts_with_split(True, agent, batched=False)
    soft = {"values": 2e-5, "rewards": 2e-5, "returns": 1e-4, "advantages": 2e-4}
    seen = bool(a.pop("_time_outs_seen"))
    assert seen or agent == "v2"                     # v1: envs reach the end of the clip inside the rollout, the bootstrap term is exercised
    if agent == "v2":
        soft["actions_log_prob"] = 2e-5              # (its per-step-critic loop samples in pbhc_policy_sample: the log-prob sums its columns in another order)
    for k in a:
        if k in soft:
            close(a[k], b[k], soft[k], "batched critic " + k, rtol=1e-5)
        else:
            assert torch.equal(a[k], b[k]), k


def test_learn_runs_two_iterations():
    from pbhc_amd.agents.mh_ppo import MHPPO

    cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", 256, noise_off=False)
    algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
    algo.setup()
    algo.learn(num_iterations=2)
    torch.cuda.synchronize()
    for p in algo.actor.parameters():
        assert torch.isfinite(p).all()
    log = env.read_log()
    assert np.isfinite(log["reward_mean"])
    # evaluate_policy_steps (mh_ppo.py:702-775): eval mode, every env restarts its clip at t = 0, deterministic actions
    obs = algo.evaluate_policy_steps(3)
    torch.cuda.synchronize()
    assert env.is_evaluating and torch.isfinite(obs["actor_obs"]).all()
    assert int(env.episode_length_buf.max()) <= 4


def test_policy_sample_and_rollout_post():
    """pbhc_policy_sample: log-prob equals torch.distributions.Normal on the sampled action, samples are N(mu, std);
    pbhc_rollout_post: time-out bootstrap, dones and the device-side episode book-keeping (mh_ppo.py:300-323)."""
    from pbhc_amd import _lib

    lib = _lib.lib()
    N, A, R = 4096, 23, 21
    g = torch.Generator(device=DEV).manual_seed(0)
    mu = torch.randn(N, A, device=DEV, generator=g)
    std = 0.2 + torch.rand(A, device=DEV, generator=g)
    value = torch.randn(N, R, device=DEV, generator=g)
    counter = torch.tensor([7.0], dtype=torch.float64, device=DEV)
    act, am, asg = torch.zeros(N, A, device=DEV), torch.zeros(N, A, device=DEV), torch.zeros(N, A, device=DEV)
    lp, vout = torch.zeros(N, device=DEV), torch.zeros(N, R, device=DEV)
    st = _lib.current_stream()
    _lib.check(lib.pbhc_policy_sample(mu.data_ptr(), std.data_ptr(), value.data_ptr(), N, A, R, 1234, counter.data_ptr(), act.data_ptr(), am.data_ptr(),
                                      asg.data_ptr(), lp.data_ptr(), vout.data_ptr(), st))
    torch.cuda.synchronize()
    ref_lp = torch.distributions.Normal(mu, mu * 0 + std).log_prob(act).sum(-1)
    close(lp, ref_lp, 2e-4, "logp", rtol=1e-5)
    assert torch.equal(am, mu) and torch.equal(vout, value) and torch.equal(asg, std.expand(N, A))
    z = (act - mu) / std
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01 and abs(float((z ** 4).mean()) - 3.0) < 0.1
    act2 = torch.zeros_like(act)
    counter.fill_(8.0)
    _lib.check(lib.pbhc_policy_sample(mu.data_ptr(), std.data_ptr(), value.data_ptr(), N, A, R, 1234, counter.data_ptr(), act2.data_ptr(), am.data_ptr(),
                                      asg.data_ptr(), lp.data_ptr(), vout.data_ptr(), st))
    assert float((act2 - act).abs().mean()) > 0.1                      # a new counter gives new draws
    # rollout_post
    rew = torch.randn(N, R, device=DEV, generator=g)
    reset = (torch.rand(N, device=DEV, generator=g) < 0.1).long()
    tout = ((torch.rand(N, device=DEV, generator=g) < 0.5) & (reset > 0))
    cur_r, cur_l = torch.rand(N, device=DEV, generator=g), torch.randint(0, 50, (N,), device=DEV, generator=g).float()
    cr0, cl0 = cur_r.clone(), cur_l.clone()
    out_r = torch.zeros(N, R, device=DEV); dones = torch.zeros(N, 1, dtype=torch.bool, device=DEV)
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    _lib.check(lib.pbhc_rollout_post(rew.data_ptr(), value.data_ptr(), reset.data_ptr(), tout.data_ptr(), N, R, 0.99, out_r.data_ptr(), dones.data_ptr(),
                                     cur_r.data_ptr(), cur_l.data_ptr(), stats.data_ptr(), st))
    torch.cuda.synchronize()
    close(out_r, rew + 0.99 * value * tout.unsqueeze(1), 1e-6, "bootstrap")
    assert torch.equal(dones[:, 0], reset > 0)
    nr, nl = cr0 + rew.sum(-1), cl0 + 1
    d = reset > 0
    close(cur_r, torch.where(d, torch.zeros_like(nr), nr), 1e-5, "cur_reward_sum")
    close(cur_l, torch.where(d, torch.zeros_like(nl), nl), 0, "cur_episode_length")
    ref = torch.stack([nr[d].double().sum(), nl[d].double().sum(), d.double().sum()])
    assert torch.allclose(stats, ref, rtol=1e-6)


def test_observation_noise_is_bounded_uniform_and_only_where_configured():
    """Noise-on vs noise-off step from the same state (reference: helpers.parse_observation, (x + (2U-1) * noise) * scale, helpers.py:128-152):
    elements of noise-free keys are identical, noisy elements differ by at most noise*scale, and over 4096 envs the differences have
    the mean ~0 and the variance (noise*scale)^2 / 3 of a uniform on [-1, 1] — separately in every group that carries the key."""
    N = 4096
    outs = []
    for noise_off in (True, False):
        torch.manual_seed(11)
        cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", N, noise_off=noise_off)
        torch.manual_seed(12)
        env.reset_all()
        obs, _, _, _ = env.step({"actions": torch.zeros(N, env.num_dof, device=DEV)})
        torch.cuda.synchronize()
        outs.append(({k: v.clone() for k, v in obs.items()}, {k: v.clone() for k, v in env.history.items()}, cfg, env.layout))
    (clean, hclean, _, _), (noisy, hnoisy, cfg, L) = outs
    ob = cfg.obs
    for g, keys in ob.obs_dict.items():
        pos = 0
        for key in sorted(keys):
            d = L.obs_dims[key] if key in L.obs_dims else sum(L.obs_dims[k] * n for k, n in ob.obs_auxiliary[key].items())
            diff = (noisy[g][:, pos:pos + d] - clean[g][:, pos:pos + d]).cpu()
            amp = float(ob.noise_scales[key]) * float(ob.obs_scales[key])
            if amp == 0.0 or key in ob.obs_auxiliary:
                if key not in ob.obs_auxiliary:
                    assert float(diff.abs().max()) == 0.0, (g, key)
            else:
                assert float(diff.abs().max()) <= amp * (1 + 1e-5) + 1e-6, (g, key, float(diff.abs().max()), amp)
                assert abs(float(diff.mean())) < 0.05 * amp and abs(float(diff.var()) / (amp * amp / 3.0) - 1.0) < 0.1, (g, key, float(diff.mean()), float(diff.var()), amp)
            pos += d
    k = "dof_pos"                                                # the newest history entry carries its own independent draw
    dh = (hnoisy[k][:, 0] - hclean[k][:, 0]).cpu()
    amp = float(ob.noise_scales[k]) * float(ob.obs_scales[k])
    assert 0 < float(dh.abs().max()) <= amp * (1 + 1e-5)
    da = (noisy["actor_obs"] - clean["actor_obs"]).cpu()
    assert float((dh[:, 0] - da[:, 0]).abs().max()) > 0        # different draws than the actor group's


def test_reset_draws_have_the_reference_distributions():
    """The reset path's IN-KERNEL Philox draws (every other reset test injects the reference's recorded draws): start phase
    `U * motion_len` (sample_time, motion_lib_base.py:486-495 via motion_tracking.py:369-378), kp / kd / rfi-limit scales `U(lo, hi)` and
    rao `U(-lim, lim)` per dof (legged_robot_base.py:599-635), control delay `randint(lo, hi + 1)` (:627-631).  4096 envs are forced to
    time out twice; checked: ranges, first two moments, ten-bin histograms, independence between envs / dofs / consecutive resets, and
    that the same seed reproduces the same draws."""
    N = 4096
    runs = []
    for rep in range(2):
        torch.manual_seed(21)
        cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", N, noise_off=True)
        dr = cfg.domain_rand
        torch.manual_seed(22)
        env.reset_all()
        snaps = []
        for k in range(2):
            env.episode_length_buf = torch.full((N,), int(env.max_episode_length) + 5, device=DEV)       # -> every env times out this step
            _, _, reset, extras = env.step({"actions": torch.zeros(N, env.num_dof, device=DEV)})
            torch.cuda.synchronize()
            assert int(reset.sum()) == N and bool(extras["time_outs"].all())
            snaps.append(dict(start=env.motion_start_times.cpu().clone(), kp=env._kp_scale.cpu().clone(), kd=env._kd_scale.cpu().clone(),
                              rfi=env._rfi_lim_scale.cpu().clone(), rao=env._rao_scale.cpu().clone(), delay=env.action_delay_idx.cpu().clone(),
                              mlen=env.motion_len.cpu().clone()))
        runs.append(snaps)
    for a, b in zip(runs[0], runs[1]):
        for k in a:
            assert torch.equal(a[k], b[k]), f"same seed, different {k}"

    def uniform(x, lo, hi, what):
        x = x.double().flatten()
        n = x.numel()
        assert float(x.min()) >= lo - 1e-6 and float(x.max()) < hi + 1e-6 * max(1.0, abs(hi)), (what, float(x.min()), float(x.max()))
        u = (x - lo) / (hi - lo)
        assert abs(float(u.mean()) - 0.5) < 4.0 * (1.0 / 12.0 / n) ** 0.5 + 1e-4, (what, "mean", float(u.mean()))
        assert abs(float(u.var()) * 12.0 - 1.0) < 0.05 + 6.0 / n ** 0.5, (what, "variance", float(u.var()))
        h = torch.histc(u.float(), bins=10, min=0.0, max=1.0).double() / n
        assert float((h - 0.1).abs().max()) < 5.0 * (0.09 / n) ** 0.5 + 1e-3, (what, "histogram", h.tolist())

    def uncorrelated(x, y, what):
        x, y = x.double().flatten(), y.double().flatten()
        r = float(((x - x.mean()) * (y - y.mean())).mean() / (x.std() * y.std()))
        assert abs(r) < 5.0 / x.numel() ** 0.5, (what, r)

    s0, s1 = runs[0]
    mlen = float(s0["mlen"][0])
    for tag, s in (("first reset", s0), ("second reset", s1)):
        uniform(s["start"], 0.0, mlen, tag + " start phase")
        uniform(s["kp"], dr.kp_range[0], dr.kp_range[1], tag + " kp")
        uniform(s["kd"], dr.kd_range[0], dr.kd_range[1], tag + " kd")
        uniform(s["rfi"], dr.rfi_lim_range[0], dr.rfi_lim_range[1], tag + " rfi limit")
        uniform(s["rao"], -dr.rao_lim, dr.rao_lim, tag + " rao")
        lo, hi = dr.ctrl_delay_step_range
        cnt = torch.bincount(s["delay"] - lo, minlength=hi - lo + 1).double() / N
        assert int(s["delay"].min()) >= lo and int(s["delay"].max()) <= hi
        assert float((cnt - 1.0 / (hi - lo + 1)).abs().max()) < 5.0 * (0.25 / N) ** 0.5, (tag, "delay", cnt.tolist())
        uncorrelated(s["kp"], s["kd"], tag + " kp vs kd")
        uncorrelated(s["kp"], s["rfi"], tag + " kp vs rfi")
        uncorrelated(s["kp"][:, 0], s["kp"][:, 1], tag + " kp dof 0 vs dof 1")
        uncorrelated(s["kp"][:-1], s["kp"][1:], tag + " kp env i vs env i+1")
        uncorrelated(s["start"], s["kp"][:, 0], tag + " start vs kp")
        uncorrelated(s["start"][:-1], s["start"][1:], tag + " start env i vs env i+1")
    uncorrelated(s0["start"], s1["start"], "start phase, consecutive resets")
    uncorrelated(s0["kp"], s1["kp"], "kp, consecutive resets")
    assert not torch.equal(s0["delay"], s1["delay"])


def test_act_bwd_bias_matches_torch():
    """pbhc_act_bwd_bias: dz = dy * act'(saved) and grad_bias = colsum(dz) against torch autograd for ELU / SiLU / ReLU / none, at the
    shapes the update uses (24576 rows; 768-, 128- and 23-wide layers)."""
    import torch.nn.functional as F

    from pbhc_amd import _lib

    lib = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(0)
    for B, n in ((24576, 768), (24576, 128), (24576, 23), (100, 300)):
        scratch = torch.empty(_lib.K["PBHC_ACT_MAX_BLOCKS"] * n, device=DEV)
        for act, fn in ((1, F.elu), (2, F.silu), (3, F.relu), (0, None)):
            z = torch.randn(B, n, device=DEV, generator=g, requires_grad=True)
            dy = torch.randn(B, n, device=DEV, generator=g)
            if fn is None:
                ref_dz, saved = dy, None
            else:
                y = fn(z)
                (ref_dz,) = torch.autograd.grad(y, z, dy)
                saved = z.detach() if act == 2 else y.detach()
            dz = dy.clone()
            gb = torch.zeros(n, device=DEV)
            _lib.check(lib.pbhc_act_bwd_bias(dz.data_ptr(), None if saved is None else saved.data_ptr(), B, n, act, dz.data_ptr(), gb.data_ptr(),
                                             scratch.data_ptr(), _lib.current_stream()))
            torch.cuda.synchronize()
            close(dz, ref_dz, 1e-6, f"dz act={act} {B}x{n}", rtol=1e-5)
            close(gb, ref_dz.double().sum(0).float(), 2e-3 if B > 1000 else 1e-4, f"grad_bias act={act} {B}x{n}", rtol=1e-5)


def test_fused_mlp_keeps_autograd_accumulation():
    """the fused MLP backward may STORE weight / bias gradients into `.grad` (the agents' flat buffer, zeroed before every
    backward) — but only for a stack its owner declared `grad_direct`, and only with one live application.  User code (no declaration), a
    stack applied twice in one graph, and two backwards without zeroing must all behave like plain autograd (`grad += ...`)."""
    from pbhc_amd.agents import fused_mlp
    from pbhc_amd.agents.modules import BaseModule

    def make(declare):
        torch.manual_seed(3)
        m = BaseModule({"o": 40}, {"input_dim": ["o"], "output_dim": [7], "layer_config": {"type": "MLP", "hidden_dims": [64, 32], "activation": "ELU"}}).to(DEV)
        flat = torch.zeros(sum(p.numel() for p in m.parameters()), device=DEV)
        o = 0
        for p in m.parameters():
            p.grad = flat[o:o + p.numel()].view_as(p)
            o += p.numel()
        if declare:
            for b in m.modules():
                if isinstance(b, BaseModule):
                    fused_mlp.grad_direct(b.module)
        return m, flat

    g = torch.Generator(device=DEV).manual_seed(0)
    x1, x2 = torch.randn(512, 40, device=DEV, generator=g), torch.randn(512, 40, device=DEV, generator=g)
    ref, rflat = make(False)
    ref._fused = False                                     # plain nn.Sequential + autograd
    (ref(x1).square().sum() + ref(x2).sum()).backward()
    want_twice = rflat.clone()
    rflat.zero_()
    ref(x1).square().sum().backward()
    want_once = rflat.clone()
    for declare in (False, True):
        m, flat = make(declare)
        assert m._fused
        # (1) applied twice in ONE graph
        (m(x1).square().sum() + m(x2).sum()).backward()
        close(flat, want_twice, 2e-4, f"declare={declare}: module applied twice in one graph", rtol=1e-4)
        def zero():                                        # the owner's pattern: zero the buffer, tell the declared stacks (MHPPO._zero_grads)
            flat.zero_()
            if declare:
                for b in m.modules():
                    if isinstance(b, BaseModule):
                        fused_mlp.grads_zeroed(b.module)

        # (2) one application after zeroing (the agents' pattern): the direct store when declared
        zero()
        m(x1).square().sum().backward()
        close(flat, want_once, 2e-4, f"declare={declare}: single application", rtol=1e-4)
        # (3) a second backward WITHOUT zeroing accumulates — undeclared stacks never overwrite, and a declared one stores only on the
        # FIRST backward after its owner's zeroing (an auxiliary loss / gradient accumulation between two zeroings)
        m(x1).square().sum().backward()
        close(flat, 2.0 * want_once, 4e-4, f"declare={declare}: two backwards without zeroing", rtol=1e-4)
        # (3b) a frozen parameter with a .grad view receives no store
        if declare:
            zero()
            lin0 = next(q for q in m.module if isinstance(q, torch.nn.Linear))
            lin0.weight.requires_grad_(False)
            m(x1).square().sum().backward()
            assert float(lin0.weight.grad.abs().max()) == 0.0
            lin0.weight.requires_grad_(True)
        # (4) two graphs alive together, backward one after the other
        zero()
        ya, yb = m(x1).square().sum(), m(x2).sum()
        ya.backward(); yb.backward()
        close(flat, want_twice, 2e-4, f"declare={declare}: two live graphs", rtol=1e-4)
        # ... and the direct path is back afterwards
        zero()
        m(x1).square().sum().backward()
        close(flat, want_once, 2e-4, f"declare={declare}: single application again", rtol=1e-4)


def test_adam_clip_matches_torch_adam_and_adamw():
    """pbhc_adam_clip over a flat segment == clip_grad_norm_ + torch.optim.Adam / AdamW (decoupled weight decay) for 5 steps."""
    from pbhc_amd import _lib

    lib = _lib.lib()
    n = 300_000
    for wd, opt_cls in ((0.0, torch.optim.Adam), (0.01, torch.optim.AdamW)):
        g = torch.Generator(device=DEV).manual_seed(1)
        p0 = torch.randn(n, device=DEV, generator=g)
        ref = torch.nn.Parameter(p0.clone())
        opt = opt_cls([ref], lr=1e-3, **({"weight_decay": wd} if wd else {}))
        p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        lr, step = torch.tensor([1e-3], device=DEV), torch.zeros(1, device=DEV)
        scratch, norm = torch.zeros(512, dtype=torch.float64, device=DEV), torch.zeros(1, device=DEV)
        for it in range(5):
            grad = torch.randn(n, device=DEV, generator=g) * (0.01 if it % 2 else 1.0)        # with and without clipping
            ref.grad = grad.clone()
            tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
            opt.step()
            gbuf = grad.clone()
            _lib.check(lib.pbhc_adam_clip(p.data_ptr(), gbuf.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr.data_ptr(), step.data_ptr(), 1.0, 0.9, 0.999, 1e-8,
                                          wd, scratch.data_ptr(), norm.data_ptr(), _lib.current_stream()))
            torch.cuda.synchronize()
            assert abs(float(norm) - float(tn)) < 1e-4 * float(tn)
            close(p, ref.detach(), 2e-6, f"{opt_cls.__name__} step {it}", rtol=2e-6)
        assert float(step) == 5.0


def test_motion_lib_max_len_crops_match_oracle_on_the_cropped_clips():
    """load_motions(max_len) (motion_lib_base.py:420-434): every env slot plays its own random crop of at most max_len frames, FK and the
    velocity filter run on the crop.  The HIP library's per-slot tables against the oracle library built from the same crops."""
    from oracle.motion_lib import MotionLib as OML
    from pbhc_amd.motion_lib import MotionLib
    from pbhc_amd.skeleton import Skeleton

    walk, horse = clip_from_env_golden(load_env_golden("walk")), clip_from_env_golden(load_env_golden("horse"))
    horse = {k: v for k, v in horse.items() if k != "contact_mask"}
    short = dict(pose_aa=walk["pose_aa"][:30], root_trans_offset=walk["root_trans_offset"][:30], fps=walk["fps"])        # shorter than max_len: kept whole
    clips = [walk, horse, short]
    N, K = 9, 48
    sk = Skeleton.from_json(os.path.join(GOLDEN, "skeleton_g1_23dof_lock_wrist_fitmotionONLY.json"))
    ml = MotionLib(sk, clips, N, DEV, max_len=K)
    starts = torch.tensor([0, 5, 0, 74, 162, 0, 33, 100, 0])
    ml.load_motions(random_sample=False, max_len=K, crop_starts=starts)
    assert ml.slot_clip.tolist() == [i % 3 for i in range(N)] and ml.slot_table.tolist() == list(range(N))
    crops = []
    for i in range(N):
        c = clips[i % 3]
        F = c["pose_aa"].shape[0]
        a, b = (0, F) if F < K else (int(starts[i]), int(starts[i]) + K)
        crops.append(dict(pose_aa=c["pose_aa"][a:b], root_trans_offset=c["root_trans_offset"][a:b], fps=c["fps"]))
    assert ml.num_frames.tolist() == [c["pose_aa"].shape[0] for c in crops] == [48, 48, 30] * 3
    oml = OML(skel_from_golden(), crops)
    assert torch.allclose(ml.get_motion_length().cpu(), oml.motion_len, atol=1e-6)
    gen = torch.Generator().manual_seed(0)
    ids = torch.arange(N).repeat(8)
    times = torch.rand(ids.shape[0], generator=gen) * oml.motion_len[ids] * 1.1 - 0.05          # incl. t < 0 and t > len
    off = torch.randn(ids.shape[0], 3, generator=gen)
    res = ml.get_motion_state(ids.to(DEV), times.to(DEV), off.to(DEV))
    ref = oml.get_motion_state(ids, times, off)
    for k in ["root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos_t", "rg_rot_t", "body_vel_t", "body_ang_vel_t"]:
        extra = ANGVEL if "ang_vel" in k else (SLERP if "rot" in k else {})
        close(res[k], ref[k], 5e-5, f"crop:{k}", **extra)
    # a second load re-draws the crops in place (same device buffers: the step kernel keeps its pointers)
    ptr = ml.frames.data_ptr()
    ml.load_motions(random_sample=True, max_len=K)
    assert ml.frames.data_ptr() == ptr and ml.table.frames == ptr and all(0 <= int(a) <= 210 - K for a in ml.crop_starts)


def test_general_tracking_env_with_motion_max_len_runs_and_resamples():
    cfg, env = build_hip_env("v2_g1_23dof_student.yaml", 64, general=True, overrides={"robot.motion.motion_max_len": 40})
    assert env.max_len == 40 and env._motion_lib.max_len == 40
    obs = env.reset_all()
    assert torch.allclose(env.motion_len, torch.full_like(env.motion_len, 39 / 30.0), atol=1e-6)
    first = env._motion_lib.crop_starts.clone()
    for _ in range(45):                                  # past the end of the 40-frame crops: motion-end resets
        obs, rew, reset, extras = env.step({"actions": torch.zeros(64, env.num_dof, device=DEV)})
    assert all(torch.isfinite(v).all() for v in obs.values()) and torch.isfinite(rew).all()
    assert float(env.read_log()["terminate_by_motion_end"]) >= 0.0
    env.resample_motion()
    assert not torch.equal(env._motion_lib.crop_starts, first)
    obs, rew, reset, extras = env.step({"actions": torch.zeros(64, env.num_dof, device=DEV)})
    assert all(torch.isfinite(v).all() for v in obs.values())


def test_minibatch_gather_is_one_launch_and_equals_indexing():
    """`RolloutStorage.mini_batch_generator` shuffles every key by one permutation (data_utils.py:134-152: `flatten(0, 1)[indices]`); the f32 keys
    — contiguous buffers and the 128-byte-padded observation slabs alike — go through ONE `pbhc_gather_rows` launch, other dtypes through torch
    indexing: bit-identical to indexing, rows of every width class (16- / 8- / 4-byte pieces)."""
    from pbhc_amd.agents.modules import RolloutStorage

    T, N = 6, 130
    st = RolloutStorage(N, T, DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    for key, shape, kw in [("actor_obs", (380,), dict(pad_rows=True, tail_slab=True)), ("critic_obs", (630,), dict(pad_rows=True, tail_slab=True)),
                           ("odd", (23,), {}), ("scalar", (1,), {}), ("even", (6,), {}), ("wide4", (64,), {}), ("dones", (1,), dict(dtype=torch.bool))]:
        st.register_key(key, shape=shape, **kw)
        buf = getattr(st, key)
        if buf.dtype == torch.bool:
            buf.copy_(torch.rand(buf.shape, device=DEV, generator=g) > 0.5)
        else:
            buf.copy_(torch.randn(buf.shape, device=DEV, generator=g))
    idx = torch.randperm(T * N, device=DEV, generator=g)
    keys = list(st.stored_keys)
    batches = list(st.mini_batch_generator(3, 2, keys=keys, indices=idx))
    assert len(batches) == 6
    mb = T * N // 3
    for e in range(2):
        for i in range(3):
            b = batches[3 * e + i]
            for k in keys:
                ref = getattr(st, k).flatten(0, 1)[idx][i * mb:(i + 1) * mb]
                assert b[k].is_contiguous() or b[k].shape[0] == mb
                assert torch.equal(b[k], ref), k
