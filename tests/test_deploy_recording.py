"""Known-answer check against a rollout the reference itself recorded: the 877-wide `actor_obs` rows its deploy stack fed to
the exported student policy (tests/golden/deploy_student23_recording.npz, from logs/MotionTracking/phuma_student/motions/…, see
oracle/ref_harness/gen_deploy_recording_fixture.py).  The env — here the oracle, in tests/test_gpu_parity_v2.py the HIP kernel — is
driven with the recorded actions and the robot states recovered from those rows and must reproduce every row.

Tolerances (absolute, fp32 observations of magnitude <= ~3): 5e-6 on everything assembled from the robot state and the history;
3e-4 on `next_step_ref_motion` (reference-motion interpolation: the deploy stack evaluates the clip at float64 (k+1)*dt rounded to fp32,
the env at fp32 (k+1)*dt — a time difference of ~1e-7 s times filtered velocities)."""
import os

import numpy as np
import torch

from oracle.fk import sim_fk
from tests.helpers import GOLDEN, deploy_layout_slices, deploy_recording_states, fresh_episode_state
from tests.test_oracle_env_v2 import build_oracle_v2

TOL = {"next_step_ref_motion": 3e-4}
HIST_FROM = 11            # the history holds 10 steps; before that it still contains the unrecorded step-0 entry


def check_rows(got, want, r, worst):
    for k, sl in deploy_layout_slices().items():
        if k == "history_actor" and r < HIST_FROM:
            continue
        e = float(np.abs(got[sl] - want[sl]).max())
        worst[k] = max(worst.get(k, 0.0), e)
        assert e <= TOL.get(k, 5e-6), f"row {r} {k}: {e}"


def test_oracle_reproduces_recorded_deploy_observations():
    rec = dict(np.load(os.path.join(GOLDEN, "deploy_student23_recording.npz")))
    g, env, skel = build_oracle_v2("student23", "v2_g1_23dof_student.yaml", "g1_23dof")
    N = env.N
    assert abs(1.0 / float(rec["fps"]) - env.dt) < 1e-9 and np.allclose(rec["motion_times"], (np.arange(141) + 1) * env.dt, atol=1e-5)
    env.load_state(fresh_episode_state(env))
    env.env_origins = torch.zeros(N, 3)
    root, qp, qv = deploy_recording_states(rec, env.ml, env.default_dof_pos.reshape(-1)[: env.D], env.cfg.obs.obs_scales, env.dt)
    worst = {}
    for r in range(1, rec["actor_obs"].shape[0]):
        frame = dict(root=root[r - 1][None].repeat(N, 1), dof_pos=qp[r - 1][None].repeat(N, 1), dof_vel=qv[r - 1][None].repeat(N, 1),
                     contact=torch.zeros(N, env.B, 3))
        body = sim_fk(skel, frame["root"], frame["dof_pos"], frame["dof_vel"])
        obs, rew, reset, extras = env.step(torch.from_numpy(rec["action"][r - 1])[None].repeat(N, 1), frame, body)
        assert not reset.any(), f"row {r}: the recorded robot tracks the clip, no termination expected"
        assert obs["actor_obs"].shape[1] == 877 and torch.equal(obs["actor_obs"][0], obs["actor_obs"][N - 1])
        check_rows(obs["actor_obs"][0].numpy(), rec["actor_obs"][r], r, worst)
        # the student's history input is the same 740 values (obs_log files of the reference: prop_history == history_actor slice)
        assert torch.equal(obs["prop_history"][0], obs["actor_obs"][0][deploy_layout_slices()["history_actor"]])
    assert worst["roll_pitch"] > 0 and worst["actions"] == 0.0


# ---- second recording: per-step log with the exact robot state each observation was built from ------------------------------------
WALK_CLIP = "tests/golden/clips/g1_walk_45cms_23dof.npz"
LAST_ROW = 199           # (row+1)*dt = 4.00 s < clip length 4.033 s: no motion-end reset; the 20 future targets run past the clip end (clamped)


def student_walk_oracle(N):
    from oracle.env_v2 import GeneralTrackingOracle
    from oracle.motion_lib import MotionLib
    from tests.helpers import clip_from_env_golden, fixture_config, load_env_golden, skel_from_golden

    skel = skel_from_golden("g1_23dof")
    cfg = fixture_config("v2_g1_23dof_student.yaml", N)
    nlink = len(cfg.domain_rand.get("randomize_link_body_names", [])) or 22
    dr = dict(base_com_bias=torch.zeros(N, 3), link_mass_scale=torch.ones(N, nlink), friction_coeffs=torch.ones(N, 1, 1), base_mass_scale=torch.ones(N, 1))
    env = GeneralTrackingOracle(cfg, skel, MotionLib(skel, [clip_from_env_golden(load_env_golden("walk"))]), N, dr)
    env.env_origins = torch.zeros(N, 3)
    env.ref_init_yaw = 0.0
    return env, skel


def sim2sim_log_frames(log, oml):
    """Replay frames of rows 1..LAST_ROW: the logged base quaternion / body-frame angular velocity / joint state; the base position
    (not logged, not observed by the actor) is put on the reference so that the height terminations stay quiet."""
    from oracle import rotations as R

    rows = range(1, LAST_ROW + 1)
    q = torch.from_numpy(log["robot_quat_xyzw"][1:LAST_ROW + 1])
    w_world = R.quat_rotate(q, torch.from_numpy(log["robot_ang_vel"][1:LAST_ROW + 1]))
    ref = oml.get_motion_state(torch.zeros(len(rows), dtype=torch.long), torch.from_numpy(log["motion_time"][1:LAST_ROW + 1]).float())
    root = torch.cat([ref["root_pos"], q, torch.zeros(len(rows), 3), w_world], dim=1)
    return root, torch.from_numpy(log["robot_dof_pos"][1:LAST_ROW + 1]), torch.from_numpy(log["robot_dof_vel"][1:LAST_ROW + 1])


def check_log_row(obs, log, r, worst):
    check_rows(obs["actor_obs"][0].numpy(), log["actor_obs"][r], r, worst)
    e = float(np.abs(obs["future_motion_targets"][0].numpy() - log["future_motion_targets"][r]).max())
    worst["future_motion_targets"] = max(worst.get("future_motion_targets", 0.0), e)
    assert e <= TOL["next_step_ref_motion"], f"row {r} future_motion_targets: {e}"


def test_oracle_reproduces_sim2sim_log():
    """204 consecutive control steps of the reference's deploy stack on g1_walk_45cms with the observation-time robot state logged:
    nothing is recovered from the observation here, so `anchor_ref_rot`, `roll_pitch`, `base_ang_vel`, `dof_pos`, `dof_vel` are
    independent checks too, and `future_motion_targets` (20 steps x 30) is covered."""
    log = dict(np.load(os.path.join(GOLDEN, "deploy_sim2sim_log_walk.npz")))
    N = 4
    env, skel = student_walk_oracle(N)
    assert abs(float(log["dt"]) - env.dt) < 1e-12 and np.allclose(log["motion_time"], (np.arange(204) + 1) * env.dt)
    env.load_state(fresh_episode_state(env))
    root, qp, qv = sim2sim_log_frames(log, env.ml)
    worst = {}
    for r in range(1, LAST_ROW + 1):
        frame = dict(root=root[r - 1][None].repeat(N, 1), dof_pos=qp[r - 1][None].repeat(N, 1), dof_vel=qv[r - 1][None].repeat(N, 1),
                     contact=torch.zeros(N, env.B, 3))
        obs, rew, reset, extras = env.step(torch.from_numpy(log["action"][r - 1])[None].repeat(N, 1), frame, sim_fk(skel, frame["root"], frame["dof_pos"], frame["dof_vel"]))
        assert not reset.any(), f"row {r}"
        check_log_row(obs, log, r, worst)
    assert set(worst) == set(deploy_layout_slices()) | {"future_motion_targets"}
