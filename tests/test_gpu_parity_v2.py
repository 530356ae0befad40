"""GPU parity of the general-tracking (KungfuBot2, MODE 1) step: the HIP path through the C ABI against traces of the reference's
own LeggedRobotGeneralTracking.step (tests/golden/env_v2_*.npz).  Tolerances as in test_gpu_parity.py; the angle of a near-identity
quaternion difference, 2 acos(w), is ill-conditioned in fp32 (d angle = 2 dw / sin(angle/2)), which the angle-based rewards inherit."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import GOLDEN, build_hip_env, load_state_into_hip_env, state_dict_from_golden
from tests.test_gpu_parity import ANGVEL, SLERP, angvel_tol, close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = [("student23", "v2_g1_23dof_student.yaml", "g1_23dof"), ("teacher29", "v2_g1_29dof_teacher.yaml", "g1_29dof")]


@pytest.mark.parametrize("tag,cfgname,robot", CASES)
def test_general_tracking_step_matches_reference_trace(tag, cfgname, robot):
    """The fused HIP step replays the reference's own general-tracking trace.  Observation rows: every element against the bound ITS
    operands give it (`_conditioned_obs_tolerances`: 3e-5 + the conditioning of the frame pair a slerp / a table angular velocity came
    from) — round 3 allowed 2 % of a row's elements up to 1.2e-3.  The frame pairs are those the CPU oracle blends when it replays the same
    trace next to the kernel (the oracle itself is held to the trace at 2e-5 by tests/test_oracle_env_v2.py)."""
    from oracle.fk import sim_fk
    from tests.test_gpu_parity import trace_slerp_bounds
    from tests.test_oracle_env_v2 import build_oracle_v2

    g, orc, oskel = build_oracle_v2(tag, cfgname, robot)
    orc.slot_clip = torch.zeros(orc.N, dtype=torch.long) if not hasattr(orc, "slot_clip") else orc.slot_clip
    T, N, D = g["actions_in"].shape
    cfg, env = build_hip_env(cfgname, N, general=True, overrides={"domain_rand.push_robots": False})
    assert env.reward_names == list(g["reward_names"])
    assert env.key_body_id == list(g["key_body_id"]) and env.anchor_index == int(g["anchor_index"])
    load_state_into_hip_env(env, state_dict_from_golden(g), g)
    dev = env.device
    tg = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    env.simulator.set_replay(tg(g["replay_root"]), tg(g["replay_dof_pos"]), tg(g["replay_dof_vel"]), tg(g["replay_contact"]))
    for k in range(T):
        st = lambda name, dt=torch.float32: tg(g["step__state__" + name][k]).to(dt)
        env.set_injected_draws(u_rfi=tg(g["step__u_rfi"][k]), start_time=st("motion_start_times"), kp=st("kp_scale"), kd=st("kd_scale"),
                               rfi_lim=st("rfi_lim_scale"), rao=st("rao_scale"), delay=st("action_delay_idx", torch.long))
        obs, rew, reset, extras = env.step({"actions": tg(g["actions_in"][k])})
        torch.cuda.synchronize()
        # the oracle on the same step: which frame pairs were blended (orc.fut_times), for the conditioned bounds
        frame = dict(root=torch.from_numpy(g["replay_root"][k]), dof_pos=torch.from_numpy(g["replay_dof_pos"][k]),
                     dof_vel=torch.from_numpy(g["replay_dof_vel"][k]), contact=torch.from_numpy(g["replay_contact"][k]))
        sto = lambda name: torch.from_numpy(g["step__state__" + name][k])
        samp = dict(motion_start_times=sto("motion_start_times"), kp_scale=sto("kp_scale"), kd_scale=sto("kd_scale"),
                    rfi_lim_scale=sto("rfi_lim_scale"), rao_scale=sto("rao_scale"), action_delay_idx=sto("action_delay_idx"))
        orc.step(torch.from_numpy(g["actions_in"][k]), frame, sim_fk(oskel, frame["root"], frame["dof_pos"], frame["dof_vel"]),
                 u_rfi=torch.from_numpy(g["step__u_rfi"][k]), reset_samples=samp)
        w = f"{tag} step {k}: "
        assert torch.equal(reset.cpu(), torch.from_numpy(g["step__reset_buf_out"][k])), w + "reset_buf"
        assert torch.equal(extras["time_outs"].cpu(), torch.from_numpy(g["step__time_outs"][k])), w + "time_outs"
        close(extras["ref_body_pos_extend"], g["step__ref_body_pos_extend"][k], 2e-5, w + "ref_body_pos_extend")
        close(rew, g["step__rew_buf"][k], 3e-5, w + "rew_buf", rtol=2e-4)
        assert set(obs.keys()) == {"actor_obs", "priv_obs", "future_motion_targets", "prop_history"}
        ref_obs = {ok: torch.from_numpy(g["step__obs__" + ok][k]) for ok in obs}
        tols = _conditioned_obs_tolerances(cfg, env, orc, orc.ml, ref_obs)
        for ok in obs:
            close(obs[ok], ref_obs[ok], tols[ok], w + ok)
        for name in ["torques", "last_contacts", "actions", "last_actions", "action_queue", "motion_len", "end_time_ratio_buf", "contacts_filt",
                     "last_dof_vel", "motion_start_times"]:
            close(getattr(env, name), g["step__state__" + name][k], 3e-5, w + "state " + name)
        close(env.simulator.dof_pos, g["step__state__dof_pos"][k], 3e-5, w + "dof_pos")
        close(env.simulator.dof_vel, g["step__state__dof_vel"][k], 3e-5, w + "dof_vel", rtol=1e-4)
        gs = lambda name, j: torch.from_numpy(g["state0__" + name] if j < 0 else g["step__state__" + name][j])
        _, root_tol = trace_slerp_bounds(orc.ml, gs("episode_length_buf", k - 1), gs("motion_start_times", k - 1), gs("episode_length_buf", k),
                                         gs("motion_start_times", k), torch.from_numpy(g["step__reset_buf_out"][k]), float(env.dt), 3e-5)
        rs_tol = torch.full((N, 10), 3e-5)
        rs_tol[:, 3:7] = root_tol.expand(-1, 4)
        close(env.simulator.robot_root_states[:, :10], g["step__state__root_states"][k][:, :10], rs_tol, w + "root_states", rtol=3e-5)
        # a reset writes the looked-up root angular velocity of the reference motion: bounded by ITS conditioning (angvel_tol), everything else 3e-5
        ref_w = torch.from_numpy(g["step__state__root_states"][k][:, 10:])
        close(env.simulator.robot_root_states[:, 10:], ref_w, angvel_tol(ref_w, float(env._motion_lib._motion_dt[0]), k=32.0, base=3e-5), w + "root ang vel", rtol=3e-5)
        assert torch.equal(env.episode_length_buf.cpu(), torch.from_numpy(g["step__state__episode_length_buf"][k]))
        for name, col in env.episode_sums.items():
            close(col, g["step__state__sum__" + name][k], 3e-5, w + "sum " + name, rtol=2e-4)
        for name, view in env.history.items():
            close(view, g["step__state__hist__" + name][k], 3e-5, w + "hist " + name)
        log = env.read_log()
        for lk in ["terminate_by_ref_pos_z", "terminate_by_ref_ori", "terminate_by_body_z", "terminate_by_time_out", "terminate_by_motion_end",
                   "key_body_diff_norm", "local_key_body_diff_norm", "local_upper_body_diff_norm", "joint_pos_diff_norm", "action_clip_frac"]:
            close(torch.tensor(log[lk]), g["step__log__" + lk][k], 1e-4, w + "log " + lk)


def _v2_algo(N, overrides=None, noise_off=True):
    from pbhc_amd.agents.ppo_mimic import PPO

    ov = {"domain_rand.push_robots": False}
    ov.update(overrides or {})
    cfg, env = build_hip_env("v2_g1_29dof_teacher.yaml", N, general=True, overrides=ov, noise_off=noise_off)
    algo = PPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
    algo.setup()
    return cfg, env, algo


def test_update_with_inputs_assembled_in_place_equals_the_concatenations(monkeypatch):
    """ppo_mimic's update lays the stacks' observation columns out once per update and copies only the encoder outputs per optimiser step
    (`_assemble_inputs`, fused_mlp._FusedMLPInto) instead of concatenating [observations | encoder outputs] every step (ppo_mimic.py:596-630 via
    agent_modules.py:118-128): the GEMMs see the same operands — weights, Adam moments and learning rate after a rollout + update are bit-identical."""
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("PBHC_ASSEMBLE_INPUTS", mode)
        torch.manual_seed(5)
        np.random.seed(5)
        cfg, env, algo = _v2_algo(512, noise_off=False)
        algo._train_mode()
        obs = env.reset_all()
        algo.storage.clear()
        algo._rollout_step(obs)
        n = algo.storage.num_envs * algo.storage.num_transitions_per_env
        algo._training_step(indices=torch.randperm(n, generator=torch.Generator().manual_seed(3)).to(DEV))
        torch.cuda.synchronize()
        assert (algo.__dict__.get("_xin") is not None) == (mode == "1")
        outs.append([algo._pflat.clone(), algo._lr.clone()] + [m.clone() for m in algo._mflat] + [v.clone() for v in algo._vflat])
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_ppo_mimic_update_matches_reference():
    """One _training_step + one _training_step_dagger of pbhc_amd PPO on the reference's rollout buffer, initial weights and
    permutations reproduce the reference's updated weights, losses and learning rate (tests/golden/ppo_v2.npz)."""
    from tests.helpers import PPO_V2_NARROW

    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v2.npz")).items()}
    N = g["st__actions"].shape[1]
    cfg, env, algo = _v2_algo(N, PPO_V2_NARROW)
    algo.alg.load_state_dict({k[len("w0__"):]: v for k, v in g.items() if k.startswith("w0__")}, strict=True)
    for k in algo.storage.stored_keys:
        getattr(algo.storage, k).copy_(g["st__" + k].to(DEV))
    algo._train_mode()
    algo.counter = int(g["counter0"])
    # stored rollout quantities are functions of the stored observations and the initial weights
    with torch.no_grad():
        b = {k: getattr(algo.storage, k).flatten(0, 1) for k in ["actor_obs", "priv_obs", "future_motion_targets", "prop_history"]}
        mu, value, _ = algo._forward(b, hist_encoding=False)
        close(mu, g["st__action_mean"].flatten(0, 1), 2e-5, "mu")
        close(value, g["st__values"].flatten(0, 1), 2e-5, "value")
        last = {k[len("last__"):]: v.to(DEV) for k, v in g.items() if k.startswith("last__")}
        close(algo.alg.act_inference(last, hist_encoding=True), g["infer_hist"], 2e-5, "act_inference(hist)")
        algo._compute_returns(last)
        close(algo.storage.returns, g["st__returns"], 2e-5, "returns")
        close(algo.storage.advantages, g["st__advantages"], 5e-5, "advantages")
    loss = algo._training_step(indices=g["perm1"].to(DEV))
    torch.cuda.synchronize()
    for k in ["Value", "Surrogate", "Entropy", "priv_reg_loss"]:
        assert abs(float(loss[k]) - float(g["loss1__" + k])) < 2e-4 * max(1.0, abs(float(g["loss1__" + k]))), (k, float(loss[k]), float(g["loss1__" + k]))
    assert abs(float(algo._lr[0]) - float(g["lr1"])) < 1e-9
    # AdamW moves a weight by ~lr per step whatever the gradient's size (see test_mhppo_update_matches_reference)
    for k, v in algo.alg.state_dict().items():
        ref = g["w1__" + k]
        close(v, ref, 2e-4, "w1 " + k, rtol=2e-4)
        assert float((v.cpu() - ref).norm() / ref.norm().clamp(min=1e-6)) < 1e-4, k
    assert torch.equal(algo.alg.state_dict()["actor_module.history_encoder.encoder.0.weight"].cpu(), g["w0__actor_module.history_encoder.encoder.0.weight"])
    loss = algo._training_step_dagger(indices=g["perm2"].to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss["hist_latent_loss"]) - float(g["loss2__hist_latent_loss"])) < 2e-4
    for k, v in algo.alg.state_dict().items():
        ref = g["w2__" + k]
        close(v, ref, 2e-4, "w2 " + k, rtol=2e-4)
        assert float((v.cpu() - ref).norm() / ref.norm().clamp(min=1e-6)) < 1e-4, k
    assert algo.counter == int(g["counter2"])


def test_ppo_mimic_checkpoint_roundtrip_uses_reference_keys(tmp_path):
    g = np.load(os.path.join(GOLDEN, "ppo_v2.npz"))
    cfg, env, algo = _v2_algo(16)
    p = str(tmp_path / "model_0.pt")
    algo.save(p, infos={"x": 1})
    d = torch.load(p, map_location="cpu", weights_only=False)
    assert set(d.keys()) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"}
    assert list(d["model_state_dict"].keys()) == [k[len("w0__"):] for k in g.files if k.startswith("w0__")]
    assert d["optimizer_state_dict"]["param_groups"][0]["weight_decay"] == 0.01
    assert algo.load(p) == {"x": 1}
    assert algo.inference_model["actor"] is algo.alg.actor
    # deploy export of the encoder policy (inference_helpers.py:95-138): three named inputs, graph == the GPU actor (unfolded-window GEMMs)
    from pbhc_amd.utils import inference_helpers as ih
    from pbhc_amd.utils import onnx_lite

    ex = algo.get_example_obs()
    file = ih.export_policy_and_encoder_as_onnx(algo.inference_model, str(tmp_path), "model_0.onnx", ex)
    m = onnx_lite.read_model(file)
    feeds = {n: ex[n][:1].cpu().numpy() for n, _ in m["inputs"]}
    with torch.no_grad():
        want = algo.alg.act_inference({k: ex[k][:1] for k in ex}, hist_encoding=True).cpu().numpy()
    assert np.abs(onnx_lite.run(m, feeds)[0] - want).max() <= 1e-5


def test_ppo_mimic_learn_runs_two_iterations():
    cfg, env, algo = _v2_algo(256, noise_off=False)
    algo.learn(num_iterations=2)                     # iteration 0 also runs the DAgger step (0 % dagger_update_freq == 0)
    torch.cuda.synchronize()
    for p in algo.alg.parameters():
        assert torch.isfinite(p).all()
    log = env.read_log()
    assert np.isfinite(log["reward_mean"])
    obs = algo.evaluate_policy_steps(3)              # ppo_mimic.py:880-975: history-encoder inference path
    torch.cuda.synchronize()
    assert env.is_evaluating and all(torch.isfinite(v).all() for v in obs.values())


def test_general_tracking_multi_clip_matches_oracle():
    """Mixed library (BASELINE configs[2]): 3 synthetic clips of different lengths, slots cycling through them, 512 envs — the HIP step
    against the oracle on identical replay tensors, with resets (per-slot motion lengths) and futures running past clip ends."""
    _multi_clip_vs_oracle(512, 4, 3)


def test_general_tracking_full_size_library_matches_oracle():
    """BASELINE configs[2] at its full size: 4096 envs, G1 29-DoF, the 256-clip synthetic library `bench.py --workload v2_teacher29
    --clips 256` measures (the AMASS / LAFAN sets are not shipped), per-env clip and phase — two control steps of the HIP step against
    the oracle on identical replay tensors (the oracle's Python FK of 256 clips and two 4096-env general-tracking steps: about a minute)."""
    _multi_clip_vs_oracle(4096, 2, 256, library_seed=7)


def _key_slices(cfg, env, group):
    """(key, start, width) of every observation key inside its group row (sorted-key concatenation, legged_robot_base.py:787-793; future
    keys list their per-step dim, ppo_mimic.py:206-216)"""
    S = int(env._c.future_num_steps)
    dims = cfg.obs.obs_dims if isinstance(cfg.obs.obs_dims, dict) else {k: v for d_ in cfg.obs.obs_dims for k, v in d_.items()}
    out, pos = [], 0
    for key in sorted(cfg.obs.obs_dict[group]):
        kk = key[:-4] if key.endswith("_raw") else key
        if kk in dims:
            n = int(dims[kk]) * (S if kk.startswith("future_motion_") and S else 1)
        else:                                                    # auxiliary (history) keys
            n = sum(int(dims[hk]) * int(cnt) for hk, cnt in cfg.obs.obs_auxiliary[kk].items())
        out.append((kk, pos, n))
        pos += n
    return out


def _conditioned_obs_tolerances(cfg, env, orc, oml, o_obs, base=3e-5):
    """Elementwise bounds for the general-tracking observations of a multi-clip library, each from the conditioning of ITS operands
    (the 512-env and 4096-env x 256-clip tests used a blanket 1e-4 with 0.2 % of the elements allowed up to 2e-3 before round 3):
      * every key: `base` = 3e-5 absolute + 3e-5 relative, the bound of the small reference traces;
      * keys that are DIFFERENCES OF WORLD POSITIONS (local key-body positions of robot and reference: x - x_anchor with both operands
        at the env origin, up to ~400 m at 4096 envs, general_tracking.py:739-782 keeps that order): + 8 ulp(|origin|);
      * keys rotated by a slerp'ed quaternion (roll / pitch, base-frame velocities, anchor-local key bodies): + twice the relative
        scale error of that slerp (`slerp_scale_bound`: k ulp / sin^2(half angle) of the frame pair — the reference's slerp does not
        return a unit quaternion for slowly turning pairs — or the fall-back jump inside the flip zone) times the rotated vector's length;
      * keys made of the table's ANGULAR velocity (base yaw / angular velocity): + the table's own conditioning `angvel_tol`
        (4 dw / (omega dt^2) per ulp of a near-identity quaternion's w, floored by the slowest raw sample inside the reference's Gaussian
        window), the vector bound applied to every component since the rotation into the base frame mixes them."""
    from tests.test_gpu_parity import slerp_scale_bound, table_speed_floor

    N = orc.N
    S = int(env._c.future_num_steps)
    eps = 6e-8
    origin = orc.env_origins.norm(dim=-1)                                     # [N]
    pos_tol = base + 8.0 * eps * origin                                       # [N]
    tols = {}
    fut = None
    if S:
        times = orc.fut_times                                                 # [N, S]
        ids = orc.slot_clip[:, None].expand(-1, S)
        f0, f1, _ = oml.calc_frame_blend(times, oml.motion_len[ids], oml.num_frames[ids], oml.motion_dt[ids])
        f0, f1 = f0 + oml.length_starts[ids], f1 + oml.length_starts[ids]
        grs, gav = oml.cat["grs_t"], oml.cat["gavs_t"]
        an = orc.anchor
        j_root = slerp_scale_bound(grs[f0][..., 0, :], grs[f1][..., 0, :])[..., 0]               # [N, S]
        j_anc = slerp_scale_bound(grs[f0][..., an, :], grs[f1][..., an, :])[..., 0]
        sp = table_speed_floor(gav)[..., 0, 0]                                                   # [F]: root body
        dtc = oml.motion_dt[ids]
        wn = torch.minimum(sp[f0], sp[f1]).clamp(min=1e-3 / dtc)
        w_tol = 32.0 * eps / (dtc * dtc * wn) * 3.0 ** 0.5        # k = 32 ulps, as for the root angular velocity a reset writes                                        # vector bound of the looked-up omega
        fut = dict(j_root=j_root, j_anc=j_anc, w_tol=w_tol)
    for g in o_obs:
        ref = o_obs[g].float()
        tol = base + base * ref.abs()
        for key, pos, n in _key_slices(cfg, env, g):
            sl = slice(pos, pos + n)
            if key in ("local_key_body_pos", "dif_local_key_body_pos", "local_ref_key_body_pos"):
                tol[:, sl] += (pos_tol - base)[:, None]
            if fut is None:
                continue
            steps = S if key.startswith("future_motion_") else 1                                   # next_step_ref_motion: step 0 only
            per = n // steps
            view = lambda t_: t_[:, :steps]                                                          # [N, steps]
            if key in ("future_motion_roll_pitch",):
                tol[:, sl] += (2.5 * view(fut["j_root"]))[:, :, None].expand(-1, -1, per).reshape(N, n)
            elif key in ("future_motion_base_lin_vel", "future_motion_base_ang_vel", "future_motion_base_yaw_vel"):
                mag = ref[:, sl].view(N, steps, per).norm(dim=-1, keepdim=True)
                add = 2.0 * view(fut["j_root"])[:, :, None] * (mag + 1.0)
                if key != "future_motion_base_lin_vel":
                    add = add + view(fut["w_tol"])[:, :, None]
                tol[:, sl] += add.expand(-1, -1, per).reshape(N, n)
            elif key == "future_motion_local_ref_key_body_pos":
                add = (pos_tol - base)[:, None, None] + 2.0 * view(fut["j_anc"])[:, :, None] * 1.5       # key bodies sit within 1.5 m of the anchor
                tol[:, sl] += add.expand(-1, -1, per).reshape(N, n)
            elif key == "next_step_ref_motion":
                # [height 1 | roll pitch 2 | base lin vel 3 | yaw vel 1 | dof pos D | local key bodies 3K]  (general_tracking.py:554-564)
                jr, ja, wt = fut["j_root"][:, 0], fut["j_anc"][:, 0], fut["w_tol"][:, 0]
                D = orc.D
                t = tol[:, sl]
                t[:, 1:3] += (2.5 * jr)[:, None]
                t[:, 3:6] += (2.0 * jr * (ref[:, pos + 3:pos + 6].norm(dim=-1) + 1.0))[:, None]
                t[:, 6:7] += (2.0 * jr * (ref[:, pos + 6].abs() + 1.0) + wt)[:, None]
                t[:, 7 + D:] += ((pos_tol - base) + 3.0 * ja)[:, None]
        tols[g] = tol
    return tols


def _attribute_obs_errors(path, step, cfg, env, obs, o_obs, orc, tols=None):
    """per observation key: how many elements sit above 1e-4, the largest residual, where it is (env, element) and the size of that env's
    origin — appended to `path` (profiles/round3_tolerance_probe.txt is a copy of one run)"""
    with open(path, "a") as f:
        for g in o_obs:
            a, b = obs[g].detach().float().cpu(), o_obs[g].float()
            err = (a - b).abs()
            f.write(f"step {step} group {g} [{tuple(b.shape)}]: max {float(err.max()):.3e}, share above 1e-4 {float((err > 1e-4).float().mean()):.2e}\n")
            for key, pos, n in _key_slices(cfg, env, g):
                e = err[:, pos:pos + n]
                if e.numel() and float(e.max()) > 3e-6:
                    i = int(e.max(dim=1).values.argmax()); j = int(e[i].argmax())
                    ratio = f"  worst residual / bound {float((e / tols[g][:, pos:pos + n]).max()):.2f}" if tols is not None else ""
                    f.write(f"    {key:36s} dim {n:4d}  max {float(e.max()):.3e} at env {i} el {j} (oracle {float(b[i, pos + j]):+.5f}, |origin| "
                            f"{float(orc.env_origins[i].norm()):.1f} m)  >1e-4: {int((e > 1e-4).sum())}  >1e-5: {int((e > 1e-5).sum())} of {e.numel()}{ratio}\n")


def _multi_clip_vs_oracle(N, T, M, library_seed=3):
    import bench
    from oracle.env_v2 import GeneralTrackingOracle
    from oracle.fk import sim_fk
    from oracle.motion_lib import MotionLib as OML
    from pbhc_amd import motion_lib as ML
    from tests.helpers import clip_from_env_golden, fixture_config, skel_from_golden, synth_replay

    cfgname = "v2_g1_29dof_teacher.yaml"
    g = dict(np.load(os.path.join(GOLDEN, "env_v2_teacher29.npz")))
    clips = bench.synth_library(clip_from_env_golden(g), M, seed=library_seed)
    orig = ML.load_motion_file
    ML.load_motion_file = lambda path: [(f"c{i}", c) for i, c in enumerate(clips)]
    try:
        cfg, env = build_hip_env(cfgname, N, general=True, overrides={"domain_rand.push_robots": False})
    finally:
        ML.load_motion_file = orig
    skel = skel_from_golden("g1_29dof")
    oml = OML(skel, clips)
    ocfg = fixture_config(cfgname, N, {"domain_rand.push_robots": False})
    s = env.simulator
    dr = dict(base_com_bias=s._base_com_bias.cpu(), link_mass_scale=s._link_mass_scale.cpu(), friction_coeffs=s.friction_coeffs.cpu(), base_mass_scale=s._base_mass_scale.cpu())
    orc = GeneralTrackingOracle(ocfg, skel, oml, N, dr)
    orc.env_origins = env.env_origins.cpu()
    orc.ref_init_yaw = env.ref_init_yaw
    orc.slot_clip = env._motion_lib.slot_clip.cpu().clone()
    assert len(set(orc.slot_clip.tolist())) == min(M, N) or M > 64           # every clip in use (small libraries)
    D = 29
    gen = torch.Generator().manual_seed(5)
    mlen = oml.motion_len[orc.slot_clip]
    start = torch.rand(N, generator=gen) * mlen
    start[:48] = mlen[:48] - 0.05                        # motion-end time-outs (several clips)
    ep = torch.randint(0, 50, (N,), generator=gen)
    st = {k: v.clone() for k, v in orc.s.items()}
    st["motion_start_times"] = start; st["episode_length_buf"] = ep; st["last_episode_length_buf"] = ep.clone(); st["motion_len"] = mlen.clone()
    st["kp_scale"] = 0.9 + 0.2 * torch.rand(N, D, generator=gen); st["kd_scale"] = 0.9 + 0.2 * torch.rand(N, D, generator=gen)
    st["rfi_lim_scale"] = 0.5 + torch.rand(N, D, generator=gen); st["rao_scale"] = 0.1 * (torch.rand(N, D, generator=gen) - 0.5)
    st["action_delay_idx"] = torch.randint(0, 3, (N,), generator=gen)
    root, qp, qv, cf = synth_replay(oml, skel, N, T + 1, start, ep, orc.dt, orc.env_origins, 6, orc.feet, ids=orc.slot_clip)
    st["root_states"], st["dof_pos"], st["dof_vel"], st["contact_forces"] = root[0], qp[0], qv[0], cf[0]
    flat = {k: v.numpy() for k, v in st.items()}
    for k in orc.sums:
        flat["sum__" + k] = np.zeros(N, np.float32)
    for k in orc.hist:
        flat["hist__" + k] = (0.1 * torch.randn(orc.hist[k].shape, generator=gen)).numpy()
    for k in orc.sigma:
        flat["sigma__" + k] = orc.sigma[k]
    flat.update(reward_penalty_scale=1.0, average_episode_length=0.0, motion_far_threshold=1.5)
    orc.load_state(flat)
    load_state_into_hip_env(env, flat)
    tg = lambda a: a.contiguous().to(DEV)
    env.simulator.set_replay(tg(root[1:]), tg(qp[1:]), tg(qv[1:]), tg(cf[1:]))
    nreset = 0
    for k in range(T):
        act = 0.5 * torch.randn(N, D, generator=gen)
        u = torch.rand(N, D, generator=gen)
        samp = dict(motion_start_times=torch.rand(N, generator=gen) * mlen, kp_scale=0.9 + 0.2 * torch.rand(N, D, generator=gen),
                    kd_scale=0.9 + 0.2 * torch.rand(N, D, generator=gen), rfi_lim_scale=0.5 + torch.rand(N, D, generator=gen),
                    rao_scale=0.1 * (torch.rand(N, D, generator=gen) - 0.5), action_delay_idx=torch.randint(0, 3, (N,), generator=gen))
        frame = dict(root=root[k + 1], dof_pos=qp[k + 1], dof_vel=qv[k + 1], contact=cf[k + 1])
        body = sim_fk(skel, frame["root"], frame["dof_pos"], frame["dof_vel"])
        o_obs, o_rew, o_reset, o_ex = orc.step(act, frame, body, u_rfi=u, reset_samples=samp)
        env.set_injected_draws(u_rfi=tg(u), start_time=tg(samp["motion_start_times"]), kp=tg(samp["kp_scale"]), kd=tg(samp["kd_scale"]),
                               rfi_lim=tg(samp["rfi_lim_scale"]), rao=tg(samp["rao_scale"]), delay=tg(samp["action_delay_idx"]))
        obs, rew, reset, extras = env.step({"actions": tg(act)})
        torch.cuda.synchronize()
        w = f"step {k}: "
        assert torch.equal(reset.cpu(), o_reset), w + f"reset mismatch {int((reset.cpu() != o_reset).sum())}"
        nreset += int(o_reset.sum())
        close(rew, o_rew, 3e-5, w + "rew", rtol=2e-4)
        tols = _conditioned_obs_tolerances(cfg, env, orc, oml, o_obs)
        if os.environ.get("PBHC_TOLERANCE_PROBE"):             # diagnosis: which keys carry the residual, where, and how large the operands are
            _attribute_obs_errors(os.environ["PBHC_TOLERANCE_PROBE"], k, cfg, env, obs, o_obs, orc, tols)
        for ok in o_obs:       # every element against the bound ITS operands give it (no blanket tolerance, no allowed share of outliers)
            close(obs[ok], o_obs[ok], tols[ok], w + ok)
        close(env.motion_len, orc.s["motion_len"], 1e-6, w + "motion_len")
        for name, view in env.history.items():
            close(view, orc.hist[name], 3e-5, w + "hist " + name)
    assert nreset >= 48


def test_general_tracking_config5_shard_with_noise_and_dr_matches_oracle():
    """BASELINE configs[4] at its per-GPU shard: 4096 envs, G1 29-DoF, a 1024-clip synthetic library (the AMASS / LAFAN sets are not
    shipped), observation noise ON (config/obs/motion_tracking/obs_ppo_teacher.yaml:86-113) and the shipped domain randomisation
    (config/domain_rand/main.yaml) ON — against the oracle, not only through statistics.  The HIP side ingests all 1024 clips and steps all
    4096 envs; the oracle (Python FK per clip) holds the tables of the first 40 clips and steps the ~160 envs whose slot maps to one of
    them, on the same replay tensors, with the reset / torque-noise draws injected on both sides.  Batch statistics (adaptive sigma, the
    curricula) are those of all 4096 envs on the HIP side: the oracle is handed them after every step (their own parity: the full-size
    test above).  Observation noise: the oracle computes the noise-free rows, the kernel's own generator is restated on the host
    (tests/helpers.expected_obs_noise) and added — every element then has to agree to the usual bound."""
    import bench
    from oracle.env_v2 import GeneralTrackingOracle
    from oracle.fk import sim_fk
    from oracle.motion_lib import MotionLib as OML
    from pbhc_amd import _lib
    from pbhc_amd import motion_lib as ML
    from pbhc_amd.envs.env_config import SIGMA_KEYS
    from tests.helpers import clip_from_env_golden, expected_obs_noise, fixture_config, skel_from_golden

    N, T, M, MSUB = 4096, 2, 1024, 40
    cfgname = "v2_g1_29dof_teacher.yaml"
    g = dict(np.load(os.path.join(GOLDEN, "env_v2_teacher29.npz")))
    clips = bench.synth_library(clip_from_env_golden(g), M, seed=11)
    orig = ML.load_motion_file
    ML.load_motion_file = lambda path: [(f"c{i}", c) for i, c in enumerate(clips)]
    try:
        cfg, env = build_hip_env(cfgname, N, general=True, noise_off=False, overrides={"domain_rand.push_robots": False})
    finally:
        ML.load_motion_file = orig
    assert any(float(v) > 0 for v in cfg.obs.noise_scales.values()) and cfg.domain_rand.randomize_pd_gain and cfg.domain_rand.randomize_ctrl_delay
    K = _lib.K
    slot_clip = env._motion_lib.slot_clip.cpu().clone()
    ids = (slot_clip < MSUB).nonzero().flatten()
    n = len(ids)
    assert 64 <= n <= 512 and len(set(slot_clip.tolist())) > 900, (n, len(set(slot_clip.tolist())))
    skel = skel_from_golden("g1_29dof")
    oml = OML(skel, clips[:MSUB])
    ocfg = fixture_config(cfgname, n, {"domain_rand.push_robots": False})          # noise scales zeroed: the oracle's rows are noise-free
    sim = env.simulator
    dr = dict(base_com_bias=sim._base_com_bias.cpu()[ids], link_mass_scale=sim._link_mass_scale.cpu()[ids], friction_coeffs=sim.friction_coeffs.cpu()[ids],
              base_mass_scale=sim._base_mass_scale.cpu()[ids])
    orc = GeneralTrackingOracle(ocfg, skel, oml, n, dr)
    orc.env_origins = env.env_origins.cpu()[ids]
    orc.ref_init_yaw = env.ref_init_yaw
    orc.slot_clip = slot_clip[ids].clone()
    D = 29
    gen = torch.Generator().manual_seed(15)
    mlen = env._motion_lib.get_motion_length(env.motion_ids).cpu()
    start = torch.rand(N, generator=gen) * mlen
    start[ids[:12]] = mlen[ids[:12]] - 0.05                  # motion-end time-outs among the compared envs
    ep = torch.randint(0, 50, (N,), generator=gen)
    full = {k: (v.new_zeros((N,) + tuple(v.shape[1:])) if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n else v) for k, v in orc.s.items()}
    full["motion_start_times"] = start; full["episode_length_buf"] = ep; full["last_episode_length_buf"] = ep.clone(); full["motion_len"] = mlen.clone()
    full["kp_scale"] = 0.9 + 0.2 * torch.rand(N, D, generator=gen); full["kd_scale"] = 0.9 + 0.2 * torch.rand(N, D, generator=gen)
    full["rfi_lim_scale"] = 0.5 + torch.rand(N, D, generator=gen); full["rao_scale"] = 0.1 * (torch.rand(N, D, generator=gen) - 0.5)
    full["action_delay_idx"] = torch.randint(0, 3, (N,), generator=gen)
    # the replay window of ALL envs from the HIP lookup (reference state + noise), state 0 = the initial simulator state
    env._episode_length_buf.copy_(ep.to(DEV)); env.motion_start_times.copy_(start.to(DEV)); env.motion_len.copy_(mlen.to(DEV))
    root, qp, qv, cf = [t.cpu() for t in bench.make_replay_on_device(env, T + 1, seed=21)]
    full["root_states"], full["dof_pos"], full["dof_vel"], full["contact_forces"] = root[0], qp[0], qv[0], cf[0]
    flat = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in full.items()}
    for k in orc.sums:
        flat["sum__" + k] = np.zeros(N, np.float32)
    for k in orc.hist:
        flat["hist__" + k] = (0.1 * torch.randn((N,) + tuple(orc.hist[k].shape[1:]), generator=gen)).numpy()
    for k in orc.sigma:
        flat["sigma__" + k] = orc.sigma[k]
    flat.update(reward_penalty_scale=1.0, average_episode_length=0.0, motion_far_threshold=1.5)
    sub = lambda v: v[ids.numpy()] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == N else v
    orc.load_state({k: sub(v) for k, v in flat.items()})
    load_state_into_hip_env(env, flat)
    tg = lambda a: a.contiguous().to(DEV)
    env.simulator.set_replay(tg(root[1:]), tg(qp[1:]), tg(qv[1:]), tg(cf[1:]))
    L = env.layout
    nreset = 0
    for k in range(T):
        act = 0.5 * torch.randn(N, D, generator=gen)
        u = torch.rand(N, D, generator=gen)
        samp = dict(motion_start_times=torch.rand(N, generator=gen) * mlen, kp_scale=0.9 + 0.2 * torch.rand(N, D, generator=gen),
                    kd_scale=0.9 + 0.2 * torch.rand(N, D, generator=gen), rfi_lim_scale=0.5 + torch.rand(N, D, generator=gen),
                    rao_scale=0.1 * (torch.rand(N, D, generator=gen) - 0.5), action_delay_idx=torch.randint(0, 3, (N,), generator=gen))
        frame = dict(root=root[k + 1][ids], dof_pos=qp[k + 1][ids], dof_vel=qv[k + 1][ids], contact=cf[k + 1][ids])
        body = sim_fk(skel, frame["root"], frame["dof_pos"], frame["dof_vel"])
        o_obs, o_rew, o_reset, o_ex = orc.step(act[ids], frame, body, u_rfi=u[ids], reset_samples={a: b[ids] for a, b in samp.items()})
        env.set_injected_draws(u_rfi=tg(u), start_time=tg(samp["motion_start_times"]), kp=tg(samp["kp_scale"]), kd=tg(samp["kd_scale"]),
                               rfi_lim=tg(samp["rfi_lim_scale"]), rao=tg(samp["rao_scale"]), delay=tg(samp["action_delay_idx"]))
        env.wait_finalize()
        step_ctr = int(env.globals[K["PBHC_G_STEP_COUNTER"]])
        noise_cur = float(env.globals[K["PBHC_G_NOISE_CURRICULUM"]])
        obs, rew, reset, extras = env.step({"actions": tg(act)})
        torch.cuda.synchronize()
        w = f"step {k}: "
        assert torch.equal(reset.cpu()[ids], o_reset), w + f"reset mismatch {int((reset.cpu()[ids] != o_reset).sum())}"
        nreset += int(o_reset.sum())
        close(rew[ids.to(DEV)], o_rew, 3e-5, w + "rew", rtol=2e-4)
        tols = _conditioned_obs_tolerances(cfg, env, orc, oml, o_obs)
        seen_noise = 0.0
        def by_dst(gi):                                            # (noise, scale) of a row indexed by the element's position in the row
            td, _, tsc, tn = L.map_tensors[gi]
            dim = int(td.max()) + 1
            ns, sc = np.zeros(dim, np.float32), np.ones(dim, np.float32)
            ns[td.cpu().numpy()], sc[td.cpu().numpy()] = tn.cpu().numpy(), tsc.cpu().numpy()
            return ns, sc

        clipv = float(env._c.clip_observations)
        for gi, gname in enumerate(L.group_names[:-1]):
            add = expected_obs_noise(env._seed, ids.numpy(), step_ctr, gi, *by_dst(gi), noise_cur)
            want = (o_obs[gname] + add).clamp(-clipv, clipv)
            seen_noise = max(seen_noise, float(add.abs().max()))
            close(obs[gname][ids.to(DEV)], want, tols[gname] + 2e-6, w + gname + " (noise on)")
        assert seen_noise > 1e-3                                 # noise was on, and of visible size
        # the history: every key's newest entry carries ITS OWN draw (the write-back is a group of its own, helpers.py:128-152 is called once
        # per group); checked against the oracle's noise-free history + the restated noise, then handed to the oracle for the next step
        gi = len(L.group_names) - 1
        addh = expected_obs_noise(env._seed, ids.numpy(), step_ctr, gi, *by_dst(gi), noise_cur)
        for name, view in env.history.items():
            o0 = L.hist_off[name]
            hv = view.cpu()[ids]
            exp = orc.hist[name].clone()
            exp[:, 0] = exp[:, 0] + addh[:, o0:o0 + exp.shape[-1]]   # (a reset env too: its history was zeroed BEFORE this step's entry was added)
            close(hv, exp, 3e-5, w + "hist " + name, rtol=3e-5)
            orc.hist[name] = hv.clone()
        # batch statistics of ALL envs -> the oracle (see the docstring)
        gl = env.globals.cpu().numpy()
        for i, name in enumerate(SIGMA_KEYS):
            if name in orc.sigma:
                orc.sigma[name], orc.ema[name] = float(gl[K["PBHC_G_SIGMA"] + i]), float(gl[K["PBHC_G_EMA"] + i])
        orc.penalty_scale = float(gl[K["PBHC_G_PENALTY_SCALE"]])
        orc.avg_ep_len = torch.tensor(float(gl[K["PBHC_G_AVG_EP_LEN"]]), dtype=torch.float32)
        orc.motion_far_thr = float(gl[K["PBHC_G_MOTION_FAR_THR"]])
    assert nreset >= 12


def test_ppo_mimic_distillation_matches_reference(tmp_path):
    """Student distillation (teacher_model_path + dagger_only): the teacher observation groups are added to the existing env, the teacher
    actor is loaded from a checkpoint + config.yaml, and one _training_step_distill on the reference's rollout buffer / permutation
    reproduces the reference's student weights and bc loss (tests/golden/ppo_distill.npz)."""
    from pbhc_amd.agents.ppo_mimic import PPO
    from pbhc_amd.utils.config import load_unresolved, save_unresolved, set_by_path
    from tests.helpers import PPO_V2_NARROW

    raw = np.load(os.path.join(GOLDEN, "ppo_distill.npz"))
    g = {k: torch.from_numpy(raw[k]) for k in raw.files if raw[k].dtype.kind in "fiub"}
    # the teacher's checkpoint + composed config next to it, as the reference expects them (ppo_mimic.py:127-128,165)
    tdir = tmp_path / "teacher"
    tdir.mkdir()
    tc = load_unresolved(os.path.join(GOLDEN, "configs", "v2_g1_23dof_teacher.yaml"))
    for k, v in PPO_V2_NARROW.items():
        set_by_path(tc, k, v)
    for k in ("motion_file",):
        tc["robot"]["motion"][k] = os.path.join(os.path.dirname(GOLDEN), "..", tc["robot"]["motion"][k])
    save_unresolved(tc, str(tdir / "config.yaml"))
    torch.save({"model_state_dict": {k[len("teacher__"):]: v for k, v in g.items() if k.startswith("teacher__")}, "iter": 0, "infos": None}, str(tdir / "model_0.pt"))
    N = g["st__actions"].shape[1]
    ov = dict(PPO_V2_NARROW)
    ov.update({"domain_rand.push_robots": False, "algo.config.teacher_model_path": str(tdir / "model_0.pt"), "algo.config.dagger_only": True})
    cfg, env = build_hip_env("v2_g1_23dof_student.yaml", N, general=True, overrides=ov)
    algo = PPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
    algo.setup()
    assert list(env.obs_buf_dict.keys()) == [str(k) for k in raw["obs_keys"]]
    assert [f"{k}={v}" for k, v in algo.algo_obs_dim_dict.items()] == [str(x) for x in raw["algo_obs_dims"]]
    algo.alg.load_state_dict({k[len("w0__"):]: v for k, v in g.items() if k.startswith("w0__")}, strict=True)
    for k in algo.storage.stored_keys:
        getattr(algo.storage, k).copy_(g["st__" + k].to(DEV))
    algo._train_mode()
    with torch.no_grad():
        b = {k: getattr(algo.storage, k).flatten(0, 1) for k in algo._obs_width}
        close(algo.teacher_actor_act_step(b), g["st__teacher_actions"].flatten(0, 1), 2e-5, "teacher actions")
        close(algo.alg.act_inference(b, hist_encoding=True), g["st__actions"].flatten(0, 1), 2e-5, "student mean")
    loss = algo._training_step_distill(indices=g["perm"].to(DEV))
    torch.cuda.synchronize()
    assert abs(float(loss["bc_loss"]) - float(g["loss__bc_loss"])) < 2e-4
    for k, v in algo.alg.state_dict().items():
        ref = g["w1__" + k]
        close(v, ref, 2e-4, "w1 " + k, rtol=2e-4)
    # the env still steps with the two extra groups, and a short distillation run stays finite
    algo.learn(num_iterations=1)
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in algo.alg.parameters())
    assert env.obs_buf_dict["teacher_future_motion_targets"].shape == (N, 57 * 20)
    d = algo.optimizer.state_dict()
    assert len(d["param_groups"][0]["params"]) == len(list(algo.alg.actor.parameters()))


def test_periodic_motion_resampling_resets_every_env():
    """resample_motion_when_training (general_tracking.py:216-222,291-297): every resample_time_interval steps the slot -> clip map is
    redrawn from the sampling probabilities and every env is reset onto its new clip."""
    import bench
    from pbhc_amd import motion_lib as ML
    from tests.helpers import clip_from_env_golden

    g = dict(np.load(os.path.join(GOLDEN, "env_v2_teacher29.npz")))
    clips = bench.synth_library(clip_from_env_golden(g), 3, seed=3)
    orig = ML.load_motion_file
    ML.load_motion_file = lambda path: [(f"c{i}", c) for i, c in enumerate(clips)]
    try:
        cfg, env = build_hip_env("v2_g1_29dof_teacher.yaml", 64, general=True,
                                 overrides={"domain_rand.push_robots": False, "env.config.resample_time_interval_s": 0.02 * 3})
    finally:
        ML.load_motion_file = orig
    assert env.resample_time_interval == 3
    env.reset_all()                                         # one step
    torch.manual_seed(0)
    act = torch.zeros(64, env.num_dof, device=DEV)
    env.step({"actions": act})                              # step 2
    before = env._motion_lib.slot_clip.clone()
    assert before.tolist() == [i % 3 for i in range(64)]
    env._motion_lib._sampling_prob.copy_(torch.tensor([0.0, 0.0, 1.0], device=DEV))       # the sampling hook: only clip 2 from now on
    env.step({"actions": act})                              # step 3 -> resample
    torch.cuda.synchronize()
    assert env._motion_lib.slot_clip.tolist() == [2] * 64
    assert int(env.episode_length_buf.abs().sum()) == 0
    assert torch.allclose(env.motion_len, env._motion_lib._motion_lengths[2].expand(64))
    assert bool((env.motion_start_times <= env.motion_len).all())
    obs, _, reset, _ = env.step({"actions": act})
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in obs.values())


def test_hip_env_reproduces_recorded_deploy_observations():
    """The reference's own recorded sim2sim rollout (tests/test_deploy_recording.py explains fixture, state recovery and tolerances): the
    fused HIP step, fed the recorded actions and recovered robot states through the replay stub, reproduces every recorded 877-wide
    `actor_obs` row — layout, scales, Euler convention, anchor frame, key-major history and the reference-motion block."""
    from tests.helpers import deploy_layout_slices, deploy_recording_states, fresh_episode_state
    from tests.test_deploy_recording import check_rows
    from tests.test_oracle_env_v2 import build_oracle_v2

    rec = dict(np.load(os.path.join(GOLDEN, "deploy_student23_recording.npz")))
    g, orc, skel = build_oracle_v2("student23", "v2_g1_23dof_student.yaml", "g1_23dof")
    N = orc.N
    cfg, env = build_hip_env("v2_g1_23dof_student.yaml", N, general=True, overrides={"domain_rand.push_robots": False})
    env.env_origins.zero_()
    load_state_into_hip_env(env, fresh_episode_state(orc))
    root, qp, qv = deploy_recording_states(rec, orc.ml, orc.default_dof_pos.reshape(-1)[: orc.D], orc.cfg.obs.obs_scales, orc.dt)
    T = root.shape[0]
    rep = lambda a: a[:, None].repeat(1, N, *([1] * (a.dim() - 1))).contiguous().to(DEV)
    env.simulator.set_replay(rep(root), rep(qp), rep(qv), torch.zeros(T, N, env.simulator.num_bodies, 3, device=DEV))
    acts = torch.from_numpy(rec["action"]).to(DEV)
    worst = {}
    for r in range(1, T + 1):
        obs, rew, reset, extras = env.step({"actions": acts[r - 1][None].repeat(N, 1).contiguous()})
        a = obs["actor_obs"].cpu()
        assert not reset.any(), f"row {r}: unexpected termination"
        assert a.shape == (N, 877) and torch.equal(a[0], a[N - 1])
        check_rows(a[0].numpy(), rec["actor_obs"][r], r, worst)
        assert torch.equal(obs["prop_history"].cpu()[0], a[0][deploy_layout_slices()["history_actor"]])
    assert worst["actions"] == 0.0 and worst["roll_pitch"] > 0


def test_hip_env_reproduces_sim2sim_log():
    """Second recording of the reference's deploy stack (exact observation-time robot state logged, 199 consecutive steps on g1_walk_45cms,
    futures running past the clip end): actor_obs (877) and future_motion_targets (600) of the fused HIP step against the logged rows."""
    from tests.helpers import fresh_episode_state
    from tests.test_deploy_recording import LAST_ROW, WALK_CLIP, check_log_row, sim2sim_log_frames, student_walk_oracle

    log = dict(np.load(os.path.join(GOLDEN, "deploy_sim2sim_log_walk.npz")))
    N = 8
    orc, skel = student_walk_oracle(N)
    cfg, env = build_hip_env("v2_g1_23dof_student.yaml", N, general=True, overrides={"domain_rand.push_robots": False, "robot.motion.motion_file": WALK_CLIP})
    env.env_origins.zero_()
    load_state_into_hip_env(env, fresh_episode_state(orc))
    root, qp, qv = sim2sim_log_frames(log, orc.ml)
    rep = lambda a: a[:, None].repeat(1, N, 1).contiguous().to(DEV)
    env.simulator.set_replay(rep(root), rep(qp), rep(qv), torch.zeros(LAST_ROW, N, env.simulator.num_bodies, 3, device=DEV))
    acts = torch.from_numpy(log["action"]).to(DEV)
    worst = {}
    for r in range(1, LAST_ROW + 1):
        obs, rew, reset, extras = env.step({"actions": acts[r - 1][None].repeat(N, 1).contiguous()})
        assert not reset.any(), f"row {r}: unexpected termination"
        o = {k: obs[k].cpu() for k in ("actor_obs", "future_motion_targets")}
        assert torch.equal(o["actor_obs"][0], o["actor_obs"][N - 1])
        check_log_row(o, log, r, worst)
    assert worst["anchor_ref_rot"] > 0 and worst["actions"] == 0.0


def test_general_tracking_observation_noise_at_shard_size():
    """BASELINE configs[4]'s per-GPU shard: 4096 envs, 29-DoF teacher, a mixed clip library, observation noise ON and the shipped domain
    randomisation ON (`obs_ppo_teacher.yaml:86-113`, `domain_rand/main.yaml`).  Noise-on vs noise-off step from the same state and seeds
    (reference: helpers.parse_observation, (x + (2U-1) * noise) * scale, helpers.py:128-152): per key and group the differences are bounded by
    noise * scale, have mean ~ 0 and the variance of a uniform on [-1, 1]; noise-free keys are identical."""
    import bench
    from pbhc_amd import motion_lib as ML
    from tests.helpers import clip_from_env_golden

    N, M = 4096, 64
    g = dict(np.load(os.path.join(GOLDEN, "env_v2_teacher29.npz")))
    clips = bench.synth_library(clip_from_env_golden(g), M, seed=11)
    outs = []
    for noise_off in (True, False):
        orig = ML.load_motion_file
        ML.load_motion_file = lambda path: [(f"c{i}", c) for i, c in enumerate(clips)]
        try:
            torch.manual_seed(21)
            cfg, env = build_hip_env("v2_g1_29dof_teacher.yaml", N, general=True, noise_off=noise_off)
        finally:
            ML.load_motion_file = orig
        torch.manual_seed(22)
        env.reset_all()
        obs, rew, reset, _ = env.step({"actions": torch.zeros(N, env.num_dof, device=DEV)})
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(v).all()) for v in obs.values()) and bool(torch.isfinite(rew).all())
        outs.append(({k: v.clone() for k, v in obs.items()}, cfg, env.layout, reset.clone()))
    (clean, _, _, r0), (noisy, cfg, L, r1) = outs
    assert torch.equal(r0, r1)                                   # the noise does not feed termination
    ob = cfg.obs
    checked = 0
    def width(key):
        return L.obs_dims[key] if key in L.obs_dims else sum(L.obs_dims[k] * n for k, n in ob.obs_auxiliary[key].items())

    for grp, keys in ob.obs_dict.items():
        pos = 0
        if noisy[grp].shape[1] != sum(width(k) for k in keys):   # the future-target group: one row of its keys per future step (general_tracking.py:501-560)
            assert noisy[grp].shape[1] % sum(width(k) for k in keys) == 0 and all(float(ob.noise_scales[k]) == 0.0 for k in keys), grp
            assert torch.equal(noisy[grp], clean[grp]), grp
            continue
        for key in sorted(keys):
            d = width(key)
            if key not in ob.obs_auxiliary:
                diff = (noisy[grp][:, pos:pos + d] - clean[grp][:, pos:pos + d]).cpu()
                amp = float(ob.noise_scales[key]) * float(ob.obs_scales[key])
                if amp == 0.0:
                    assert float(diff.abs().max()) == 0.0, (grp, key)
                else:
                    assert float(diff.abs().max()) <= amp * (1 + 1e-5) + 1e-6, (grp, key, float(diff.abs().max()), amp)
                    assert abs(float(diff.mean())) < 0.05 * amp and abs(float(diff.var()) / (amp * amp / 3.0) - 1.0) < 0.1, (grp, key, float(diff.mean()), float(diff.var()), amp)
                    checked += 1
            pos += d
        assert pos == noisy[grp].shape[1], grp
    assert checked >= 4
