"""Data-parallel equivalence on the GPU box: 2 ranks x 4 envs (gloo, both on cuda:0) take the same MHPPO update as 1 process x 8 envs.
The reference is single-process; sharding must not change the maths: global advantage moments, rank-averaged gradients, one KL decision."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.helpers import GOLDEN, build_hip_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make_algo(g, N):
    from pbhc_amd.agents.mh_ppo import MHPPO

    cfg, env = build_hip_env("v1_g1_23dof_horse_stance.yaml", N)
    hd = [int(x) for x in g["hidden_dims"]]
    cfg.algo.config.module_dict.actor.layer_config.hidden_dims = hd
    cfg.algo.config.module_dict.critic.layer_config.hidden_dims = hd
    algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
    algo.setup()
    algo.actor.load_state_dict({k[len("actor__"):]: v for k, v in g.items() if k.startswith("actor__")}, strict=True)
    algo.critic.load_state_dict({k[len("critic__"):]: v for k, v in g.items() if k.startswith("critic__")}, strict=True)
    algo._train_mode()
    return algo


def _load_storage(algo, g, env_slice):
    for k in algo.storage.stored_keys:
        if k in ("returns", "advantages"):
            continue
        getattr(algo.storage, k).copy_(g["st__" + k][:, env_slice].to(DEV))


def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, perms, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v1.npz")).items()}
        n = g["st__actions"].shape[1] // world
        algo = _make_algo(g, n)
        assert algo.world_size == world
        sl = slice(rank * n, (rank + 1) * n)
        _load_storage(algo, g, sl)
        last = {"critic_obs": g["last_critic_obs"][sl].to(DEV)}
        with torch.no_grad():
            algo._compute_returns(last)
        algo._training_step(indices=perms[rank].to(DEV))
        torch.cuda.synchronize()
        if rank == 0:
            sd = {"actor." + k: v.cpu() for k, v in algo.actor.state_dict().items()}
            sd.update({"critic." + k: v.cpu() for k, v in algo.critic.state_dict().items()})
            sd["adv"] = algo.storage.advantages.cpu()
            sd["lr"] = algo._lr.cpu()
            torch.save(sd, out_path)
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_big_batch(tmp_path):
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v1.npz")).items()}
    T, N = g["st__actions"].shape[:2]
    world, n = 2, N // 2
    gen = torch.Generator().manual_seed(0)
    perms = [torch.randperm(T * n, generator=gen) for _ in range(world)]
    # the single-process permutation whose minibatch i is the union of the ranks' minibatches i (sample (t, e) of rank r is (t, r*n + e))
    mbr = (T * n) // 4
    to_global = lambda r, p: (p // n) * N + r * n + (p % n)
    big = torch.cat([torch.cat([to_global(r, perms[r][i * mbr:(i + 1) * mbr]) for r in range(world)]) for i in range(4)])
    assert sorted(big.tolist()) == list(range(T * N))
    # ---- 1 process x 8 envs
    algo = _make_algo(g, N)
    _load_storage(algo, g, slice(0, N))
    with torch.no_grad():
        algo._compute_returns({"critic_obs": g["last_critic_obs"].to(DEV)})
    adv1 = algo.storage.advantages.cpu().clone()
    algo._training_step(indices=big.to(DEV))
    torch.cuda.synchronize()
    ref = {"actor." + k: v.cpu() for k, v in algo.actor.state_dict().items()}
    ref.update({"critic." + k: v.cpu() for k, v in algo.critic.state_dict().items()})
    lr1 = algo._lr.cpu().clone()
    del algo
    # ---- 2 processes x 4 envs (both on this GPU, gloo)
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_rank_main, args=(world, _free_port(), perms, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    assert torch.allclose(got["adv"], adv1[:, :n], atol=2e-5, rtol=1e-5)            # globally normalised advantages
    assert torch.allclose(got["lr"], lr1, rtol=1e-6)                                # same KL decisions
    for k, v in ref.items():
        assert torch.allclose(got[k], v, atol=2e-4, rtol=2e-4), k                   # Adam-amplified rounding, see test_mhppo_update_matches_reference
        assert float((got[k] - v).norm() / v.norm().clamp(min=1e-6)) < 1e-4, k


# ---- env batch statistics over ranks -----------------------------------------------------------------------------------------
def _slice_envs(d, sl, N):
    return {k: (v[sl] if getattr(v, "ndim", 0) >= 1 and v.shape[0] == N else v) for k, v in d.items()}


def _env_rank_main(rank, world, port, out_path):
    from pbhc_amd import _lib
    from pbhc_amd.envs.env_config import SIGMA_KEYS
    from tests.helpers import load_env_golden, load_state_into_hip_env, state_dict_from_golden

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        g = load_env_golden("horse")
        T, N, D = g["actions_in"].shape
        n = N // world
        sl = slice(rank * n, (rank + 1) * n)
        cfg, env = build_hip_env("v1_g1_23dof_horse_stance.yaml", n)
        gs = {k: g[k][sl] for k in ("env_origins", "base_com_bias", "link_mass_scale", "friction_coeffs", "base_mass_scale") if k in g}
        load_state_into_hip_env(env, _slice_envs(state_dict_from_golden(g), sl, N), gs)
        assert env.enable_global_statistics(mode="step") and env._num_envs_total == N       # exact single-process equivalence, step by step
        tg = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        env.simulator.set_replay(*[tg(g[k][:, sl]) for k in ("replay_root", "replay_dof_pos", "replay_dof_vel", "replay_contact")])
        K = _lib.K
        rows = []
        for k in range(T):
            st = lambda name, dt=torch.float32: tg(g["step__state__" + name][k][sl]).to(dt)
            env.set_injected_draws(u_rfi=tg(g["step__u_rfi"][k][sl]), start_time=st("motion_start_times"), kp=st("kp_scale"), kd=st("kd_scale"),
                                   rfi_lim=st("rfi_lim_scale"), rao=st("rao_scale"), delay=st("action_delay_idx", torch.long))
            obs, rew, reset, extras = env.step({"actions": tg(g["actions_in"][k][sl])})
            log = env.read_log()                       # flushes the pending all-reduce + finalize
            gl = env.globals.cpu().numpy()
            rows.append(dict(rew=rew.cpu().clone(), reset=reset.cpu().clone(), sigma=gl[K["PBHC_G_SIGMA"]:K["PBHC_G_SIGMA"] + len(SIGMA_KEYS)].copy(),
                             penalty=gl[K["PBHC_G_PENALTY_SCALE"]], avg=gl[K["PBHC_G_AVG_EP_LEN"]], far=gl[K["PBHC_G_MOTION_FAR_THR"]],
                             upper=log["upper_body_diff_norm"], grav=log["terminate_by_gravity"]))
        torch.save(rows, out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_env_statistics_over_two_ranks_follow_the_single_process_trace(tmp_path):
    """2 ranks x 16 envs replay the two halves of the reference's 32-env trace (tests/golden/env_v1_horse.npz) with global batch statistics
    on: adaptive sigma, penalty / motion-far curricula, average episode length and the logged means on BOTH ranks follow the reference's
    single-process values step by step, and so do the rewards of each half (they depend on the sigma of the previous step)."""
    from pbhc_amd.envs.env_config import SIGMA_KEYS
    from tests.helpers import load_env_golden
    from tests.test_gpu_parity import close

    g = load_env_golden("horse")
    T, N, D = g["actions_in"].shape
    world = 2
    out = str(tmp_path / "rows.pt")
    mp.spawn(_env_rank_main, args=(world, _free_port(), out), nprocs=world, join=True)
    n = N // world
    moved = 0
    for rank in range(world):
        rows = torch.load(out + f".{rank}", weights_only=False)
        sl = slice(rank * n, (rank + 1) * n)
        for k, r in enumerate(rows):
            w = f"rank {rank} step {k}: "
            assert torch.equal(r["reset"], torch.from_numpy(g["step__reset_buf_out"][k][sl])), w + "reset"
            close(r["rew"], g["step__rew_buf"][k][sl], 3e-5, w + "rew_buf", rtol=1e-4)
            for i, name in enumerate(SIGMA_KEYS):
                if "step__state__sigma__" + name in g:
                    ref = float(g["step__state__sigma__" + name][k])
                    assert abs(r["sigma"][i] - ref) <= 2e-6 * abs(ref), w + "sigma " + name
                    moved += int(k > 0 and ref != float(g["step__state__sigma__" + name][k - 1]))
            assert abs(r["penalty"] - float(g["step__state__reward_penalty_scale"][k])) < 1e-9, w + "penalty scale"
            assert abs(r["avg"] - float(g["step__state__average_episode_length"][k])) < 1e-5, w + "average episode length"
            assert abs(r["far"] - float(g["step__state__motion_far_threshold"][k])) < 1e-9, w + "motion far threshold"
            close(torch.tensor(r["upper"]), g["step__log__upper_body_diff_norm"][k], 1e-4, w + "log upper_body_diff_norm")
            close(torch.tensor(r["grav"]), g["step__log__terminate_by_gravity"][k], 1e-4, w + "log terminate_by_gravity")
    assert moved > 0            # sigma did change during the trace


def _rollout_mode_rank_main(rank, world, port, out_path):
    from tests.helpers import load_env_golden, load_state_into_hip_env, state_dict_from_golden

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        g = load_env_golden("horse")
        T, N, D = g["actions_in"].shape
        n = N // world
        sl = slice(rank * n, (rank + 1) * n)
        cfg, env = build_hip_env("v1_g1_23dof_horse_stance.yaml", n)
        gs = {k: g[k][sl] for k in ("env_origins", "base_com_bias", "link_mass_scale", "friction_coeffs", "base_mass_scale") if k in g}
        load_state_into_hip_env(env, _slice_envs(state_dict_from_golden(g), sl, N), gs)
        assert env.enable_global_statistics() and env._totals is None                          # default mode: "rollout"
        tg = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        env.simulator.set_replay(*[tg(g[k][:, sl]) for k in ("replay_root", "replay_dof_pos", "replay_dof_vel", "replay_contact")])
        from pbhc_amd import dist as pdist

        pdist.reset_counters()
        for k in range(T):
            st = lambda name, dt=torch.float32: tg(g["step__state__" + name][k][sl]).to(dt)
            env.set_injected_draws(u_rfi=tg(g["step__u_rfi"][k][sl]), start_time=st("motion_start_times"), kp=st("kp_scale"), kd=st("kd_scale"),
                                   rfi_lim=st("rfi_lim_scale"), rao=st("rao_scale"), delay=st("action_delay_idx", torch.long))
            env.step({"actions": tg(g["actions_in"][k][sl])})
        assert pdist.COUNTERS["all_reduce"] == 0                                                # nothing exchanged inside the rollout
        env.wait_finalize()
        before = env.globals.cpu().clone()
        env.sync_globals()
        torch.cuda.synchronize()
        assert pdist.COUNTERS["all_reduce"] == 1
        torch.save(dict(before=before, after=env.globals.cpu().clone()), out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_rollout_mode_averages_the_statistics_once_per_rollout(tmp_path):
    """sync_env_statistics "rollout" (the default): inside a rollout every rank updates sigma / the curricula / the log means from its OWN
    shard and no collective runs; `sync_globals()` — one 1 KB all-reduce per PPO iteration — then leaves the mean over the ranks on every
    rank.  The two shards of the reference's 32-env trace diverge (different envs), so the mean differs from both."""
    world = 2
    out = str(tmp_path / "g.pt")
    mp.spawn(_rollout_mode_rank_main, args=(world, _free_port(), out), nprocs=world, join=True)
    r = [torch.load(out + f".{k}", weights_only=False) for k in range(world)]
    mean = 0.5 * (r[0]["before"] + r[1]["before"])
    assert float((r[0]["before"] - r[1]["before"]).abs().max()) > 1e-6
    for k in range(world):
        assert torch.allclose(r[k]["after"], mean, rtol=1e-12, atol=0.0), k


# ---- equal seeds on every rank must not replicate the random streams ---------------------------------------------------------
def _seed_rank_main(rank, world, port, out_path):
    from pbhc_amd.agents.mh_ppo import MHPPO

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(11)                              # the same config.seed on both ranks
        cfg, env = build_hip_env("v1_g1_23dof_walk.yaml", 64)
        algo = MHPPO(env=env, config=cfg.algo.config, log_dir=None, device=DEV)
        algo.setup()
        obs = env.reset_all()
        algo._train_mode()
        algo._rollout_step(obs)
        torch.cuda.synchronize()
        torch.save(dict(seed=env._seed, sample_seed=algo._sample_seed, actions=algo.storage.actions.cpu().clone(), start=env.motion_start_times.cpu().clone(),
                        kp=env._kp_scale.cpu().clone(), w0=algo.actor.actor_module.module[0].weight.detach().cpu().clone()), out_path + f".{rank}")
    finally:
        dist.destroy_process_group()


def test_equal_seeds_give_different_streams_but_equal_weights(tmp_path):
    out = str(tmp_path / "seed.pt")
    mp.spawn(_seed_rank_main, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + ".0", weights_only=True), torch.load(out + ".1", weights_only=True)
    assert a["seed"] != b["seed"] and a["sample_seed"] != b["sample_seed"]
    assert torch.equal(a["w0"], b["w0"])                                          # replicas start from rank 0's weights
    assert not torch.allclose(a["actions"], b["actions"])                         # action noise (in-kernel Philox, sample seed)
    assert not torch.allclose(a["start"], b["start"])                             # start phases (reset_all: host generator; resets: env Philox)
    assert not torch.allclose(a["kp"], b["kp"])                                   # episodic domain randomisation


# ---- ppo_mimic (general tracking, BASELINE configs[4]) under data parallelism ------------------------------------------------------
def _v2_make_algo(g, N):
    from tests.helpers import PPO_V2_NARROW
    from tests.test_gpu_parity_v2 import _v2_algo

    cfg, env, algo = _v2_algo(N, PPO_V2_NARROW)
    algo.alg.load_state_dict({k[len("w0__"):]: v for k, v in g.items() if k.startswith("w0__")}, strict=True)
    algo._train_mode()
    algo.counter = int(g["counter0"])
    return algo


def _v2_load_storage(algo, g, sl):
    for k in algo.storage.stored_keys:
        if k in ("returns", "advantages"):
            continue
        getattr(algo.storage, k).copy_(g["st__" + k][:, sl].to(DEV))


def _v2_state(algo):
    sd = {"w." + k: v.detach().cpu().clone() for k, v in algo.alg.state_dict().items()}
    sd["lr"] = algo._lr.cpu().clone()
    return sd


def _v2_run(algo, g, sl, perm_ppo, perm_dagger):
    """returns -> one PPO `_training_step` (priv_reg coefficient from the golden's counter) -> one DAgger step; snapshots after each"""
    last = {k[len("last__"):]: v[sl].to(DEV) for k, v in g.items() if k.startswith("last__")}
    with torch.no_grad():
        algo._compute_returns(last)
    adv = algo.storage.advantages.cpu().clone()
    loss1 = algo._training_step(indices=perm_ppo.to(DEV))
    torch.cuda.synchronize()
    s1 = _v2_state(algo)
    _v2_load_storage(algo, g, sl)                                   # (the DAgger step reads observations only; the buffer was cleared, not erased)
    loss2 = algo._training_step_dagger(indices=perm_dagger.to(DEV))
    torch.cuda.synchronize()
    s2 = _v2_state(algo)
    return dict(adv=adv, s1=s1, s2=s2, priv_reg=float(loss1["priv_reg_loss"]), hist_loss=float(loss2["hist_latent_loss"]), counter=algo.counter)


def _v2_rank_main(rank, world, port, perms, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        from pbhc_amd import dist as pdist

        g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v2.npz")).items()}
        n = g["st__actions"].shape[1] // world
        algo = _v2_make_algo(g, n)
        assert algo.world_size == world and algo._dp
        sl = slice(rank * n, (rank + 1) * n)
        _v2_load_storage(algo, g, sl)
        pdist.reset_counters()
        out = _v2_run(algo, g, sl, perms[0][rank], perms[1][rank])
        out["all_reduces"] = int(pdist.COUNTERS["all_reduce"])
        out["steps"] = algo.num_learning_epochs * algo.num_mini_batches
        if rank == 0:
            torch.save(out, out_path)
    finally:
        dist.destroy_process_group()


def test_ppo_mimic_two_ranks_equal_one_big_batch(tmp_path):
    """`ppo_mimic.PPO` sharded over 2 ranks (gloo, both on this GPU) against 1 process with all the envs, on the reference's rollout buffer
    (tests/golden/ppo_v2.npz, `priv_reg` coefficient 0.05 at its counter): globally normalised advantages (ppo_mimic.py:489-491), the PPO step
    (:596-691: rank-averaged gradients of the main segment, ONE learning-rate decision from the all-rank KL carried in the gradient bucket),
    the DAgger regression of the history encoder (:693-709) — weights after each, learning rate, loss meters; and the collective count:
    one all-reduce per optimiser step + the advantage moments."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v2.npz")).items()}
    T, N = g["st__actions"].shape[:2]
    world, n = 2, N // 2
    gen = torch.Generator().manual_seed(3)
    perms = [[torch.randperm(T * n, generator=gen) for _ in range(world)] for _ in range(2)]          # [ppo | dagger][rank]
    algo = _v2_make_algo(g, N)
    nmb = algo.num_mini_batches
    mbr = (T * n) // nmb
    to_global = lambda r, p: (p // n) * N + r * n + (p % n)
    big = [torch.cat([torch.cat([to_global(r, perms[j][r][i * mbr:(i + 1) * mbr]) for r in range(world)]) for i in range(nmb)]) for j in range(2)]
    assert sorted(big[0].tolist()) == list(range(T * N))
    # ---- 1 process x N envs
    _v2_load_storage(algo, g, slice(0, N))
    ref = _v2_run(algo, g, slice(0, N), big[0], big[1])
    del algo
    # ---- 2 processes x N/2 envs
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_v2_rank_main, args=(world, _free_port(), perms, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=False)
    assert torch.allclose(got["adv"], ref["adv"][:, :n], atol=2e-5, rtol=1e-5)
    assert got["counter"] == ref["counter"]
    for stage in ("s1", "s2"):
        assert torch.allclose(got[stage]["lr"], ref[stage]["lr"], rtol=1e-6), stage                    # same KL decisions on every step
        for k, v in ref[stage].items():
            if k == "lr":
                continue
            assert torch.allclose(got[stage][k], v, atol=2e-4, rtol=2e-4), (stage, k)                  # AdamW-amplified rounding (see test_ppo_mimic_update_matches_reference)
            assert float((got[stage][k] - v).norm() / v.norm().clamp(min=1e-6)) < 2e-4, (stage, k)
    # the history encoder does not move in the PPO step, everything else does not move in the DAgger step
    hk = [k for k in ref["s1"] if "history_encoder" in k]
    assert hk and all(torch.equal(got["s1"][k], g["w0__" + k[2:]]) for k in hk)
    assert all(torch.equal(got["s2"][k], got["s1"][k]) for k in ref["s1"] if k != "lr" and "history_encoder" not in k)
    assert abs(got["hist_loss"] - ref["hist_loss"]) < 2e-3 * max(1.0, abs(ref["hist_loss"]))            # (each rank meters its own shard)
    # collectives: advantage moments (1) + one per PPO optimiser step + one per DAgger step
    assert got["all_reduces"] == 1 + 2 * got["steps"], (got["all_reduces"], got["steps"])
