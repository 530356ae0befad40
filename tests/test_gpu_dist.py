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
    mp.spawn(_rank_main, args=(world, 29533, perms, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    assert torch.allclose(got["adv"], adv1[:, :n], atol=2e-5, rtol=1e-5)            # globally normalised advantages
    assert torch.allclose(got["lr"], lr1, rtol=1e-6)                                # same KL decisions
    for k, v in ref.items():
        assert torch.allclose(got[k], v, atol=2e-4, rtol=2e-4), k                   # Adam-amplified rounding, see test_mhppo_update_matches_reference
        assert float((got[k] - v).norm() / v.norm().clamp(min=1e-6)) < 1e-4, k
