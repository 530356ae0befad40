"""One full PPO iteration (24-step rollout + GAE + 5x4 minibatch update) on the CPU oracle — the timed
`cpu_baseline` ("port") leg of bench.py and nothing else.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import time

import numpy as np
import torch

from . import ppo
from .env_v1 import MotionTrackingOracle
from .fk import sim_fk
from .motion_lib import MotionLib


def init_params(in_dim, hidden, out_dim, prefix, gen):
    p = {}
    dims = [in_dim] + list(hidden) + [out_dim]
    for i in range(len(dims) - 1):
        bound = 1.0 / np.sqrt(dims[i])
        p[f"{prefix}.module.{2 * i}.weight"] = (torch.rand(dims[i + 1], dims[i], generator=gen) * 2 - 1) * bound
        p[f"{prefix}.module.{2 * i}.bias"] = (torch.rand(dims[i + 1], generator=gen) * 2 - 1) * bound
    return p


def run_iteration(cfg, skel, clip, num_envs, seed=0, return_state=False):
    """Returns dict(env_steps, seconds, rollout_s, update_s) [+ the rollout buffer, the actor's weights before / after and the adapted
    learning rate with return_state]."""
    g = torch.Generator().manual_seed(seed)
    N = num_envs
    ml = MotionLib(skel, [clip])
    D = skel["dof_axis"].shape[0]
    dr = dict(base_com_bias=torch.zeros(N, 3), link_mass_scale=torch.ones(N, len(cfg.domain_rand.randomize_link_body_names)), friction_coeffs=torch.ones(N, 1, 1))
    env = MotionTrackingOracle(cfg, skel, ml, N, dr)
    L = float(ml.motion_len[0])
    env.s["motion_len"][:] = L
    env.s["motion_start_times"] = torch.rand(N, generator=g) * L * 0.5
    acfg = cfg.algo.config
    T = acfg.num_steps_per_env
    obs_dims = {"actor_obs": 380, "critic_obs": 630}
    # replay window (setup, not timed)
    from tests.helpers import synth_replay
    root, qp, qv, cf = synth_replay(ml, skel, N, T + 1, env.s["motion_start_times"], env.s["episode_length_buf"], env.dt, env.env_origins, seed, env.feet)
    frame0 = dict(root=root[0], dof_pos=qp[0], dof_vel=qv[0], contact=cf[0])
    obs, _, _, _ = env.step(torch.zeros(N, D), frame0, sim_fk(skel, root[0], qp[0], qv[0]), reset_samples=_samples(N, D, L, g))
    obs_dims = {k: v.shape[1] for k, v in obs.items()}
    ap = init_params(obs_dims["actor_obs"], acfg.module_dict.actor.layer_config.hidden_dims, D, "actor_module", g)
    ap["std"] = acfg.init_noise_std * torch.ones(D)
    cp = init_params(obs_dims["critic_obs"], acfg.module_dict.critic.layer_config.hidden_dims, env.R, "critic_module", g)
    up = ppo.MHPPOUpdate(ap, cp, acfg)
    ap0 = {k: v.clone() for k, v in ap.items()} if return_state else None
    st = {k: torch.zeros(T, N, d) for k, d in obs_dims.items()}
    st.update(actions=torch.zeros(T, N, D), rewards=torch.zeros(T, N, env.R), dones=torch.zeros(T, N, 1, dtype=torch.bool), values=torch.zeros(T, N, env.R),
              actions_log_prob=torch.zeros(T, N, 1), action_mean=torch.zeros(T, N, D), action_sigma=torch.zeros(T, N, D))
    t0 = time.perf_counter()
    with torch.no_grad():
        for t in range(T):
            mu, sigma = up.actor_dist(obs["actor_obs"])
            act = mu + sigma * torch.randn(mu.shape, generator=g)
            val = up.critic(obs["critic_obs"])
            for k in obs:
                st[k][t] = obs[k]
            st["actions"][t] = act; st["action_mean"][t] = mu; st["action_sigma"][t] = sigma
            st["actions_log_prob"][t] = ppo.gaussian_log_prob(act, mu, sigma).unsqueeze(1); st["values"][t] = val
            frame = dict(root=root[t + 1], dof_pos=qp[t + 1], dof_vel=qv[t + 1], contact=cf[t + 1])
            body = sim_fk(skel, frame["root"], frame["dof_pos"], frame["dof_vel"])          # the sim-stub's FK is part of the path
            obs, rew, reset, ex = env.step(act, frame, body, u_rfi=torch.rand(N, D, generator=g), reset_samples=_samples(N, D, L, g))
            st["rewards"][t] = rew + acfg.gamma * val * ex["time_outs"].unsqueeze(1)
            st["dones"][t] = reset.unsqueeze(1).bool()
        last = up.critic(obs["critic_obs"])
        st["returns"], st["advantages"] = ppo.compute_returns(st["rewards"], st["values"], st["dones"], last, acfg.gamma, acfg.lam)
    t1 = time.perf_counter()
    up.training_step(st, torch.randperm(T * N, generator=g))
    t2 = time.perf_counter()
    out = dict(env_steps=T * N, seconds=t2 - t0, rollout_s=t1 - t0, update_s=t2 - t1)
    if return_state:
        out.update(state=st, actor_params0=ap0, actor_params={k: v.detach().clone() for k, v in up.ap.items()}, lr=float(up.lr_a))
    return out


def _samples(N, D, L, g):
    return dict(motion_start_times=torch.rand(N, generator=g) * L, kp_scale=0.9 + 0.2 * torch.rand(N, D, generator=g),
                kd_scale=0.9 + 0.2 * torch.rand(N, D, generator=g), rfi_lim_scale=0.5 + torch.rand(N, D, generator=g),
                rao_scale=0.1 * (torch.rand(N, D, generator=g) - 0.5), action_delay_idx=torch.randint(0, 3, (N,), generator=g))
