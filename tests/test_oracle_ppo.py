"""oracle.ppo vs one rollout + one _training_step of the reference's own MHPPO (tests/golden/ppo_v1.npz)."""
import os

import numpy as np
import torch

from oracle import ppo
from tests.helpers import GOLDEN, fixture_config


def test_mhppo_maths_match_reference():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v1.npz")).items()}
    cfg = fixture_config("v1_g1_23dof_horse_stance.yaml", 8).algo.config
    ap = {k[len("actor__"):]: v for k, v in g.items() if k.startswith("actor__")}
    cp = {k[len("critic__"):]: v for k, v in g.items() if k.startswith("critic__")}
    up = ppo.MHPPOUpdate(ap, cp, cfg)
    st = {k[len("st__"):]: v for k, v in g.items() if k.startswith("st__")}
    with torch.no_grad():
        # stored rollout quantities are functions of the stored observations and the initial weights
        mu, sigma = up.actor_dist(st["actor_obs"])
        assert torch.allclose(mu, st["action_mean"], atol=1e-5)
        assert torch.allclose(sigma, st["action_sigma"], atol=1e-6)
        assert torch.allclose(ppo.gaussian_log_prob(st["actions"], mu, sigma).unsqueeze(-1), st["actions_log_prob"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(up.critic(st["critic_obs"]), st["values"], atol=1e-5)
        last_values = up.critic(g["last_critic_obs"])
        ret, adv = ppo.compute_returns(st["rewards"], st["values"], st["dones"], last_values, cfg.gamma, cfg.lam)
        assert torch.allclose(ret, st["returns"], atol=1e-5, rtol=1e-5)
        assert torch.allclose(adv, st["advantages"], atol=1e-5, rtol=1e-5)
    loss = up.training_step(st, g["perm"])
    for k in ["Value", "Surrogate", "Entropy"]:
        assert abs(loss[k] - float(g["loss__" + k])) < 1e-4 * max(1.0, abs(loss[k])), (k, loss[k], float(g["loss__" + k]))
    assert abs(up.lr_a - float(g["lr_actor"])) < 1e-12 and abs(up.lr_c - float(g["lr_critic"])) < 1e-12
    for k, v in up.ap.items():
        assert torch.allclose(v.detach(), g["actor1__" + k], atol=2e-5, rtol=1e-4), k
    for k, v in up.cp.items():
        assert torch.allclose(v.detach(), g["critic1__" + k], atol=2e-5, rtol=1e-4), k
