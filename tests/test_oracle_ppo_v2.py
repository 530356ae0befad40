"""oracle.ppo_v2 vs one rollout + one _training_step + one _training_step_dagger of the reference's own ppo_mimic.PPO
(tests/golden/ppo_v2.npz, teacher-29 configuration with narrowed layers)."""
import os

import numpy as np
import torch

from oracle import ppo, ppo_v2
from tests.helpers import GOLDEN, PPO_V2_NARROW, fixture_config


def load():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_v2.npz")).items()}
    cfg = fixture_config("v2_g1_29dof_teacher.yaml", 8, PPO_V2_NARROW)
    w0 = {k[len("w0__"):]: v for k, v in g.items() if k.startswith("w0__")}
    st = {k[len("st__"):]: v for k, v in g.items() if k.startswith("st__")}
    return g, cfg, w0, st


def test_ppo_mimic_maths_match_reference():
    g, cfg, w0, st = load()
    ac = ppo_v2.ActorCriticOracle(w0, cfg.algo.config.module_dict, cfg.obs.future_num_steps, cfg.obs.history_length)
    up = ppo_v2.PPOMimicUpdate(ac, cfg.algo.config, counter=int(g["counter0"]))
    c = cfg.algo.config
    with torch.no_grad():
        fl = {k: v.flatten(0, 1) for k, v in st.items()}
        mu, sigma = ac.dist(fl, hist_encoding=False)
        assert torch.allclose(mu, fl["action_mean"], atol=1e-5)
        assert torch.allclose(sigma, fl["action_sigma"], atol=1e-6)
        assert torch.allclose(ppo.gaussian_log_prob(fl["actions"], mu, sigma).unsqueeze(-1), fl["actions_log_prob"], atol=2e-4, rtol=1e-5)
        assert torch.allclose(ac.evaluate(fl), fl["values"], atol=1e-5)
        last = {k[len("last__"):]: v for k, v in g.items() if k.startswith("last__")}
        assert torch.allclose(ac.actor_mean(last, hist_encoding=True), g["infer_hist"], atol=1e-5)
        last_values = ac.evaluate(last)
        assert torch.allclose(last_values, g["last_values"], atol=1e-5)
        ret, adv = ppo_v2.compute_returns(st["rewards"], st["values"], st["dones"], last_values, c.gamma, c.lam)
        assert torch.allclose(ret, st["returns"], atol=1e-5, rtol=1e-5)
        assert torch.allclose(adv, st["advantages"], atol=1e-5, rtol=1e-5)
    loss = up.training_step(st, g["perm1"])
    for k in ["Value", "Surrogate", "Entropy", "priv_reg_loss"]:
        assert abs(loss[k] - float(g["loss1__" + k])) < 1e-4 * max(1.0, abs(loss[k])), (k, loss[k], float(g["loss1__" + k]))
    assert abs(up.lr - float(g["lr1"])) < 1e-12
    for k, v in ac.p.items():
        assert torch.allclose(v.detach(), g["w1__" + k], atol=2e-5, rtol=1e-4), k
    assert not torch.equal(g["w1__actor_module.priv_encoder.module.0.weight"], g["w0__actor_module.priv_encoder.module.0.weight"])
    assert torch.equal(g["w1__actor_module.history_encoder.encoder.0.weight"], g["w0__actor_module.history_encoder.encoder.0.weight"])   # no grad in the PPO step
    loss = up.training_step_dagger(st, g["perm2"])
    assert abs(loss["hist_latent_loss"] - float(g["loss2__hist_latent_loss"])) < 1e-4
    for k, v in ac.p.items():
        assert torch.allclose(v.detach(), g["w2__" + k], atol=2e-5, rtol=1e-4), k
    assert up.counter == int(g["counter2"])


def test_distillation_maths_match_reference():
    """Student distillation (DAgger-only) of the reference ppo_mimic.PPO: teacher actions from the teacher observation groups,
    student mean on the history latent, bc loss, AdamW over the actor (tests/golden/ppo_distill.npz)."""
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "ppo_distill.npz")).items() if v.dtype.kind in "fiub"}
    scfg = fixture_config("v2_g1_23dof_student.yaml", 8, PPO_V2_NARROW)
    tcfg = fixture_config("v2_g1_23dof_teacher.yaml", 8, PPO_V2_NARROW)
    S, H = scfg.obs.future_num_steps, scfg.obs.history_length
    student = ppo_v2.ActorCriticOracle({k[len("w0__"):]: v for k, v in g.items() if k.startswith("w0__")}, scfg.algo.config.module_dict, S, H)
    teacher = ppo_v2.ActorCriticOracle({k[len("teacher__"):]: v for k, v in g.items() if k.startswith("teacher__")}, tcfg.algo.config.module_dict, S, H)
    st = {k[len("st__"):]: v for k, v in g.items() if k.startswith("st__")}
    up = ppo_v2.DistillUpdate(student, teacher, scfg.algo.config)
    fl = {k: v.flatten(0, 1) for k, v in st.items()}
    assert torch.allclose(up.teacher_actions(fl), fl["teacher_actions"], atol=1e-5)
    with torch.no_grad():
        assert torch.allclose(student.actor_mean(fl, hist_encoding=True), fl["actions"], atol=1e-5)       # dagger_only rollouts act with the mean
    for k, v in student.p.items():          # the student's history encoder is the teacher's
        if k.startswith("actor_module.history_encoder."):
            assert torch.equal(v.detach(), g["teacher__" + k])
    loss = up.training_step(st, g["perm"])
    assert abs(loss["bc_loss"] - float(g["loss__bc_loss"])) < 1e-4
    for k, v in student.p.items():
        assert torch.allclose(v.detach(), g["w1__" + k], atol=2e-5, rtol=1e-4), k
