"""ppo_mimic.PPO maths restated (torch CPU fp32).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Networks   : agents/modules/agent_modules.py:11-166 (Actor = motion ConvEncoder + history ConvEncoder | priv MLP -> MLP;
             ActorCritic: sigma = clamp(std, min, max), critic on [obs, priv_obs, motion embedding]),
             encoder_modules.py:22-107 (per-step Linear+ReLU, two Conv1d + activation, Linear on 3 remaining steps),
             modules.py:5-66 (Linear/activation stack; `use_layernorm` is read by nobody)
Returns    : agents/ppo/ppo_mimic.py:440-491 (GAE; scalar reward: normalise over all [T,N,1] entries)
PPO update : ppo_mimic.py:596-691 (policy on the PRIV latent, KL-adaptive LR, clipped surrogate / value loss, entropy,
             priv_reg = mean ||enc_priv - sg(enc_hist)||_2 with the counter schedule, one clip_grad_norm_ over all
             parameters, one AdamW)
DAgger step: ppo_mimic.py:693-709 (history encoder regressed onto sg(enc_priv), its own AdamW)
Minibatches: data_utils.py:116-152 (one permutation, same slices every epoch)
"""
import math

import torch
import torch.nn.functional as F

from .ppo import gaussian_entropy, gaussian_log_prob

CONV = {5: ([20, 10], [2, 2], [1, 1]), 10: ([20, 10], [4, 2], [2, 1]), 20: ([40, 20], [6, 4], [2, 2])}     # encoder_modules.py:60-77


def _act(name):
    return dict(SiLU=F.silu, ELU=F.elu, ReLU=F.relu, Tanh=torch.tanh)[name]


def mlp(p, prefix, x, act):
    n = len([k for k in p if k.startswith(prefix + ".module.") and k.endswith(".weight")])
    for i in range(n):
        x = F.linear(x, p[f"{prefix}.module.{2 * i}.weight"], p[f"{prefix}.module.{2 * i}.bias"])
        if i < n - 1:
            x = act(x)
    return x


def conv_encoder(p, prefix, x, tsteps, act):
    W = p[prefix + ".encoder.0.weight"]
    x = x.reshape(-1, W.shape[1])                                    # x.view(-1, input_dim): chunks of the flat group, sic
    x = F.relu(F.linear(x, W, p[prefix + ".encoder.0.bias"]))
    x = x.view(-1, tsteps, W.shape[0]).permute(0, 2, 1)
    _, _, strides = CONV[tsteps]
    for i, s in enumerate(strides):
        x = act(F.conv1d(x, p[f"{prefix}.conv_module.{2 * i}.weight"], p[f"{prefix}.conv_module.{2 * i}.bias"], stride=s))
    return F.linear(x.flatten(start_dim=1), p[prefix + ".output_layer.weight"], p[prefix + ".output_layer.bias"])


class ActorCriticOracle:
    def __init__(self, params, mcfg, future_steps, hist_steps):
        """params: the reference ActorCritic.state_dict() (same key names)."""
        self.p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        a = mcfg.actor
        self.act_actor = _act(a.layer_config.activation)
        self.act_critic = _act(mcfg.critic.layer_config.activation)
        self.act_menc = _act(a.motion_encoder.layer_config.activation)
        self.act_henc = _act(a.history_encoder.layer_config.activation)
        self.act_priv = _act(a.priv_encoder.layer_config.activation)
        self.S, self.H = future_steps, hist_steps
        self.min_sigma, self.max_sigma = a.get("min_sigma", 0.1), a.get("max_sigma", 1.0)

    def motion(self, x):
        return conv_encoder(self.p, "actor_module.motion_encoder", x, self.S, self.act_menc)

    def history(self, x):
        return conv_encoder(self.p, "actor_module.history_encoder", x, self.H, self.act_henc)

    def priv(self, x):
        return mlp(self.p, "actor_module.priv_encoder", x, self.act_priv)

    def actor_mean(self, b, hist_encoding):
        latent = self.history(b["prop_history"]) if hist_encoding else self.priv(b["priv_obs"])
        return mlp(self.p, "actor_module.actor_module", torch.cat([b["actor_obs"], self.motion(b["future_motion_targets"]), latent], dim=-1), self.act_actor)

    def dist(self, b, hist_encoding):
        mean = self.actor_mean(b, hist_encoding)
        return mean, (mean * 0.0 + self.p["std"]).clamp(min=self.min_sigma, max=self.max_sigma)

    def evaluate(self, b):
        return mlp(self.p, "critic_module", torch.cat([b["actor_obs"], b["priv_obs"], self.motion(b["future_motion_targets"])], dim=-1), self.act_critic)


def compute_returns(rewards, values, dones, last_values, gamma, lam, vec=False):
    T = rewards.shape[0]
    returns = torch.zeros_like(values)
    adv = 0
    for t in reversed(range(T)):
        nxt = last_values if t == T - 1 else values[t + 1]
        nt = 1.0 - dones[t].float()
        delta = rewards[t] + nt * gamma * nxt - values[t]
        adv = delta + nt * gamma * lam * adv
        returns[t] = adv + values[t]
    tot = returns - values
    if not vec:
        return returns, (tot - tot.mean()) / (tot.std() + 1e-8)
    agg = tot.sum(dim=-1)
    return returns, ((agg - agg.mean()) / (agg.std() + 1e-8)).unsqueeze(-1)


class PPOMimicUpdate:
    def __init__(self, ac: ActorCriticOracle, cfg, counter=0):
        self.ac, self.cfg = ac, cfg
        self.lr = cfg.learning_rate
        self.counter = counter
        self.opt = torch.optim.AdamW(list(ac.p.values()), lr=self.lr)
        self.hist_params = [v for k, v in ac.p.items() if k.startswith("actor_module.history_encoder.")]
        self.hist_opt = torch.optim.AdamW(self.hist_params, lr=self.lr)

    def update_ppo(self, b):
        c, ac = self.cfg, self.ac
        mu, sigma = ac.dist(b, hist_encoding=False)
        logp = gaussian_log_prob(b["actions"], mu, sigma)
        value = ac.evaluate(b)
        entropy = gaussian_entropy(sigma)
        priv_latent = ac.priv(b["priv_obs"])
        with torch.no_grad():
            hist_latent = ac.history(b["prop_history"])
        priv_reg = (priv_latent - hist_latent).norm(p=2, dim=1).mean()
        sch = c.priv_reg_coef_schedual
        stage = min(max(self.counter - sch[2], 0) / sch[3], 1)
        coef = stage * (sch[1] - sch[0]) + sch[0]
        if c.desired_kl is not None and c.schedule == "adaptive":
            with torch.no_grad():
                old_s, old_m = b["action_sigma"], b["action_mean"]
                kl = torch.sum(torch.log(sigma / (old_s + 1e-5)) + (old_s ** 2 + (old_m - mu) ** 2) / (2.0 * sigma ** 2) - 0.5, axis=-1).mean()
                if kl > c.desired_kl * 2.0:
                    self.lr = max(1e-5, self.lr / 1.5)
                elif kl < c.desired_kl / 2.0 and kl > 0.0:
                    self.lr = min(1e-2, self.lr * 1.5)
                for g in self.opt.param_groups:
                    g["lr"] = self.lr
        adv = torch.squeeze(b["advantages"])
        ratio = torch.exp(logp - torch.squeeze(b["actions_log_prob"]))
        surrogate = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - c.clip_param, 1.0 + c.clip_param)).mean()
        if c.use_clipped_value_loss:
            vclip = b["values"] + (value - b["values"]).clamp(-c.clip_param, c.clip_param)
            vl = torch.max((value - b["returns"]).pow(2), (vclip - b["returns"]).pow(2)).sum(dim=-1).mean()
        else:
            vl = (b["returns"] - value).pow(2).sum(dim=-1).mean()
        ent = entropy.mean()
        total = surrogate - c.entropy_coef * ent + c.value_loss_coef * vl + coef * priv_reg
        self.opt.zero_grad()
        total.backward()
        torch.nn.utils.clip_grad_norm_(list(ac.p.values()), c.max_grad_norm)
        self.opt.step()
        return dict(Surrogate=surrogate.item(), Value=vl.item(), Entropy=ent.item(), priv_reg_loss=priv_reg.item())

    def update_dagger(self, b):
        ac = self.ac
        with torch.no_grad():
            priv_latent = ac.priv(b["priv_obs"])
        loss = (priv_latent - ac.history(b["prop_history"])).norm(p=2, dim=1).mean()
        self.hist_opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.hist_params, self.cfg.max_grad_norm)
        self.hist_opt.step()
        return dict(hist_latent_loss=loss.item())

    def _epochs(self, storage, perm, fn):
        c = self.cfg
        flat = {k: v.flatten(0, 1)[perm].contiguous() for k, v in storage.items()}
        mb = perm.numel() // c.num_mini_batches
        tot = {}
        for _ in range(c.num_learning_epochs):
            for i in range(c.num_mini_batches):
                for k, v in fn({k: v[i * mb:(i + 1) * mb] for k, v in flat.items()}).items():
                    tot[k] = tot.get(k, 0.0) + v
        n = c.num_learning_epochs * c.num_mini_batches
        self.counter += 1
        return {k: v / n for k, v in tot.items()}

    def training_step(self, storage, perm):
        return self._epochs(storage, perm, self.update_ppo)

    def training_step_dagger(self, storage, perm):
        return self._epochs(storage, perm, self.update_dagger)


class DistillUpdate:
    """Student distillation, DAgger-only (ppo_mimic.py:157-191,343-357,711-724): the student acts with its mean on the history latent,
    the frozen teacher actor acts on the teacher observation groups, bc_loss = mean ||a_teacher - mu_student||_2, one
    clip_grad_norm_ + AdamW over the student ACTOR's parameters (history encoder copied from the teacher and frozen)."""

    def __init__(self, student: ActorCriticOracle, teacher: ActorCriticOracle, cfg):
        self.s, self.t, self.cfg = student, teacher, cfg
        for k, v in student.p.items():
            if k.startswith("actor_module.history_encoder."):
                v.requires_grad_(False)
        self.actor_params = [v for k, v in student.p.items() if k.startswith("actor_module.")]
        self.opt = torch.optim.AdamW(self.actor_params, lr=cfg.learning_rate)

    def teacher_actions(self, b):
        tb = dict(b, actor_obs=b["teacher_actor_obs"], future_motion_targets=b["teacher_future_motion_targets"])
        with torch.no_grad():
            return self.t.actor_mean(tb, hist_encoding=True)

    def update(self, b):
        mu = self.s.actor_mean(b, hist_encoding=True)
        bc = (b["teacher_actions"] - mu).norm(p=2, dim=1).mean()
        self.opt.zero_grad()
        bc.backward()
        torch.nn.utils.clip_grad_norm_(self.actor_params, self.cfg.max_grad_norm)
        self.opt.step()
        return dict(bc_loss=bc.item())

    def training_step(self, storage, perm):
        c = self.cfg
        flat = {k: v.flatten(0, 1)[perm].contiguous() for k, v in storage.items()}
        mb = perm.numel() // c.num_mini_batches
        tot = 0.0
        for _ in range(c.num_learning_epochs):
            for i in range(c.num_mini_batches):
                tot += self.update({k: v[i * mb:(i + 1) * mb] for k, v in flat.items()})["bc_loss"]
        return dict(bc_loss=tot / (c.num_learning_epochs * c.num_mini_batches))
