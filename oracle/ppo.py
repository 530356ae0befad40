"""MHPPO maths restated (torch CPU fp32).  TEST INFRASTRUCTURE (see oracle/__init__.py).

GAE / returns / advantage normalisation: reference humanoidverse/agents/mh_ppo/mh_ppo.py:348-395.
Update: mh_ppo.py:433-533 (analytic Gaussian KL -> adaptive LR, clipped surrogate, clipped value loss
summed over heads, entropy bonus, two losses / two clip_grad_norm_ / two Adam).
Minibatching: agents/modules/data_utils.py:116-152 (one permutation, the same 4 slices every epoch).
Networks: agents/modules/modules.py:47-63 (Linear/ELU stack), ppo_modules.py:11-99 (Normal(mean, std)).
"""
import math

import torch
import torch.nn.functional as F


def mlp_forward(params, prefix, x, n_layers=4):
    for i in range(n_layers):
        x = F.linear(x, params[f"{prefix}.module.{2 * i}.weight"], params[f"{prefix}.module.{2 * i}.bias"])
        if i < n_layers - 1:
            x = F.elu(x)
    return x


def compute_returns(rewards, values, dones, last_values, gamma, lam):
    """rewards/values [T,N,R], dones [T,N,1] bool, last_values [N,R] -> returns [T,N,R], advantages [T,N,1]."""
    T = rewards.shape[0]
    returns = torch.zeros_like(values)
    adv = 0
    for t in reversed(range(T)):
        nxt = last_values if t == T - 1 else values[t + 1]
        nt = 1.0 - dones[t].float()
        delta = rewards[t] + nt * gamma * nxt - values[t]
        adv = delta + nt * gamma * lam * adv
        returns[t] = adv + values[t]
    tot = (returns - values).sum(dim=-1)
    advantages = (tot - tot.mean()) / (tot.std() + 1e-8)
    return returns, advantages.unsqueeze(-1)


def gaussian_log_prob(x, mean, std):
    var = std * std
    return (-((x - mean) ** 2) / (2 * var) - torch.log(std) - math.log(math.sqrt(2 * math.pi))).sum(dim=-1)


def gaussian_entropy(std):
    return (0.5 + 0.5 * math.log(2 * math.pi) + torch.log(std)).sum(dim=-1)


class MHPPOUpdate:
    def __init__(self, actor_params, critic_params, cfg):
        """actor_params: dict with 'std' and 'actor_module.module.{0,2,4,6}.{weight,bias}';
        critic_params: 'critic_module.module.*' — the reference's state_dict key names."""
        self.ap = {k: v.clone().requires_grad_(True) for k, v in actor_params.items()}
        self.cp = {k: v.clone().requires_grad_(True) for k, v in critic_params.items()}
        self.cfg = cfg
        self.lr_a = cfg.actor_learning_rate
        self.lr_c = cfg.critic_learning_rate
        self.opt_a = torch.optim.Adam(list(self.ap.values()), lr=self.lr_a)
        self.opt_c = torch.optim.Adam(list(self.cp.values()), lr=self.lr_c)

    def actor_dist(self, obs):
        mean = mlp_forward(self.ap, "actor_module", obs)
        return mean, mean * 0.0 + self.ap["std"]

    def critic(self, obs):
        return mlp_forward(self.cp, "critic_module", obs)

    def update(self, b):
        c = self.cfg
        mu, sigma = self.actor_dist(b["actor_obs"])
        logp = gaussian_log_prob(b["actions"], mu, sigma)
        value = self.critic(b["critic_obs"])
        entropy = gaussian_entropy(sigma)
        if c.desired_kl is not None and c.schedule == "adaptive":
            with torch.no_grad():
                old_s, old_m = b["action_sigma"], b["action_mean"]
                kl = torch.sum(torch.log(sigma / old_s + 1.0e-5) + (old_s.square() + (old_m - mu).square()) / (2.0 * sigma.square()) - 0.5, axis=-1)
                kl_mean = kl.mean()
                if kl_mean > c.desired_kl * 2.0:
                    self.lr_a = max(1e-5, self.lr_a / 1.5)
                    self.lr_c = max(1e-5, self.lr_c / 1.5)
                elif kl_mean < c.desired_kl / 2.0 and kl_mean > 0.0:
                    self.lr_a = min(1e-2, self.lr_a * 1.5)
                    self.lr_c = min(1e-2, self.lr_c * 1.5)
                for g in self.opt_a.param_groups:
                    g["lr"] = self.lr_a
                for g in self.opt_c.param_groups:
                    g["lr"] = self.lr_c
        adv = b["advantages"].squeeze()
        ratio = torch.exp(logp - b["actions_log_prob"].squeeze())
        s1 = -adv * ratio
        s2 = -adv * torch.clamp(ratio, 1.0 - c.clip_param, 1.0 + c.clip_param)
        surrogate = torch.max(s1, s2).mean()
        if c.use_clipped_value_loss:
            vclip = b["values"] + (value - b["values"]).clamp(-c.clip_param, c.clip_param)
            vl = torch.max((value - b["returns"]).pow(2), (vclip - b["returns"]).pow(2)).sum(dim=-1).mean()
        else:
            vl = (b["returns"] - value).pow(2).sum(dim=-1).mean()
        ent = entropy.mean()
        actor_loss = surrogate - c.entropy_coef * ent
        critic_loss = c.value_loss_coef * vl
        self.opt_a.zero_grad()
        self.opt_c.zero_grad()
        actor_loss.backward()
        critic_loss.backward()
        torch.nn.utils.clip_grad_norm_(list(self.ap.values()), c.max_grad_norm)
        torch.nn.utils.clip_grad_norm_(list(self.cp.values()), c.max_grad_norm)
        self.opt_a.step()
        self.opt_c.step()
        return dict(Value=vl.item(), Surrogate=surrogate.item(), Entropy=ent.item())

    def training_step(self, storage, perm):
        """storage: dict key -> [T,N,...]; perm: the permutation of T*N the reference drew."""
        c = self.cfg
        flat = {k: v.flatten(0, 1)[perm].contiguous() for k, v in storage.items()}
        B = perm.numel()
        mb = B // c.num_mini_batches
        tot = dict(Value=0.0, Surrogate=0.0, Entropy=0.0)
        for _ in range(c.num_learning_epochs):
            for i in range(c.num_mini_batches):
                out = self.update({k: v[i * mb:(i + 1) * mb] for k, v in flat.items()})
                for k in tot:
                    tot[k] += out[k]
        n = c.num_learning_epochs * c.num_mini_batches
        return {k: v / n for k, v in tot.items()}
