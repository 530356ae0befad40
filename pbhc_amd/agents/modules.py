"""Actor / critic networks and the rollout buffer of the MHPPO agent (PyTorch-ROCm).

Plain `nn.Linear` stacks so that `state_dict()` keys equal the reference's
(`std`, `actor_module.module.{0,2,4,6}.{weight,bias}`, `critic_module.module.*`) and checkpoints load
both ways with `strict=True` (reference: humanoidverse/agents/modules/modules.py:5-66,
ppo_modules.py:11-99, data_utils.py:22-152).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .. import _lib
from torch.distributions import Normal


class BaseModule(nn.Module):
    def __init__(self, obs_dim_dict, module_config_dict):
        super().__init__()
        self.obs_dim_dict = obs_dim_dict
        self.module_config_dict = module_config_dict
        self.input_dim = 0
        for each in module_config_dict["input_dim"]:
            if each in obs_dim_dict:
                self.input_dim += obs_dim_dict[each]
            elif isinstance(each, (int, float)):
                self.input_dim += each
            else:
                raise ValueError(f"_calculate_input_dim - Unknown input type: {each}")
        self.output_dim = 0
        for each in module_config_dict["output_dim"]:
            if isinstance(each, (int, float)):
                self.output_dim += each
            else:
                raise ValueError(f"_calculate_output_dim - Unknown output type: {each}")
        lc = module_config_dict["layer_config"]
        if lc["type"] != "MLP":
            raise NotImplementedError(f"Unsupported layer type: {lc['type']}")
        hidden = list(lc["hidden_dims"])
        act = getattr(nn, lc["activation"])()
        layers = [nn.Linear(self.input_dim, hidden[0]), act]
        for l in range(len(hidden)):
            if l == len(hidden) - 1:
                layers.append(nn.Linear(hidden[l], self.output_dim))
            else:
                layers.append(nn.Linear(hidden[l], hidden[l + 1]))
                layers.append(act)
        self.module = nn.Sequential(*layers)
        from . import fused_mlp

        self._fused = bool(fused_mlp.supported(self.module))       # a flag, not the module object: the reference's exporters deepcopy the actor

    def forward(self, x):
        if self._fused and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and (x.requires_grad or self.module[0].weight.requires_grad):
            from . import fused_mlp

            return fused_mlp.forward(self.module, x)             # training: fused activation-backward / bias-gradient path
        if self._fused and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and not torch.is_grad_enabled():
            from . import fused_mlp

            if fused_mlp.FUSED_GEMM and all(m.weight.is_contiguous() and m.bias is not None for m in self.module if isinstance(m, nn.Linear)):
                return fused_mlp.forward_inference(self.module, x)   # rollout: bias + activation in the GEMM epilogue
        return self.module(x)


def apply_cat(mod, x_const, x_grad):
    """`mod(cat([x_const, x_grad], -1))` for a BaseModule whose leading input columns `x_const` (observations) carry no gradient: on the
    fused training path the first layer's input gradient is then formed for the `x_grad` columns only (fused_mlp._FusedMLPCat)"""
    if (mod._fused and x_grad.is_cuda and x_grad.dim() == 2 and torch.is_grad_enabled() and not x_const.requires_grad
            and (x_grad.requires_grad or mod.module[0].weight.requires_grad)):
        from . import fused_mlp

        return fused_mlp.forward_cat(mod.module, x_const, x_grad)
    return mod(torch.cat([x_const, x_grad], dim=-1))


def apply_into(mod, xfull, c0, parts):
    """`mod(xfull)` for a BaseModule after copying `parts` into xfull[:, c0:] (the columns [0, c0) — observations — are in place already): the
    fused training path without the `torch.cat` copy of the whole input (fused_mlp._FusedMLPInto); anything else: the concatenation"""
    if (mod._fused and xfull.is_cuda and xfull.dim() == 2 and xfull.is_contiguous() and torch.is_grad_enabled() and not xfull.requires_grad
            and sum(t.shape[1] for t in parts) == xfull.shape[1] - c0):
        from . import fused_mlp

        return fused_mlp.forward_into(mod.module, xfull, c0, list(parts))
    return mod(torch.cat([xfull[:, :c0], *parts], dim=-1))


class PPOActor(nn.Module):
    def __init__(self, obs_dim_dict, module_config_dict, num_actions, init_noise_std):
        super().__init__()
        for i, od in enumerate(module_config_dict["output_dim"]):
            if od == "robot_action_dim":
                module_config_dict["output_dim"][i] = num_actions
        self.actor_module = BaseModule(obs_dim_dict, module_config_dict)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args = False

    @property
    def actor(self):
        return self.actor_module

    def reset(self, dones=None):                 # API of the reference class (ppo_modules.py:45-49): stateless MLP, nothing to reset
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, actor_obs):
        mean = self.actor(actor_obs)
        self.distribution = Normal(mean, mean * 0.0 + self.std)

    def act(self, actor_obs, **kwargs):
        self.update_distribution(actor_obs)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_inference(self, actor_obs):
        return self.actor(actor_obs)


class PPOCritic(nn.Module):
    def __init__(self, obs_dim_dict, module_config_dict):
        super().__init__()
        self.critic_module = BaseModule(obs_dim_dict, module_config_dict)

    @property
    def critic(self):
        return self.critic_module

    def reset(self, dones=None):                 # ppo_modules.py:94-95
        pass

    def evaluate(self, critic_obs, **kwargs):
        return self.critic(critic_obs)


class RolloutStorage(nn.Module):
    """`[T, N, C]` buffers, contiguous in the env axis per step (one coalesced slab per key and step)."""

    def __init__(self, num_envs, num_transitions_per_env, device="cuda"):
        super().__init__()
        self.device = device
        self.num_transitions_per_env = num_transitions_per_env
        self.num_envs = num_envs
        self.step = 0
        self.stored_keys = []

    def register_key(self, key, shape=(), dtype=torch.float, pad_rows=False, tail_slab=False):
        """pad_rows: the [N, C] slab of each step is a view of [N, ceil32(C)] rows, so that the env kernel writes whole 128-B lines.
        tail_slab: one more slab behind the T of the buffer (`self.tail[key]`, [N, C]; `self.with_tail(key)`, [T + 1, N, C]) — the observations
        after the last step live there, so that one forward over [T + 1, N] rows also yields the bootstrap values of GAE."""
        assert not hasattr(self, key), key
        assert isinstance(shape, (list, tuple)), "shape must be a list or tuple"
        if pad_rows and len(shape) == 1:
            T = self.num_transitions_per_env
            full = torch.zeros((T + (1 if tail_slab else 0), self.num_envs, _lib.padded_width(shape[0])), dtype=dtype, device=self.device)[..., :shape[0]]
            buf = full[:T]
            if tail_slab:
                self.__dict__.setdefault("_full", {})[key] = full
        else:
            buf = torch.zeros((self.num_transitions_per_env, self.num_envs) + tuple(shape), dtype=dtype, device=self.device)
        self.register_buffer(key, buf, persistent=False)
        self.stored_keys.append(key)

    def with_tail(self, key):
        """[T + 1, N, C]: the T slabs of `key` and its tail slab (register_key(..., tail_slab=True))"""
        return self.__dict__["_full"][key]

    def update_key(self, key, data):
        assert not data.requires_grad
        assert self.step < self.num_transitions_per_env, "Rollout buffer overflow"
        getattr(self, key)[self.step].copy_(data)

    def increment_step(self):                    # data_utils.py:50-51 (the reference agents call it once per control step)
        self.step += 1

    def batch_update_data(self, key, data):      # data_utils.py:59-63
        assert not data.requires_grad
        getattr(self, key)[:] = data

    def query_key(self, key):                    # data_utils.py:95-97
        assert hasattr(self, key), key
        return getattr(self, key)

    def _gather(self, keys, indices):
        """{key: buffer.flatten(0, 1)[indices]} — every f32 key whose [T, N, C] buffer is a uniformly pitched run of rows (contiguous, or the
        128-byte-padded slabs) in ONE launch of `pbhc_gather_rows`; anything else (other dtypes, CPU) by torch indexing."""
        import ctypes as C

        out, jobs = {}, []
        rows = self.num_envs * self.num_transitions_per_env
        fast = indices.is_cuda and indices.dtype == torch.int64 and indices.is_contiguous() and indices.dim() == 1
        for k in keys:
            t = getattr(self, k)
            w = int(t[0, 0].numel())
            ok = (fast and t.dtype == torch.float32 and t.dim() >= 2 and w >= 1 and (t.dim() == 2 or (t.stride(-1) == 1 and t.dim() == 3))
                  and t.stride(0) == t.shape[1] * t.stride(1) and (t.dim() == 2 or t.stride(1) >= w))
            if not ok:
                out[k] = t.flatten(0, 1)[indices].contiguous()
                continue
            dst = torch.empty((indices.numel(),) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            out[k] = dst
            jobs.append((t.data_ptr(), dst.data_ptr(), w, int(t.stride(1))))
        MAXJ = _lib.K["PBHC_MAX_GATHER_JOBS"]
        for a in range(0, len(jobs), MAXJ):
            chunk = jobs[a:a + MAXJ]
            arr = (_lib._S["PbhcGatherJob"] * len(chunk))()
            for j, (src, dst, w, pitch) in zip(arr, chunk):
                j.src, j.dst, j.width, j.src_pitch = src, dst, w, pitch
            _lib.check(_lib.lib().pbhc_gather_rows(arr, len(chunk), indices.data_ptr(), indices.numel(), _lib.current_stream()), "pbhc_gather_rows")
        return out

    def clear(self):
        self.step = 0

    def mini_batch_generator(self, num_mini_batches, num_epochs=8, keys=None, indices=None, on_gather=None):
        """One permutation per call, the same contiguous slices every epoch (data_utils.py:134-152).
        `keys` restricts the gather to what the update reads (the reference gathers every key).  `on_gather(shuffled)`: called once with the
        shuffled [T * N, C] tensors — entries it adds are sliced into the minibatches like every other key."""
        batch_size = self.num_envs * self.num_transitions_per_env
        mb = batch_size // num_mini_batches
        if indices is None:
            indices = torch.randperm(batch_size, device=self.device)
        keys = self.stored_keys if keys is None else keys
        shuffled = self._gather(keys, indices)
        if on_gather is not None:
            on_gather(shuffled)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                yield {k: v[i * mb:(i + 1) * mb] for k, v in shuffled.items()}
