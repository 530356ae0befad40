"""Data-parallel helpers: envs shard over ranks (one process per GPU, `torch.distributed` over RCCL/xGMI; gloo on CPU in
the tests).  The reference is single-process; these three exchanges make G ranks x N envs behave like ONE
batch of G*N envs:

  * `allreduce_mean_`      ONE all-reduce of the flat actor+critic gradient buffer per optimiser step (≈5 MB fp32;
                           latency-bound on a 7 x 153 GB/s xGMI mesh, so a single un-chunked collective);
  * `global_normalize_`    advantage mean / unbiased std over all ranks' T*N samples (mh_ppo.py:392-394) from three
                           moments (sum, sum of squares, count) in one tiny all-reduce;
  * `kl_lr_rule_`          the adaptive-KL learning-rate rule (mh_ppo.py:455-466) on the all-reduced KL mean, so every
                           rank takes the same branch.
All three are plain tensor code and run on any device.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_mean_(flat: torch.Tensor):
    w = world()
    if w > 1:
        dist.all_reduce(flat)
        flat.div_(w)
    return flat


def global_normalize_(raw: torch.Tensor):
    """raw: this rank's un-normalised advantages; normalised in place with global moments."""
    x = raw.double()
    mom = torch.stack([x.sum(), (x * x).sum(), torch.tensor(float(raw.numel()), dtype=torch.float64, device=raw.device)])
    if world() > 1:
        dist.all_reduce(mom)
    mean = mom[0] / mom[2]
    var = (mom[1] - mom[2] * mean * mean) / (mom[2] - 1.0)
    raw.copy_((raw - mean.float()) / (var.clamp(min=0).sqrt().float() + 1e-8))
    return raw


def kl_lr_rule_(lr: torch.Tensor, kl_mean_local: torch.Tensor, desired_kl: float):
    """lr: device tensor of learning rates (updated in place); kl_mean_local: this rank's minibatch KL mean."""
    kl = kl_mean_local.clone()
    w = world()
    if w > 1:
        dist.all_reduce(kl)
        kl = kl / w
    up = kl > desired_kl * 2.0
    down = (kl < desired_kl / 2.0) & (kl > 0.0)
    lr.copy_(torch.where(up, torch.clamp(lr / 1.5, min=1e-5), torch.where(down, torch.clamp(lr * 1.5, max=1e-2), lr)))
    return lr
