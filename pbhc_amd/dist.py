"""Data-parallel helpers: envs shard over ranks (one process per GPU, `torch.distributed` over RCCL/xGMI; gloo on CPU in
the tests).  The reference is single-process; these three exchanges make G ranks x N envs behave like ONE
batch of G*N envs:

  * `allreduce_mean_`      ONE all-reduce (RCCL: ReduceOp.AVG, no separate division) of the flat actor+critic gradient buffer per
                           optimiser step (≈5 MB fp32; latency-bound on a 7 x 153 GB/s xGMI mesh, so a single un-chunked collective);
  * `global_normalize_`    advantage mean / unbiased std over all ranks' T*N samples (mh_ppo.py:392-394) from three
                           moments (sum, sum of squares, count) in one tiny all-reduce;
  * `kl_lr_rule_`          the adaptive-KL learning-rate rule (mh_ppo.py:455-466) on the all-reduced KL mean, so every
                           rank takes the same branch.
All three are plain tensor code and run on any device.

Every product collective goes through `all_reduce` / `broadcast` below, which count calls and payload bytes
(`COUNTERS`; bench.py reports them per PPO iteration).  `PBHC_DIST_FORCE=1` makes a process group of ONE rank take the
distributed code path too (`active()`), so that the RCCL call pattern — async all-reduces on gradient-segment views,
the per-step statistics exchange next to hipGraph replays — can be rehearsed on a one-GPU box.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


COUNTERS = {"all_reduce": 0, "all_reduce_bytes": 0, "broadcast": 0}


def reset_counters():
    for k in COUNTERS:
        COUNTERS[k] = 0


def initialized():
    return dist.is_available() and dist.is_initialized()


def world(group=None):
    return dist.get_world_size(group) if initialized() else 1


def rank():
    return dist.get_rank() if initialized() else 0


def active(group=None):
    """True when the data-parallel exchanges must run: more than one rank, or a forced rehearsal on one."""
    return initialized() and (dist.get_world_size(group) > 1 or os.environ.get("PBHC_DIST_FORCE", "0") == "1")


def all_reduce(t: torch.Tensor, group=None, async_op=False):
    COUNTERS["all_reduce"] += 1
    COUNTERS["all_reduce_bytes"] += t.numel() * t.element_size()
    return dist.all_reduce(t, group=group, async_op=async_op)


def broadcast(t: torch.Tensor, src=0):
    COUNTERS["broadcast"] += 1
    return dist.broadcast(t, src=src)


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def rank_seed(seed: int, bits: int = 62) -> int:
    """Philox keys are (seed, LOCAL env index, step counter): with the same torch seed on every rank (the usual `config.seed`)
    G ranks x N envs would draw G copies of the same N streams.  Mix the rank into the key; rank 0 of a single process keeps `seed`."""
    r = rank()
    if r == 0:
        return seed
    return (seed ^ _splitmix64(r)) & ((1 << bits) - 1)


_GENERATORS = {}


def host_generator(device):
    """torch.Generator for host-issued draws (slot -> clip sampling, start phases, reset_all's episodic DR, the stub's per-env DR): None on
    rank 0 — the torch global generator, as the reference — and ONE rank-keyed generator per device on the other ranks."""
    if rank() == 0:
        return None
    key = str(device)
    if key not in _GENERATORS:
        g = torch.Generator(device=device)
        g.manual_seed(rank_seed(torch.initial_seed() & ((1 << 62) - 1)))
        _GENERATORS[key] = g
    return _GENERATORS[key]


def _avg_supported(group=None):
    """RCCL reduces with ReduceOp.AVG in the collective itself (no second pass over the bucket); gloo has no AVG"""
    return dist.get_backend(group) == "nccl"


def allreduce_mean_(flat: torch.Tensor, group=None):
    """in-place mean over the ranks: ONE collective (RCCL: ReduceOp.AVG; gloo: sum, then a division)"""
    if active(group):
        if _avg_supported(group):
            COUNTERS["all_reduce"] += 1
            COUNTERS["all_reduce_bytes"] += flat.numel() * flat.element_size()
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
        else:
            all_reduce(flat, group=group)
            flat.div_(world(group))
    return flat


def global_normalize_(raw: torch.Tensor):
    """raw: this rank's un-normalised advantages; normalised in place with global moments."""
    x = raw.double()
    mom = torch.stack([x.sum(), (x * x).sum(), torch.tensor(float(raw.numel()), dtype=torch.float64, device=raw.device)])
    if active():
        all_reduce(mom)
    mean = mom[0] / mom[2]
    var = (mom[1] - mom[2] * mean * mean) / (mom[2] - 1.0)
    raw.copy_((raw - mean.float()) / (var.clamp(min=0).sqrt().float() + 1e-8))
    return raw


def kl_lr_rule_(lr: torch.Tensor, kl_mean_local: torch.Tensor, desired_kl: float, reduced: bool = False):
    """lr: device tensor of learning rates (updated in place); kl_mean_local: this rank's minibatch KL mean — or, with `reduced`, the mean over
    all ranks already (the agents carry it inside a gradient bucket's all-reduce)."""
    kl = kl_mean_local.clone()
    if active() and not reduced:
        all_reduce(kl)
        kl = kl / world()
    up = kl > desired_kl * 2.0
    down = (kl < desired_kl / 2.0) & (kl > 0.0)
    lr.copy_(torch.where(up, torch.clamp(lr / 1.5, min=1e-5), torch.where(down, torch.clamp(lr * 1.5, max=1e-2), lr)))
    return lr
