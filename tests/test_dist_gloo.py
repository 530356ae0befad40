"""The N>1 exchanges (gradient bucket all-reduce, global advantage normalisation, KL -> learning-rate rule) on
world_size = 2 with the gloo backend on CPU: two ranks with half the samples each must reproduce the single-process
result on the whole batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pbhc_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    full_adv = torch.randn(2 * 96, generator=g) * 3 + 1
    full_grad = torch.randn(2, 1000, generator=g)
    kls = torch.tensor([0.031, 0.015])
    mine = full_adv[rank * 96:(rank + 1) * 96].clone()
    pdist.global_normalize_(mine)
    flat = full_grad[rank].clone()
    pdist.allreduce_mean_(flat)
    lr = torch.tensor([1e-3, 1e-3])
    pdist.kl_lr_rule_(lr, kls[rank], 0.01)           # mean KL 0.023 > 0.02 -> both ranks divide by 1.5
    q.put((rank, mine.tolist(), flat.tolist(), lr.tolist()))      # plain lists: no shared-memory handles cross the queue
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchanges_match_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r, adv, flat, lr = q.get(timeout=120)
        out[r] = (torch.tensor(adv), torch.tensor(flat), torch.tensor(lr))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    full_adv = torch.randn(2 * 96, generator=g) * 3 + 1
    full_grad = torch.randn(2, 1000, generator=g)
    ref = (full_adv - full_adv.mean()) / (full_adv.std() + 1e-8)
    got = torch.cat([out[0][0], out[1][0]])
    assert torch.allclose(got, ref, atol=1e-5)
    assert torch.allclose(out[0][1], full_grad.mean(0), atol=1e-6) and torch.allclose(out[1][1], full_grad.mean(0), atol=1e-6)
    assert torch.allclose(out[0][2], torch.tensor([1e-3 / 1.5] * 2)) and torch.equal(out[0][2], out[1][2])


def test_single_process_is_identity_on_world_1():
    x = torch.randn(100)
    ref = (x - x.mean()) / (x.std() + 1e-8)
    assert torch.allclose(pdist.global_normalize_(x.clone()), ref, atol=1e-6)
    f = torch.randn(10)
    assert torch.equal(pdist.allreduce_mean_(f.clone()), f)


def _seed_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7)                                  # the same config.seed on every rank
    base = int(torch.randint(0, 2**62, (1,)).item())
    gen = pdist.host_generator("cpu")
    draws = torch.rand(8, generator=gen).tolist()         # rank 0: the global generator; rank 1: its own rank-keyed one
    q.put((rank, base, pdist.rank_seed(base), gen is None, draws))
    dist.barrier()
    dist.destroy_process_group()


def test_equal_torch_seeds_do_not_replicate_streams_across_ranks():
    """Philox keys were (seed, LOCAL env, step) with the seed drawn from the torch generator — equal on every rank under the
    usual config.seed.  The rank is now mixed into the key, and host-issued draws of ranks > 0 come from a rank-keyed generator."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = {}
    for _ in range(2):
        r, base, seed, is_global, draws = q.get(timeout=120)
        out[r] = (base, seed, is_global, draws)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][0] == out[1][0]                                   # same torch seed -> same base draw on both ranks ...
    assert out[0][1] == out[0][0] and out[1][1] != out[0][1]        # ... rank 0 keeps it (single-process streams unchanged), rank 1 does not
    assert out[0][2] and not out[1][2]
    assert out[0][3] != out[1][3]
    assert pdist.rank_seed(12345) == 12345 and pdist.host_generator("cpu") is None      # no process group: identity
