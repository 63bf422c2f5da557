"""Data-parallel exchange semantics on CPU with gloo, world_size 2."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coarse3d_amd import dist as D
    g = torch.Generator().manual_seed(100 + rank)
    # (1) SyncBN: all-reduced fp64 (sum, sumsq) of two half-batches == statistics of the full batch
    x = torch.randn(4, 8, 5, 7, generator=g)
    sums = torch.stack([x.double().sum(dim=(0, 2, 3)), (x.double() ** 2).sum(dim=(0, 2, 3))], 1)
    D.allreduce_sum_(sums)
    # (2) prototype bank mean over ranks
    bank = torch.nn.functional.normalize(torch.randn(3, 4, 16, generator=g), dim=-1)
    mean_bank = D.world_mean(bank)
    # (3) flat gradient buffer mean
    params = [("a.weight", torch.zeros(4, 3)), ("b.bias", torch.zeros(5)), ("c.weight", torch.zeros(2, 2, 3, 3))]
    fg = D.FlatGradients(params, order=["c.weight", "a.weight"])
    for i, (n, p) in enumerate(params):
        fg.views[n].copy_(torch.full_like(p, float(rank + 1) * (i + 1)))
    fg.all_reduce_mean(n_chunks=3)
    q.put((rank, x, sums, bank, mean_bank, {n: v.clone() for n, v in fg.views.items()}, fg.names))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_points_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, x0, s0, b0, m0, g0, names0), (_, x1, s1, b1, m1, g1, _) = res
    full = torch.cat([x0, x1], 0).double()
    torch.testing.assert_close(s0[:, 0], full.sum(dim=(0, 2, 3)))
    torch.testing.assert_close(s0[:, 1], (full ** 2).sum(dim=(0, 2, 3)))
    torch.testing.assert_close(s0, s1)
    torch.testing.assert_close(m0, (b0 + b1) / 2)      # salsanext_proto.py:397-400: mean, NOT renormalised
    torch.testing.assert_close(m0, m1)
    assert names0[:2] == ["c.weight", "a.weight"]      # backward-completion order first
    for i, n in enumerate(["a.weight", "b.bias", "c.weight"]):
        torch.testing.assert_close(g0[n], torch.full_like(g0[n], 1.5 * (i + 1)))
        torch.testing.assert_close(g0[n], g1[n])
