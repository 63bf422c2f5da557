"""Data-parallel exchange semantics on CPU with gloo, world_size 2."""
import os

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
    params = [("downCntx.conv1.weight", torch.zeros(4, 3)), ("upBlock4.conv1.bias", torch.zeros(5)),
              ("projector.proj.0.weight", torch.zeros(2, 2, 3, 3))]
    fg = D.FlatGradients(params)
    fg.begin()
    for i, (n, p) in enumerate(params):
        fg.views[n].copy_(torch.full_like(p, float(rank + 1) * (i + 1)))
    # overlapped path: buckets are launched as blocks finish (backward order), then waited for
    for tag in D.BACKWARD_ORDER:
        fg.block_done(tag, min_bytes=16)
    fg.finish()
    # numpy payloads are pickled by value (no shared-memory handles that die with the worker)
    q.put((rank, x.numpy(), sums.numpy(), bank.numpy(), mean_bank.numpy(),
           {n: v.clone().numpy() for n, v in fg.views.items()}, fg.names))
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
    t = torch.from_numpy
    res = [(r, t(x), t(s_), t(b), t(m), {k: t(v) for k, v in g.items()}, names) for r, x, s_, b, m, g, names in res]
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
    assert names0 == ["projector.proj.0.weight", "upBlock4.conv1.bias", "downCntx.conv1.weight"]   # backward order
    for i, n in enumerate(["downCntx.conv1.weight", "upBlock4.conv1.bias", "projector.proj.0.weight"]):
        torch.testing.assert_close(g0[n], torch.full_like(g0[n], 1.5 * (i + 1)))
        torch.testing.assert_close(g0[n], g1[n])


def _syncbn_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd import dist as D
    m = SalsaNextProto(5, 20, 20, 0, use_prototype=True).train()
    plain = m._bn_exchange()                       # BatchNorm2d children: rank-local statistics (stock DDP semantics)
    m.resBlock1.bn1 = torch.nn.SyncBatchNorm(64)   # partially converted model: refuse, do not silently de-sync
    try:
        m._bn_exchange()
        partial = "accepted"
    except RuntimeError as e:
        partial = str(e)
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(SalsaNextProto(5, 20, 20, 0, use_prototype=True)).train()
    fn, w = m._bn_exchange()
    sums = torch.full((4, 2), float(rank + 1), dtype=torch.float64)
    fn(sums)
    m.eval()
    ev = m._bn_exchange()
    bank = m._bank_exchange()
    q.put((rank, plain == (None, 1), partial, w, sums.numpy(), ev == (None, 1), bank is D.world_mean,
           sorted(k for k, _ in m.named_parameters())[:3]))
    dist.barrier()
    dist.destroy_process_group()


def test_reference_syncbn_wrap_switches_the_statistics_exchange_on():
    """tasks/weak_segmentation/trainer.py:54-60 wraps the model with
    SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel.  The converted children must
    turn the fp64 statistics exchange on by themselves (world-2 gloo group, CPU)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + os.getpid() % 1000
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, plain_local, partial, w, sums, eval_local, bank_ok, names in res:
        assert plain_local
        assert "SyncBatchNorm" in partial and partial != "accepted"
        assert w == 2
        assert (sums == 3.0).all()                 # 1 + 2 summed over the two ranks
        assert eval_local                          # eval mode never exchanges (running statistics)
        assert bank_ok                             # salsanext_proto.py:397-400: bank mean whenever a group exists
        assert names == ["cls_head.bias", "cls_head.weight", "downCntx.bn1.bias"]   # state_dict names survive the conversion


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher must start two ranks itself (run.sh:1 of the
    reference does the same with torch.distributed.launch).  There is no GPU here, so both ranks
    stop at the "needs an MI355X" check: the parent has to relay that as a non-zero exit code."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: covered by tests/test_gpu_dp.py::test_bench_launches_two_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") == 2, r.stderr[-2000:]
    assert "rank exit codes" in r.stderr
    assert r.stdout.strip() == ""
