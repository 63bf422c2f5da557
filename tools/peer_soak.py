#!/usr/bin/env python3
"""Soak of the peer-memory exchange protocol (csrc/peer_ops.hip): two processes on this box's GPU, N exchanges of varying sizes
queued back to back (both parities reused N / 2 times), one rank or the other delayed now and then, every result checked on the
device against the closed-form sum; the count of wrong results and the status word must stay zero.
usage: python tools/peer_soak.py [N=200000] [fences=0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, n, q):
    import torch, torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import coarse3d_amd  # noqa: F401
    from coarse3d_amd.peer import PeerExchange
    torch.cuda.set_device(0)
    px = PeerExchange(timeout_s=10.0)
    sizes = [64, 128, 1, 1408, 256, 8192, 2, 512, 1664, 33]
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    base = {s: torch.arange(s, dtype=torch.float64, device="cuda") for s in set(sizes)}
    tot_rank = sum(r + 1 for r in range(world))
    t0 = time.perf_counter()
    for it in range(n):
        s = sizes[it % len(sizes)]
        v = base[s] * (rank + 1) + float(it % 1000)
        if it % 997 == rank * 331:                   # now and then one rank is late
            torch.cuda._sleep(100_000 + 50_000 * (it % 7))
        px.allreduce_(v)
        bad += (v != base[s] * tot_rank + float(it % 1000) * world).sum()
        if it % 20000 == 19999:
            torch.cuda.synchronize()                 # (bounds the queue; also lets the two ranks drift apart and meet again)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    calls = px.check()
    q.put((rank, int(bad), calls, el))
    dist.barrier()
    px.close(collective=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    if len(sys.argv) > 2 and sys.argv[2] == "1":
        os.environ["C3D_PEER_FORCE_FENCES"] = "1"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29611, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=1800) for _ in range(2)]
    for p in procs:
        p.join(120)
    for rank, bad, calls, el in sorted(res):
        print(f"rank {rank}: {n} exchanges in {el:.1f} s ({el / n * 1e6:.1f} us each incl. the check), wrong results {bad}, exchanges counted {calls}", flush=True)
    assert all(r[1] == 0 for r in res), res
    print("ok")
