"""Worker of tests/test_gpu_dp.py::test_captured_data_parallel_step_*: a fresh process with a 1-rank RCCL group
(backend "nccl", C3D_SINGLE_RANK_COLLECTIVES=1 so that every exchange point issues its real collective).  Runs the same
training steps under coarse3d_amd.dist.DataParallel launched kernel by kernel and as ONE captured hipGraph per step
(the RCCL collectives are nodes of the graph) and prints a JSON line with what differs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import coarse3d_amd  # noqa: E402,F401  (runtime defaults before the first GPU call)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import weights as W  # noqa: E402


def main():
    os.environ["C3D_SINGLE_RANK_COLLECTIVES"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29761")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from coarse3d_amd import dist as D
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    from coarse3d_amd.trainer import TrainStep
    b, h, w, ncls = 2, 32, 128, 20
    batches = [W.synthetic_batch(b, h, w, ncls, 300 + i, 0.01 + 0.01 * i, gh=8, gw=16) for i in range(6)]
    runs = []
    for warm in (1000, 2):                  # never captured / captured after two eager steps
        torch.manual_seed(21)
        m = SalsaNextProto(5, ncls, 20, 0, use_prototype=True).to(dev).train()
        dp = D.DataParallel(m)
        ts = TrainStep(dp, ncls, proto_loss=True, lr=2e-3, num_anchor=32, graph=True, graph_warmup=warm,
                       feature_mean=[1.0, 0.1, 0.2, 0.3, 0.4], feature_std=[2.0, 1.0, 1.5, 0.5, 1.0])
        assert type(ts.optimizer).__name__ == "FlatAdamW"
        torch.manual_seed(22)
        counts0 = dict(D.COUNTS)
        losses, host = [], []
        for i, (x, tr, ev) in enumerate(batches):
            if i == 3:
                ts.optimizer.param_groups[0]["lr"] = 5e-4      # what a scheduler does between steps
            xd, td, ed = x.to(dev), tr.to(dev), ev.to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = ts.step(xd, td, ed, epoch=10 + i)                # the epoch changes every step: ONE graph all the same
            host.append((time.perf_counter() - t0) * 1e3)
            losses.append({k: res[k].clone() for k in ("loss", "ce", "lov", "contrast")})
        torch.cuda.synchronize()
        state = {k: v.detach().clone() for k, v in m.state_dict().items()}
        counts = {k: D.COUNTS[k] - counts0[k] for k in counts0}
        runs.append((losses, state, ts.optimizer.state_dict(), ts, counts, host))
    out = {"graphs": [len([e for e in r[3]._graphs.values() if e["graph"] is not None]) for r in runs],
           "replays": [r[3]._replays for r in runs], "counts": [r[4] for r in runs],
           "host_ms_last_step": [r[5][-1] for r in runs], "diff": []}
    for i, (la, lb) in enumerate(zip(runs[0][0], runs[1][0])):
        for k in la:
            if not torch.equal(la[k], lb[k]):
                out["diff"].append(f"step {i} {k}: {float(la[k])} vs {float(lb[k])}")
    for k, v in runs[0][1].items():
        if not torch.equal(v, runs[1][1][k]):
            out["diff"].append(f"state {k}")
    for i, st in runs[0][2]["state"].items():
        for k in ("step", "exp_avg", "exp_avg_sq"):
            if not torch.equal(st[k], runs[1][2]["state"][i][k]):
                out["diff"].append(f"adamw {i} {k}")
    out["trained"] = float(runs[1][0][-1]["loss"]) != float(runs[1][0][0]["loss"])
    dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
