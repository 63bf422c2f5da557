#!/usr/bin/env python3
"""Headline benchmark: range-images/sec of one COARSE3D training step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = input normalisation -> SalsaNextProto forward (return_feat, use_prototype, proto_loss)
-> focal + Lovasz -> entropy-based pseudo-label selection -> prototype contrastive loss
(512 anchors/class, T=0.07) -> backward -> AdamW, on synthetic 64x2048x5 range images, bs=8 per
GPU, fp32 (BASELINE.json configs[1]; BASELINE.md section 3 input recipe).  Inputs are resident in
HBM before the timed region.  Rank 0 prints ONE JSON line.

Two passes on one GPU (``--graph auto``, the default): W + K steps launched kernel by kernel, then the same W + K steps
each replayed as ONE hipGraph (``TrainStep(graph=True)``, bit-identical to the first pass: tests/test_gpu_step.py).
`value` / `ms_per_step` are the K timed steps of the captured pass -- the launch mode that does not depend on how busy
the host is (launch by launch the host needs ~20 ms per step; the same run's `launch_by_launch` reports that pass).
More than one rank: launch by launch (the data-parallel exchanges stay eager).

`roofline`: every launch of the dominant kernel (the template instance with the largest total time; name in
roofline.kernel, the rocprofv3 summary in profiles/ lists the same name) is bracketed by HIP events on the launch stream
during the timed steps of the kernel-by-kernel pass (events cannot bracket launches inside a replayed graph), and
`achieved` = algorithmic FLOPs of those launches / their summed duration.
`cpu_baseline`: the CPU oracle (a port: oracle/coarse3d_oracle.py) timed on this host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FEATURE_MEAN = [12.12, 10.88, 0.23, -1.04, 0.21]     # config_semantic_kitti.yaml sensor.img_means
FEATURE_STD = [12.32, 11.47, 6.91, 0.86, 0.16]       # config_semantic_kitti.yaml:148-153
PEAK_FP32_MFMA_TFLOPS = 157.3                        # MI355X_MICROARCH.md chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0                       # dense bf16 (no sparsity), same guide
# committed per-kernel HBM-traffic captures (tools/profile_round.sh <tag> <bench args>): one per (workload, engine)
PMC_FILE = "round3_{tag}_hbm.json"


def pmc_tag(args):
    """Tag of the committed rocprofv3 capture that matches this command line (workload + matrix engine), or None."""
    if args.net != "salsanext":
        return None
    key = (args.height, args.width, args.classes, args.dataset, args.batch)
    wl = {(64, 2048, 20, "SemanticKitti", 8): "kitti", (32, 1024, 17, "SemanticKitti", 16): "nuscenes",
          (32, 1024, 17, "nuScenes", 16): "nuscenes", (40, 1800, 14, "SemanticPOSS", 8): "poss"}.get(key)
    if wl is None:
        return None
    if args.matrix_dtype == "bf16" and args.storage != "bf16":
        return None
    return f"{wl}_{args.matrix_dtype}"


def synth_batch(b, h, w, ncls, seed, device, label_rate=1e-3):
    """x ~ N(0,1); blocky eval labels on an (H/8)x(W/64) grid; weak labels
    = eval * Bernoulli(rate)."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.randn(b, 5, h, w, generator=g, device=device)
    grid = torch.randint(0, ncls, (b, (h + 7) // 8, (w + 63) // 64), generator=g, device=device)
    ev = grid.repeat_interleave(8, 1).repeat_interleave(64, 2)[:, :h, :w].contiguous()   # 40x1800: ragged edge cells
    keep = torch.rand(b, h, w, generator=g, device=device) < label_rate
    return x, (ev * keep).long(), ev.long()


def host_cpu():
    """(model string, physical cores of socket 0, sockets) from /proc/cpuinfo."""
    model, cores, sockets = "unknown", set(), set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
                sockets.add(v)
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                if phys == min(sockets):
                    cores.add(core)
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), max(len(sockets), 1)


def cpu_baseline(ncls, h, w, budget_s=240.0, bs=2, timed_steps=3):
    """the CPU oracle's train step on this host, bs=2, fp32, one process,
    torch.set_num_threads(physical cores of one socket), one warm-up step (oneDNN primitive
    creation) + ``timed_steps`` timed steps, median reported.  Bounded: stops early when the
    budget is spent (the sample line says how many steps were timed)."""
    from oracle import coarse3d_oracle as oc
    model, cores, sockets = host_cpu()
    cores = min(cores, os.cpu_count() or cores)
    torch.set_num_threads(cores)
    st = oc.init_state(nclasses=ncls, seed=1)
    for k in oc.trainable_names(st):
        st[k].requires_grad_(True)
    mean, std = torch.tensor(FEATURE_MEAN), torch.tensor(FEATURE_STD)
    times = []
    t_all = time.time()
    for s in range(1 + timed_steps):
        x, tr, ev = synth_batch(bs, h, w, ncls, 1000 + s, "cpu")
        masks = {k: (torch.rand(bs, c) >= 0.2).float() * 1.25 for k, c in (
            ("resBlock2.dropout", 128), ("resBlock3.dropout", 256), ("resBlock4.dropout", 256),
            ("resBlock5.dropout", 256), ("upBlock1.dropout1", 64), ("upBlock1.dropout2", 320),
            ("upBlock1.dropout3", 128), ("upBlock2.dropout1", 32), ("upBlock2.dropout2", 288),
            ("upBlock2.dropout3", 128), ("upBlock3.dropout1", 32), ("upBlock3.dropout2", 160),
            ("upBlock3.dropout3", 64))}
        t0 = time.time()
        info, grads = oc.train_step(st, x, tr, ev, None, temperature=0.07, num_anchor=512, dropout_masks=masks,
                                    mean=mean, std=std)
        with torch.no_grad():
            for k, g in grads.items():
                if g is not None:
                    oc.adamw_update(st[k], g, torch.zeros_like(g), torch.zeros_like(g), 1, 1e-3)
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    sec = float(np.median(timed))
    return {"value": round(bs / sec, 4), "unit": "range-images/sec", "cores": cores, "kind": "port",
            "cpu": f"{model} ({sockets} socket(s), {cores} physical cores used = one socket)",
            "sample": f"oracle train step, bs={bs}, {h}x{w}x5, C={ncls}, fp32, median of {len(timed)} timed step(s) "
                      f"after {len(times) - len(timed)} warm-up (first step {times[0]:.1f} s incl. oneDNN warm-up), "
                      f"torch.set_num_threads({cores})",
            "sec_per_image": round(sec / bs, 3)}


def cpu_baseline_rangenet(ncls, h, w, layers, budget_s=150.0, family="rangenet"):
    """RangeNet / SqueezeSegV3 oracle on the host cores, bounded sample: forward + backward of the
    backbone and embedding branch at bs=1 (the losses and the prototype step are negligible beside it)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import weights as W
    if family == "rangenet":
        from oracle import rangenet_oracle as ro
        st = W.rangenet_state(layers=layers, nclasses=ncls)
        fwd = ro.rangenet_forward
    else:
        from oracle import squeezeseg_oracle as ro
        st = W.squeezeseg_state(layers=layers, nclasses=ncls)
        fwd = lambda st_, x_, tr_, m_, rf_, layers_: ro.squeezeseg_forward(st_, x_, tr_, m_, rf_, layers_)   # noqa: E731
    model_cpu, cores, _ = host_cpu()
    cores = min(cores, os.cpu_count() or cores)
    torch.set_num_threads(cores)
    names = [k for k in ro.trainable_names(st) if not k.startswith(("head1.", "head2.", "head3.", "head4."))]
    for k in names:
        st[k].requires_grad_(True)
    times = []
    t_all = time.time()
    for s in range(3):
        g = torch.Generator().manual_seed(s)
        x = torch.randn(1, 5, h, w, generator=g)
        t0 = time.time()
        out = fwd(st, x, True, None, True, layers)
        loss = (out["pred_2d"] * torch.randn(out["pred_2d"].shape, generator=g)).sum() + out["feat_2d"].sum()
        torch.autograd.grad(loss, [st[k] for k in names], allow_unused=True)
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    timed = times[1:] if len(times) > 1 else times
    sec = float(np.median(timed))
    return {"value": round(1.0 / sec, 4), "unit": "range-images/sec", "cores": cores, "kind": "port",
            "cpu": model_cpu,
            "sample": f"{family}-{layers} oracle forward+backward, bs=1, {h}x{w}x5, C={ncls}, fp32, {len(timed)} timed "
                      f"pass(es) after {len(times) - len(timed)} warm-up, torch.set_num_threads({cores})",
            "sec_per_image": round(sec, 3)}


def launch_ranks(n):
    """One process per GPU, as the reference's launcher does (tasks/weak_segmentation/run.sh:1:
    `python -m torch.distributed.launch --nproc_per_node=N`): start N copies of this script with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's single JSON line, fail if any rank
    fails.  Runs BEFORE anything initialises the GPU in this process (children are fresh
    interpreters; this parent never calls into HIP)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
    # this pool's host driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 both RCCL's peer-to-peer
    # setup and torch's cross-process tensor sharing fail with `hipIpcGetMemHandle: invalid argument`.  The image
    # exports it already; it is only ADDED here when the caller's environment has lost it, never overridden.
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):     # one rank died: the others would hang in a collective
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()                                  # exactly the children started above
        time.sleep(0.2)
    reader.join(10)
    out0 = buf[0] if buf else ""
    rcs = [p.returncode for p in procs]
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)               # anything rank 0 printed before its JSON line
    if any(rcs):
        for ln in lines[-1:]:
            print(ln, file=sys.stderr)
        print(f"bench.py: rank exit codes {rcs}", file=sys.stderr)
        return 1
    if not lines:
        print("bench.py: rank 0 printed nothing", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
    return 0


def default_shape_for_step(args):
    return (args.net, args.height, args.width, args.classes, args.dataset) == ("salsanext", 64, 2048, 20, "SemanticKitti")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--dataset", default="SemanticKitti")
    ap.add_argument("--net", choices=("salsanext", "rangenet21", "rangenet53", "squeezeseg21", "squeezeseg53"), default="salsanext",
                    help="backbone (default: SalsaNextProto, the BASELINE workload; rangenet* / squeezeseg*: SURVEY 8f N3)")
    ap.add_argument("--matrix-dtype", choices=("f32", "bf16", "bf16x3"), default="bf16x3",
                    help="matrix engine of conv / input-gradient kernels.  bf16x3 (default): every fp32 operand split "
                         "EXACTLY into three bf16 planes, eight of the nine plane products accumulated in fp32 on the "
                         "bf16 MFMA pipe -- fp32-class results (the whole -m gpu parity suite passes in this "
                         "mode: tests/test_gpu_configs.py::test_whole_gpu_suite_passes_on_the_exact_split_bf16_engine); "
                         "f32: the fp32-MFMA engine (also timed, reported under `engines`); bf16: opt-in mixed "
                         "precision (operands rounded to bf16, fp32 accumulate and storage; BASELINE configs[2])")
    ap.add_argument("--no-second-engine", action="store_true", help="skip the fp32-MFMA engine's comparison run")
    ap.add_argument("--storage", choices=("bf16", "f32"), default="bf16",
                    help="activation storage of --matrix-dtype bf16 (BASELINE configs[2]): bf16 tensors in HBM "
                         "(default) or fp32 tensors with bf16 MFMA operands only")
    ap.add_argument("--graph", nargs="?", const="on", default="auto", choices=("auto", "on", "off"),
                    help="how the step is launched.  auto (default; one GPU, SalsaNext): TWO passes of W + K steps over the same "
                         "batches -- launch by launch, with live HIP events around the dominant kernel inside its timed region "
                         "(`roofline`, `launch_by_launch`), then replayed as ONE hipGraph per step (TrainStep(graph=True), "
                         "bit-identical: tests/test_gpu_step.py), whose K timed steps give `value` (the host needs ~20 ms per "
                         "step launch by launch; on a box whose host is busy that, not the GPU, bounds the step).  on: only the "
                         "captured step (per-kernel figures from its eager warm-up step).  off: launch by launch only")
    ap.add_argument("--prewarm", type=int, default=3,
                    help="untimed steps ahead of the W warm-up steps (one-time costs of a fresh box; profiling runs pass 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--kernel-table", default=None, help="write a per-(kernel, layer shape) timing table (JSON) here")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  Nothing above or in
        # launch_ranks() touches the GPU (no torch.cuda call), so starting children is safe.
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; "
              f"measuring {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()       # (2 ranks on a 1-GPU box: debug only)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    single_rank_group = world == 1 and bool(os.environ.get("C3D_SINGLE_RANK_COLLECTIVES"))   # RCCL smoke test
    if world > 1 or single_rank_group:
        # "nccl" is RCCL on ROCm; C3D_DIST_BACKEND=gloo lets two ranks share one GPU for testing
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = os.environ.get("C3D_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    from coarse3d_amd import dist as D
    from coarse3d_amd import ops
    from coarse3d_amd.pc_processor.models import RangeNetProto, SalsaNextProto, SqueezeSegV3Proto
    from coarse3d_amd.trainer import TrainStep

    ops.set_matrix_precision(args.matrix_dtype, storage=args.storage if args.matrix_dtype == "bf16" else None)
    # bf16x3: EIGHT bf16 MFMAs (32x32x16) do the work of eight fp32 MFMAs' worth of K... i.e. per fp32 product
    # eight plane products: the fp32-equivalent ceiling of that engine is the dense bf16 peak / 8
    peak_tf = {"f32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS,
               "bf16x3": PEAK_BF16_MFMA_TFLOPS / 8.0}[args.matrix_dtype]

    def peak_for(kernel_name):
        """Matrix-pipe ceiling of one kernel instance: conv_bfp / conv_x3 / wgrad_tr instances run on the bf16
        pipe (eight plane products per fp32 product in the "<3" instances); the rest on fp32 MFMA."""
        if kernel_name.startswith("conv_pw3f_kernel"):       # <NT, WN>: the fused bf16x3 kernel, six plane products
            return PEAK_BF16_MFMA_TFLOPS / 6.0
        if kernel_name.startswith("conv_pw3_kernel"):        # <NT, NP>; NP = 3 runs six plane products
            return PEAK_BF16_MFMA_TFLOPS / 6.0 if kernel_name.endswith(", 3>") else PEAK_BF16_MFMA_TFLOPS
        if kernel_name.startswith(("conv_x3_kernel", "conv_x3f_kernel")):   # <NT, HALO, TT, SIX>
            return PEAK_BF16_MFMA_TFLOPS / (6.0 if kernel_name.endswith("true>") else 8.0)
        if kernel_name.startswith("conv_bfp_kernel"):
            return PEAK_BF16_MFMA_TFLOPS / 8.0 if kernel_name.rstrip(">").endswith("3") else PEAK_BF16_MFMA_TFLOPS
        if kernel_name.startswith("wgrad_mfma_kernel") and kernel_name.endswith("true>"):
            return PEAK_BF16_MFMA_TFLOPS
        if kernel_name.startswith("wgrad_tr_kernel"):
            return PEAK_BF16_MFMA_TFLOPS / 6.0 if kernel_name.startswith("wgrad_tr_kernel<3") else PEAK_BF16_MFMA_TFLOPS   # six plane products
        return PEAK_FP32_MFMA_TFLOPS
    torch.manual_seed(1)
    if args.net == "salsanext":
        model = SalsaNextProto(5, args.classes, 20, 0, use_prototype=True, dataset=args.dataset)
    elif args.net.startswith("rangenet"):
        model = RangeNetProto(layers=int(args.net[-2:]), nclasses=args.classes, dataset=args.dataset, use_prototype=True)
    else:
        model = SqueezeSegV3Proto(nclasses=args.classes, layers=int(args.net[-2:]), dataset=args.dataset, use_prototype=True)
    model = model.to(dev).train()
    wrapped = D.DataParallel(model) if (world > 1 or single_rank_group or os.environ.get("C3D_FORCE_DP")) else model
    ts = TrainStep(wrapped, args.classes, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512,
                   loss_w_ce_2d=1.0, loss_w_lov_2d=1.0, loss_w_contrast=0.1, feature_mean=FEATURE_MEAN,
                   feature_std=FEATURE_STD, proto_loss=True, graph=args.graph == "on" and world == 1 and not single_rank_group,
                   inputs_resident=True)      # the batches below are generated and synchronised before the timed region
    if ts.graph and args.warmup < 3:
        args.warmup = 3                       # two eager steps + the capture
    rate = 1e-4 if args.dataset == "SemanticPOSS" else 1e-3
    total_steps = args.warmup + args.steps
    batches = [synth_batch(args.batch, args.height, args.width, args.classes, 1000 + s + 7919 * rank, dev, rate)
               for s in range(total_steps)]
    torch.cuda.synchronize()               # inputs resident in HBM before the first step touches them

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def summarise(events, nsteps):
        """{kernel instance: [flops, seconds, launches]} and the per-(instance, layer shape) table."""
        per, table = {}, {}
        for name, flops, e0, e1, detail in events:
            sec_ = e0.elapsed_time(e1) * 1e-3
            for dd, key in ((per, name), (table, (name, detail))):
                d = dd.setdefault(key, [0.0, 0.0, 0])
                d[0] += flops
                d[1] += sec_
                d[2] += 1
        return per, table

    # Warm-up.  The LAST warm-up step brackets EVERY MFMA launch with HIP events on the launch
    # stream: that picks the dominant kernel instance and gives the all-kernel summary.  The timed
    # steps then bracket only the launches of that dominant instance (bracketing all ~300 launches
    # per step costs ~2 % of the step, which would distort `value`).
    survey = None
    # Three extra untimed steps ahead of the W warm-up steps: on 2 of 5 fresh boxes of round 3 the FIRST bench process
    # averaged 56-71 ms over its 20 timed steps (34.3 ms in the next process on the same box) -- one-time costs that W = 5
    # steps do not always cover (first-touch of the ~25 GB the step allocates, lazily paged-in libraries).  They change
    # nothing about what is timed: exactly K steps, bracketed by barrier + synchronize.
    if ts.graph:
        args.prewarm = max(args.prewarm, 3)
    for s in range(args.prewarm):
        # (captured step: HIP events can only bracket launches of an EAGER step -- the second pre-warm step; the third
        #  one is the capture)
        probe = ts.graph and s == 1 and not args.no_kernel_events
        if probe:
            ops.KERNEL_EVENTS = []
        ts.step(*batches[s % total_steps], epoch=10)
        if probe:
            torch.cuda.synchronize()
            survey = summarise(ops.KERNEL_EVENTS, 1)
            ops.KERNEL_EVENTS = None
    for s in range(args.warmup):
        last = s == args.warmup - 1 and not args.no_kernel_events and not ts.graph
        if last:
            ops.KERNEL_EVENTS = []
        ts.step(*batches[s], epoch=10)
        if last:
            torch.cuda.synchronize()
            survey = summarise(ops.KERNEL_EVENTS, 1)
            ops.KERNEL_EVENTS = None
    # RCCL prints a version banner through C stdio on first use; push it out now so that the
    # JSON line below is the last thing this process writes to stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if not args.no_kernel_events and not ts.graph:
        ops.KERNEL_EVENTS = []
        if survey is not None:
            ops.KERNEL_EVENT_FILTER = max(survey[0].items(), key=lambda kv: kv[1][1])[0]
    counts0 = dict(D.COUNTS)
    if world > 1 or single_rank_group:
        D.EXPOSED = []                   # event-time what the main stream waits for each blocking exchange
    barrier()
    t0 = time.perf_counter()
    for s in range(args.warmup, total_steps):
        res = ts.step(*batches[s], epoch=10)
    barrier()
    elapsed = time.perf_counter() - t0
    loss = float(res["loss"])
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    roofline = None
    pmc_all = None
    tag = None
    if ts.graph and survey is not None:     # per-kernel figures of a captured run: from its eager warm-up step
        ops.KERNEL_EVENTS = [None]
    if ops.KERNEL_EVENTS:
        per, table = (summarise(ops.KERNEL_EVENTS, args.steps) if not ts.graph else survey)
        name, (fl, sec, n) = max(per.items(), key=lambda kv: kv[1][1])
        if survey is None:                   # no warm-up step: everything was bracketed in the timed region
            survey, all_steps = (per, table), args.steps
        else:
            all_steps = 1
        timed_steps = 1 if ts.graph else args.steps    # captured run: the launches of ONE eager step were timed
        all_fl = sum(v[0] for v in survey[0].values())
        all_sec = sum(v[1] for v in survey[0].values())
        all_ideal = sum(v[0] / (peak_for(k) * 1e12) for k, v in survey[0].items())    # seconds at each kernel's own peak
        peak_tf = peak_for(name)
        if args.kernel_table and rank == 0:
            rows = [{"kernel": k[0], "h_w_cin_cout_taps_halo_acc": k[1], "launches_per_step": v[2] / all_steps,
                     "ms_per_step": round(v[1] / all_steps * 1e3, 4), "tflops": round(v[0] / v[1] / 1e12, 2)}
                    for k, v in sorted(survey[1].items(), key=lambda kv: -kv[1][1])]
            json.dump(rows, open(args.kernel_table, "w"), indent=0)
        # HBM traffic of the same kernel from the PMC passes kept under profiles/ (2*FETCH_SIZE +
        # WRITE_SIZE, separate rocprofv3 --pmc runs of this bench; tools/profile_round.sh, tools/hbm_report.py)
        traffic = None
        tag = pmc_tag(args)
        pmc_all = None
        try:
            if tag is None:
                raise KeyError("no PMC capture for this workload / engine")
            pmc_all = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE.format(tag=tag))))
            traffic = round(pmc_all[name]["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(fl / sec / 1e12, 2),
                    "peak": round(peak_tf, 1), "unit": "TFLOP/s", "frac": round(fl / sec / 1e12 / peak_tf, 4),
                    "peak_note": ("fp32-equivalent ceiling of the exact-split engine: dense bf16 MFMA peak 2500 / 8 plane "
                                  "products (forward convs; the gradient kernels run six products: 416.7)")
                                 if peak_tf == PEAK_BF16_MFMA_TFLOPS / 8.0 else
                                 "fp32-equivalent ceiling of the exact-split engine with six plane products: 2500 / 6 -- the "
                                 "NOMINAL dense bf16 peak.  On operands that toggle the chip runs these launches at its power "
                                 "budget (1.9-2.0 GHz): the same launches on zero-filled tensors are 32-35 % faster "
                                 "(profiles/round3_dvfs_zero_inputs.txt; 1.23-1.41 PF on random data vs 1.67-1.86 zero-filled, "
                                 "MI355X_MICROARCH.md's own attention kernel: 1.25 / 1.48), so frac ~0.5-0.56 is the sustained "
                                 "rate of the matrix pipe on real data, not slack in the schedule"
                                 if peak_tf == PEAK_BF16_MFMA_TFLOPS / 6.0 else
                                 "fp32 MFMA peak (MI355X_MICROARCH.md)",
                    "traffic": traffic,
                    "traffic_source": (f"committed PMC pass profiles/{PMC_FILE.format(tag=tag or '')} (2*FETCH_SIZE + WRITE_SIZE "
                                       "per launch of this kernel, separate rocprofv3 --pmc runs of this bench); not "
                                       "re-measured in this run") if traffic is not None else None,
                    "launches_per_step": n // timed_steps,
                    "avg_launch_us": round(sec / n * 1e6, 2), "gflop_per_launch": round(fl / n / 1e9, 3),
                    "all_mfma_kernels": {"achieved": round(all_fl / all_sec / 1e12, 2),
                                         "frac": round(all_ideal / all_sec, 4),
                                         "frac_note": "time at each kernel's own matrix-pipe peak / measured time",
                                         "achieved_vs_fp32_mfma_peak": round(all_fl / all_sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                         "achieved_vs_fp32_mfma_peak_note": "the same fp32-class TFLOP/s against the 157.3 TFLOP/s "
                                                                            "ceiling of the fp32 MFMA instructions the reference "
                                                                            "arithmetic would run on (round 1's yardstick)",
                                         "ms_per_step": round(all_sec / all_steps * 1e3, 2),
                                         "measured_in": "last warm-up step" if all_steps == 1 else "timed steps"},
                    "dominant_kernel_measured_in": ("the eager warm-up step before the capture (HIP events cannot bracket "
                                                    "launches inside a replayed graph)") if ts.graph else "timed steps"}
    ops.KERNEL_EVENTS = None
    ops.KERNEL_EVENT_FILTER = None
    if roofline is not None and pmc_all is not None:
        # the HBM-bound kernels of the conv blocks (north star: "achieved HBM GB/s for the conv blocks"):
        # bytes = 2*FETCH_SIZE + WRITE_SIZE of the committed PMC passes over this same bench command (same workload,
        # same matrix engine), time = that capture's kernel-trace average
        rows = {}
        wanted = ("bn_bwd_kernel", "bilinear_kernel", "bilinear_bwd_kernel", "affine_add_kernel", "maskpool_kernel",
                  "maskpool_bwd_kernel", "catskip_kernel", "catskip_bwd_kernel", "pixshuf_kernel", "pixshuf_bwd_kernel",
                  "l2norm_bwd_kernel", "rownorm_kernel", "softmax_kernel", "softmax_bwd_kernel")
        tot_ms = tot_bytes = 0.0
        for k, v in pmc_all.items():
            if k.split("<")[0] in wanted:
                rows[k] = {"GBps": round(v["gbps"]), "frac_of_8TBps": round(v["gbps"] / 8000.0, 3),
                           "ms_per_step": round(v["ms_per_step"], 3)}
                tot_ms += v["ms_per_step"]
                tot_bytes += v["hbm_bytes_per_launch"] * v["launches_per_step"]
        roofline["hbm_kernels"] = {"source": f"profiles/{PMC_FILE.format(tag=tag)} (committed rocprofv3 PMC passes of this workload "
                                             "and engine, not re-measured in this run)",
                                   "peak_GBps": 8000, "ms_per_step": round(tot_ms, 3),
                                   "achieved_GBps": round(tot_bytes / (tot_ms * 1e-3) / 1e9) if tot_ms else None,
                                   "frac_of_8TBps": round(tot_bytes / (tot_ms * 1e-3) / 8e12, 3) if tot_ms else None,
                                   "kernels": rows}
        step_bytes = sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for v in pmc_all.values())
        roofline["step_hbm_traffic_GB"] = round(step_bytes / 1e9, 2)
    if roofline is not None and default_shape_for_step(args):
        # whole-step view with SURVEY 8d's normative algorithmic work per 64x2048 image and step
        # (533.5 GFLOP, 10.96 GB fp32): fraction of the fp32 matrix peak / of the 8 TB/s HBM peak
        sec_img = elapsed / (args.batch * args.steps)
        roofline["whole_step"] = {"algorithmic_TFLOPs": round(533.5e9 / sec_img / 1e12, 2),
                                  "frac_of_mfma_peak": round(533.5e9 / sec_img / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                  "mfma_peak_used": "fp32 MFMA 157.3 TFLOP/s (SURVEY 8d's ideal; same yardstick as round 1)",
                                  "algorithmic_GBps": round(10.96e9 / sec_img / 1e9, 1),
                                  "frac_of_hbm_peak": round(10.96e9 / sec_img / 8e12, 4)}

    # ranks as the process group reports them (RCCL / gloo), not as the command line claims
    n_ranks = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    collectives = None
    if n_ranks > 1 or single_rank_group:
        collectives = {k: round((D.COUNTS[k] - counts0[k]) / args.steps, 2) for k in D.COUNTS}
        collectives["total"] = round(sum(collectives.values()), 2)
        if D.EXPOSED is not None:
            ex = D.exposed_ms(D.EXPOSED)
            D.EXPOSED = None
            collectives["comm_exposed_ms"] = {k: round(v / args.steps, 3) for k, v in ex.items()}
            collectives["comm_exposed_ms"]["total"] = round(sum(ex.values()) / args.steps, 3)
            collectives["comm_exposed_note"] = ("HIP-event time per step between the issue and the completion of every BLOCKING "
                                                "exchange on the main stream (SyncBatchNorm sums, prototype bank) and of the final "
                                                "wait for the asynchronous gradient buckets: the communication nothing hides")
        collectives["note"] = ("syncbn: 43 forward + 43 backward BatchNorm layers, minus the exchanges batched with an "
                               "independent layer's; the weight-gradient stream runs under them")
    if rank == 0:
        images = args.batch * n_ranks * args.steps
        out = {
            "metric": f"range-images/sec training step, {args.height}x{args.width}x5, bs={args.batch}/GPU",
            "value": round(images / elapsed, 3), "unit": "range-images/sec",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": ("bf16 activations in HBM + bf16 MFMA operands, f32 accumulate / statistics / master weights"
                               if args.storage == "bf16" else "bf16 MFMA operands, f32 accumulate/storage"),
                      "bf16x3": "f32 via 3xbf16 exact split on the bf16 matrix pipe (the library's default engine), f32 accumulate; "
                                "6 of 9 plane products (8 of 9 in forward 3x3 / 2x2 convs whose output has fewer than 32768 "
                                "pixels, where a small BatchNorm population amplifies the difference); f32 storage everywhere.  "
                                "Parity as measured on MI355X (profiles/round3_parity_measured.json): forward / logits / "
                                "prototypes <= 1e-4 of the reference on every golden; pseudo-label maps identical; anchor "
                                "selection bit-exact on identical weights, and END TO END 2303 of 2304 anchor draws of the "
                                "reference's golden step identical (one draw lies 6.7 fp32 ulps from its bin edge and lands "
                                "on the neighbouring candidate; the strict fp32-MFMA engine, --matrix-dtype f32: 2304 of "
                                "2304); whole-network gradient error vs the reference golden: median 1.1e-2 (fp32-MFMA engine "
                                "6.1e-3; per-layer float64 check 7e-7 on both)"}[args.matrix_dtype],
            "data": "synthetic",
            "config": {"workload": f"{args.dataset} {args.height}x{args.width}x5 range image, C={args.classes}, "
                                   f"bs={args.batch}/GPU, {type(model).__name__}{'' if args.net == 'salsanext' else args.net[-2:]} fwd+bwd + prototype bank + contrast "
                                   f"loss + AdamW" + (f" (BASELINE.json configs[{2 if args.matrix_dtype == 'bf16' else 1}])"
                                                      if args.net == "salsanext" else " (SURVEY 8f N3 backbone)"),
                       "global_batch": args.batch * n_ranks, "parallelism": f"dp{n_ranks}", "final_loss": round(loss, 4),
                       "collectives_per_step": collectives},
            "roofline": roofline,
        }
        auto_graph = (args.graph == "auto" and world == 1 and not single_rank_group and args.net == "salsanext"
                      and os.environ.get("C3D_WGRAD_STREAM", "0") != "1")
        second = world == 1 and args.matrix_dtype == "bf16x3" and not args.no_second_engine
        if auto_graph or second:
            del ts, wrapped, model, res
            torch.cuda.empty_cache()
            k2 = min(args.steps, 10)

            def quick_run(dtype, wgrad_stream, graph=False, k2=k2, storage=None):
                """value / ms_per_step of k2 steps of the same workload on another engine configuration"""
                ops.set_matrix_precision(dtype, storage=storage)
                prev = os.environ.get("C3D_WGRAD_STREAM")
                os.environ["C3D_WGRAD_STREAM"] = wgrad_stream        # read when the backbone is built
                try:
                    torch.manual_seed(1)
                    m2 = (SalsaNextProto(5, args.classes, 20, 0, use_prototype=True, dataset=args.dataset) if args.net == "salsanext"
                          else RangeNetProto(layers=int(args.net[-2:]), nclasses=args.classes, dataset=args.dataset, use_prototype=True)
                          if args.net.startswith("rangenet")
                          else SqueezeSegV3Proto(nclasses=args.classes, layers=int(args.net[-2:]), dataset=args.dataset, use_prototype=True))
                    m2 = m2.to(dev).train()
                    ts2 = TrainStep(m2, args.classes, lr=1e-3, n_epochs=100, temperature=0.07, num_anchor=512, loss_w_ce_2d=1.0,
                                    loss_w_lov_2d=1.0, loss_w_contrast=0.1, feature_mean=FEATURE_MEAN, feature_std=FEATURE_STD,
                                    proto_loss=True, inputs_resident=True, graph=graph)
                    for s_ in range((3 + min(args.warmup, 2)) if graph else (min(args.warmup, 2) or 1)):   # graph: 2 eager + capture
                        ts2.step(*batches[s_ % total_steps], epoch=10)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for s_ in range(k2):
                        ts2.step(*batches[(args.warmup + s_) % total_steps], epoch=10)
                    torch.cuda.synchronize()
                    e2 = time.perf_counter() - t1
                finally:
                    if prev is None:
                        os.environ.pop("C3D_WGRAD_STREAM", None)
                    else:
                        os.environ["C3D_WGRAD_STREAM"] = prev
                del ts2, m2
                torch.cuda.empty_cache()
                return {"value": round(args.batch * k2 / e2, 3), "ms_per_step": round(e2 / k2 * 1e3, 3), "steps": k2}

        if auto_graph:
            # second pass: the same K steps (same batches), each replayed as ONE hipGraph -- the launch mode `value` is quoted on
            eager = {"value": out["value"], "ms_per_step": out["ms_per_step"], "steps": args.steps,
                     "note": "the first pass of this run: the same K steps launched one kernel at a time (python bench.py --graph off); "
                             "`roofline` holds the HIP-event kernel times of THIS pass' timed region"}
            try:
                cap = quick_run(args.matrix_dtype, "0", graph=True, k2=args.steps,
                                storage=args.storage if args.matrix_dtype == "bf16" else None)
            except Exception as e:      # noqa: BLE001 -- a bench line with the first pass' numbers beats no line
                cap = None
                out["config"]["launch"] = f"kernel by kernel (the captured pass failed: {type(e).__name__}: {e})"
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            if cap is not None:
                out["value"], out["ms_per_step"] = cap["value"], cap["ms_per_step"]
                out["launch_by_launch"] = eager
                out["config"]["launch"] = ("one hipGraph replay per step (TrainStep(graph=True): two eager steps, capture, replay; "
                                           "bit-identical to the launch-by-launch step; `launch_by_launch` = the same K steps issued "
                                           "kernel by kernel in the same process, where the host's ~20 ms per step can be the bound)")
            ops.set_matrix_precision(args.matrix_dtype, storage=args.storage if args.matrix_dtype == "bf16" else None)
        if second:
            # the same step on the fp32-MFMA engine (v_mfma_f32_32x32x2_f32), timed the same way, for comparison
            f32_run = quick_run("f32", os.environ.get("C3D_WGRAD_STREAM", "auto"))
            f32_run["dtype"] = "f32 (v_mfma_f32_32x32x2_f32 everywhere)"
            overlap_run = quick_run("bf16x3", "1")
            overlap_run["note"] = ("the headline engine with the weight-gradient chain of the backward pass on a second HIP stream "
                                   "(C3D_WGRAD_STREAM=1; what data-parallel runs use): weight gradients then execute under the "
                                   "BatchNorm-backward / elementwise kernels of the main chain.  Off by default on one GPU because the "
                                   "kernels of the two streams share the CUs and every per-kernel duration of `roofline` would inflate")
            ops.F16X2_FWD = True
            try:
                f16_run = quick_run("bf16x3", "0")
                ops.F16X2_BWD = True
                f16_run["with_input_gradients"] = dict(quick_run("bf16x3", "0"),
                                                       note="C3D_F16X2_BWD=1 on top: the multi-tap input AND weight gradients on the same arithmetic, "
                                                            "read through a per-tensor exponent (c3d_bn_bwd_apply_gmax)")
            finally:
                ops.F16X2_FWD = False
                ops.F16X2_BWD = False
            f16_run["note"] = ("EXPERIMENT, off by default (C3D_F16X2_FWD=1): the headline engine with the FORWARD convolutions over >= 32768 "
                               "pixels on two fp16 planes (x = H + L to 2^-22 |x| worst case -- a 22-bit operand, NOT the exact split --, staged times 2^6 / 2^10) and three products "
                               "instead of six -- fused nine-tap kernel conv_x3f_kernel<..., 2>, generic kernel elsewhere; gradients "
                               "unchanged.  The whole GPU parity suite passes with it; error vs float64 equals the exact split's "
                               "(profiles/round3_f16x2_probe.txt); launched kernel by kernel")
            out["engines"] = {"f32_mfma": f32_run, "bf16x3_wgrad_on_second_stream": overlap_run, "bf16x3_f16x2_forward": f16_run,
                              "note": "`value` is the bf16x3 engine's on one stream; these are the same step, launched kernel by kernel, "
                                      "on the fp32-MFMA engine (python bench.py --matrix-dtype f32 gives its full roofline object) and "
                                      "with the second stream on"}
            ops.set_matrix_precision(args.matrix_dtype)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline(args.classes, args.height, args.width) if args.net == "salsanext"
                                   else cpu_baseline_rangenet(args.classes, args.height, args.width, int(args.net[-2:]),
                                                              family=args.net[:-2]))
        else:
            out["cpu_baseline"] = None
        line = json.dumps(out)
    if world > 1 or single_rank_group:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == "__main__":
    main()
