"""Data-parallel exchange points of the hot path, one process per GPU over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Reference semantics:
  * gradient mean over ranks                 -- DistributedDataParallel, trainer.py:55-60
  * SyncBatchNorm statistics                 -- trainer.py:54
  * prototype bank: mean over ranks of each rank's l2-normalised EMA bank
                                              -- salsanext_proto.py:397-400
MI355X-first choices: ONE flat fp32 gradient buffer (29.6 MB) all-reduced in a few large chunks
on a side stream while the backward of earlier layers is still running; BatchNorm exchanges the
tiny fp64 (sum, sumsq) vectors the fused conv epilogue already produced; no per-iteration
barrier or scalar all-reduce (trainer.py:740-743 is logging only).
"""
import os

import torch
import torch.distributed as dist


def _min_world():
    # C3D_SINGLE_RANK_COLLECTIVES=1: issue every collective even in a 1-rank group, so that the
    # RCCL calls (dtypes, streams, async handles) of the exchange points can be exercised on a
    # single-GPU box (tests/test_gpu_dp.py)
    return 1 if os.environ.get("C3D_SINGLE_RANK_COLLECTIVES") else 2


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() >= _min_world()


# collectives issued by this process since start-up, by exchange point (bench.py reports the
# per-step counts as config.collectives_per_step)
COUNTS = {"syncbn": 0, "gradient_buckets": 0, "prototype_bank": 0}


# Exposed exchange time: bench.py sets EXPOSED to a list and every BLOCKING exchange on the main stream (the SyncBatchNorm
# sums, the prototype bank) brackets itself with HIP events there -- what the main stream waited for each exchange, i.e. the
# part of the communication nothing hides (`config.collectives_per_step.comm_exposed_ms`).  The gradient buckets are
# asynchronous on torch.distributed's communication stream; their exposed part is the wait in FlatGradients.finish().
EXPOSED = None


class _exposed:
    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        self.on = EXPOSED is not None and torch.cuda.is_available()
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            EXPOSED.append((self.kind, self.e0, self.e1))


def exposed_ms(events):
    """{exchange kind: total milliseconds} of a list collected in ``EXPOSED`` (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for kind, e0, e1 in events:
        out[kind] = out.get(kind, 0.0) + e0.elapsed_time(e1)
    return out


def allreduce_sum_(t):
    """In-place sum over ranks (used for the fp64 BatchNorm sums)."""
    if is_dist():
        with _exposed("syncbn"):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        COUNTS["syncbn"] += 1
    return t


def _allreduce_sum_begin(t):
    """Asynchronous form of ``allreduce_sum_``: the all-reduce is issued on torch.distributed's communication stream
    (ordered after what the current stream has queued so far) and the caller keeps queueing independent work; the
    returned handle's ``wait()`` makes the current stream wait for the result."""
    if not is_dist():
        return None
    COUNTS["syncbn"] += 1
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def _allreduce_sum_end(work):
    if work is not None:
        with _exposed("syncbn"):
            work.wait()


allreduce_sum_.begin = _allreduce_sum_begin
allreduce_sum_.end = _allreduce_sum_end


def peer_syncbn_reduce(px):
    """The SyncBatchNorm exchange hook over a ``coarse3d_amd.peer.PeerExchange`` (IPC-mapped mailboxes, one small kernel
    per exchange, csrc/peer_ops.hip) instead of one RCCL launch per vector.  Same interface as ``allreduce_sum_``
    (call = blocking on the current stream; ``begin`` / ``end`` = on a side stream, under independent kernels)."""
    def fits(t):          # (the same answer on every rank: the vectors have the same length everywhere)
        return t.numel() <= px.desc.cap_doubles and t.dtype == torch.float64 and t.is_contiguous()

    def reduce_(t):
        if not fits(t):
            return allreduce_sum_(t)
        with _exposed("syncbn"):
            px.allreduce_(t)
        COUNTS["syncbn"] += 1
        return t

    def begin(t):
        if not fits(t):
            allreduce_sum_(t)
            return None
        COUNTS["syncbn"] += 1
        return px.begin(t)

    def end(work):
        if work is not None:
            with _exposed("syncbn"):
                px.end(work)
    # one launch per BatchNorm layer and direction: fold + exchange + finish (coarse3d_amd/backbone.py takes these where a layer
    # exchanges alone; C3D_PEER_FUSED_BN=0: the three-launch path -- same bits, tests/test_gpu_dp.py)
    def bn_forward(partial, count, gamma, beta, rm, rv, momentum, eps):
        with _exposed("syncbn"):
            out = px.bn_finalize_partials(partial, count, gamma, beta, rm, rv, momentum, eps)
        COUNTS["syncbn"] += 1
        return out

    def bn_backward(partial, count, mean, invstd, gamma, dgamma, dbeta):
        with _exposed("syncbn"):
            k = px.bn_bwd_coeffs_partials(partial, count, mean, invstd, gamma, dgamma, dbeta)
        COUNTS["syncbn"] += 1
        return k

    def fits_channels(c):
        return 2 * c <= px.desc.cap_doubles
    reduce_.begin, reduce_.end, reduce_.peer = begin, end, px
    if os.environ.get("C3D_PEER_FUSED_BN", "1") != "0":
        reduce_.bn_forward, reduce_.bn_backward, reduce_.fits_channels = bn_forward, bn_backward, fits_channels
    return reduce_


def allreduce_proto_sums_(t):
    """In-place sum over ranks of the per-class prototype feature sums + counts."""
    if is_dist():
        with _exposed("prototype_bank"):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        COUNTS["prototype_bank"] += 1
    return t


def world_mean(t):
    """Mean over ranks of a tensor (prototype bank exchange)."""
    if not is_dist():
        return t
    t = t.clone().div_(dist.get_world_size())
    with _exposed("prototype_bank"):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    COUNTS["prototype_bank"] += 1
    return t


# parameter-name prefixes in the order in which Backbone.backward finishes them
BACKWARD_ORDER = ("projector", "cls_head", "upBlock4", "upBlock3", "upBlock2", "upBlock1", "resBlock5", "resBlock4",
                  "resBlock3", "resBlock2", "resBlock1", "downCntx3", "downCntx2", "downCntx",
                  # RangeNetBackbone / SqueezeSegBackbone.backward (coarse3d_amd/rangenet.py, squeezeseg.py)
                  "head", "head5", "decoder.dec1", "decoder.dec2", "decoder.dec3", "decoder.dec4", "decoder.dec5",
                  "backbone.enc5", "backbone.enc4", "backbone.enc3", "backbone.enc2", "backbone.enc1", "backbone.conv1")


def _block_of(name):
    """Parameter name -> tag of the backward block that finishes it."""
    parts = name.split(".")
    if parts[0] in ("backbone", "decoder"):
        return "backbone.conv1" if parts[1] == "bn1" else f"{parts[0]}.{parts[1]}"
    return parts[0]


class FlatGradients:
    """All trainable gradients as views of one flat buffer laid out in BACKWARD completion
    order, so that finished prefixes can be all-reduced while backward continues."""

    def __init__(self, named_params, device=None):
        named = list(named_params)
        rank = {b: i for i, b in enumerate(BACKWARD_ORDER)}
        named.sort(key=lambda kv: rank.get(_block_of(kv[0]), len(rank)))     # stable
        self.names = [n for n, _ in named]
        total = sum(p.numel() for _, p in named)
        dev = device if device is not None else named[0][1].device
        # one element behind the gradients: the health word of the step (0 = fine).  It travels with the LAST gradient bucket,
        # so every rank learns of any rank's failed SyncBatchNorm exchange from an all-reduce the step makes anyway -- no extra
        # collective (DataParallel.watch_status / check_status).  ``flat`` -- what the optimiser and everybody else sees -- is
        # the gradients alone.
        self._storage = torch.zeros(total + 1, device=dev, dtype=torch.float32)
        self.flat = self._storage[:total]
        self.status = self._storage[total:]
        self.status_src = None        # callable(dst: fp32[1]) writing this rank's health word on the current stream, or None
        self.views, off = {}, 0
        self.block_end = {}           # block tag -> end offset of its parameters in the flat buffer
        for n, p in named:
            self.views[n] = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.block_end[_block_of(n)] = off
        self._sent = 0
        self._works = []
        self._seen = []
        self.collectives = 0

    # ---- bucketed, overlapped mean over ranks
    def begin(self):
        self._sent = 0
        self._works = []
        self._seen = []
        self.collectives = 0

    def block_done(self, tag, min_bytes=4 << 20):
        """Gradients of block ``tag`` are final: all-reduce the finished prefix once it is large
        enough (a few big messages: xGMI is latency-, not bandwidth-limited at 29.6 MB total)."""
        end = self.block_end.get(tag)
        if end is None:
            return
        # the prefix [0, end) is sent as final: that only holds if blocks report in the order the
        # buffer was laid out in (BACKWARD_ORDER must match the hook order of Backbone.backward)
        if self._seen and end <= self.block_end[self._seen[-1]]:
            raise RuntimeError(f"gradient block {tag!r} reported after {self._seen[-1]!r}: BACKWARD_ORDER in "
                               "coarse3d_amd/dist.py no longer matches the backward pass")
        self._seen.append(tag)
        if not is_dist():
            return
        if (end - self._sent) * 4 >= min_bytes or end == self.flat.numel():
            self._launch(end)

    def _launch(self, end):
        if end <= self._sent:
            return
        seg = self.flat[self._sent:end]
        if end == self.flat.numel():
            if self.status_src is not None:
                self.status_src(self.status)
            seg = self._storage[self._sent:end + 1]       # ... + the health word: mean over ranks, > 0 = some rank failed
        seg.div_(dist.get_world_size())
        self._works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        self.collectives += 1
        COUNTS["gradient_buckets"] += 1
        self._sent = end

    def finish(self):
        missing = [t for t in self.block_end if t not in self._seen]
        if self._seen and missing:
            raise RuntimeError(f"backward finished without reporting gradient blocks {missing}")
        if is_dist():
            self._launch(self.flat.numel())
            with _exposed("gradient_wait"):
                for w in self._works:
                    w.wait()
        self._works = []

    def all_reduce_mean(self, n_chunks=4):
        """Non-overlapped variant (kept for tests): mean over ranks in n_chunks messages."""
        if not is_dist():
            return
        world = dist.get_world_size()
        chunk = (self.flat.numel() + n_chunks - 1) // n_chunks
        works = []
        for i in range(n_chunks):
            seg = self.flat[i * chunk:(i + 1) * chunk]
            if seg.numel():
                seg.div_(world)
                works.append(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()


class DataParallel(torch.nn.Module):
    """Minimal DDP stand-in exposing ``.module`` (what the reference trainer reads,
    trainer.py:676-678) and wiring the three exchange points into SalsaNextProto's hooks.
    Gradient buckets are all-reduced on torch.distributed's communication stream as soon as
    the explicit backward has finished the corresponding blocks (overlap with the rest of
    backward); ``finish_gradients()`` waits for them and re-binds ``param.grad`` to the
    reduced flat views."""

    STATUS_LAG = 2      # a step's health word is checked this many steps later (its copy has long arrived: no stall)

    def __init__(self, module, sync_bn=True, proto_sync="bank_mean", syncbn_exchange="auto", peer_timeout_s=None):
        """proto_sync: "bank_mean" = reference semantics (mean over ranks of each rank's updated,
        l2-normalised bank, salsanext_proto.py:397-400); "sums" = all-reduce the per-class masked
        feature sums and counts and apply ONE momentum update with the global statistics (what a
        single process on the global batch would compute, up to the per-rank Sinkhorn).
        syncbn_exchange: how the 43 + 43 fp64 statistics vectors of a step travel (C3D_SYNCBN_EXCHANGE overrides).
        "collective" = one torch.distributed all-reduce each (RCCL / gloo).  "peer" = IPC-mapped mailboxes, one small
        kernel per exchange (coarse3d_amd/peer.py; one node, <= 8 ranks); raises if they cannot be set up.  "try_peer" =
        "peer" when every rank could set its mailboxes up and the self-test exchanges held, else "collective" -- decided
        together, the ranks never split (bench.py runs this one, behind its own consensus check and fallback).  "auto"
        (default) = "try_peer" ONLY in the placements whose kernels have run on hardware -- a 1-rank group, or every rank on
        one device (the fence-free form) -- and "collective" for ranks on different GPUs: the fenced cross-device form of
        the exchange kernels has not met an xGMI link yet, and a library default must not be the first run.
        peer_timeout_s: how long an exchange waits for a peer (default: C3D_PEER_TIMEOUT_S or 600 s -- the order of the
        process-group timeout of the collective it replaces; coarse3d_amd/peer.py).
        A failed exchange does not go unnoticed: its results are NaN, and the rank's status word travels with the last
        gradient bucket of every step, so ``finish_gradients()`` / ``TrainStep`` raise on EVERY rank, at the same step,
        ``STATUS_LAG`` steps later at most (``check_status(final=True)`` / ``TrainStep.flush()``: at once)."""
        super().__init__()
        if proto_sync not in ("bank_mean", "sums"):
            raise ValueError(f"proto_sync must be 'bank_mean' or 'sums', got {proto_sync!r}")
        syncbn_exchange = os.environ.get("C3D_SYNCBN_EXCHANGE", syncbn_exchange)
        if syncbn_exchange not in ("auto", "peer", "try_peer", "collective"):
            raise ValueError(f"syncbn_exchange must be 'auto', 'peer', 'try_peer' or 'collective', got {syncbn_exchange!r}")
        self.module = module
        world = dist.get_world_size() if is_dist() else 1
        module._world = world if sync_bn else 1
        module._bn_reduce = allreduce_sum_ if (sync_bn and is_dist()) else None
        self.peer = None
        on_gpu = next(module.parameters()).is_cuda
        if module._bn_reduce is not None and syncbn_exchange != "collective" and (on_gpu or syncbn_exchange == "peer"):
            from .peer import PeerExchange
            try:
                # collective: raises on every rank or on none
                self.peer = PeerExchange(timeout_s=peer_timeout_s, only_one_device=(syncbn_exchange == "auto"))
                module._bn_reduce = peer_syncbn_reduce(self.peer)
            except RuntimeError:
                if syncbn_exchange == "peer":
                    raise
        module._proto_mean = world_mean if (is_dist() and proto_sync == "bank_mean") else None
        module._proto_sums_reduce = allreduce_proto_sums_ if (is_dist() and proto_sync == "sums") else None
        self.flat = FlatGradients(module._trainable())
        if self.peer is not None:
            self.flat.status_src = self.peer.status_to
        self._status_ring = []            # (step, event, slot) of the health words on their way to the host
        self._status_host = None
        self._steps = 0
        module._flat_grads = self.flat.views
        module._block_done = self.flat.block_done
        if is_dist():                      # identical initial weights on every rank
            for p in module.parameters():
                dist.broadcast(p.data, 0)
            for b in module.buffers():
                dist.broadcast(b.data, 0)

    def close(self):
        """Collective: release the peer mailboxes (every rank, same point); the module goes back to rank-local statistics
        hooks only if wrapped again.  Dropping the wrapper without it is fine too (each rank then frees on its own)."""
        if self.peer is not None:
            self.peer.close(collective=True)
            self.peer = None

    def forward(self, *a, **k):
        self.flat.begin()
        return self.module(*a, **k)

    def watch_status(self):
        """Queue an asynchronous read-back of this step's health word (FlatGradients.status after the last bucket: the mean
        over ranks of the ranks' peer-exchange status words) and check the one of ``STATUS_LAG`` steps ago -- the same step on
        every rank, so the ranks raise together instead of one of them leaving the others in a collective.  Called by
        ``finish_gradients()``; a captured step cannot (host code does not replay): ``TrainStep`` calls it after each replay."""
        if self.flat.status_src is None or not self.flat.status.is_cuda:
            return
        if self._status_host is None:
            self._status_host = torch.zeros(8, dtype=torch.float32).pin_memory()
        self._steps += 1
        slot = self._steps % 8
        self._status_host[slot:slot + 1].copy_(self.flat.status, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._status_ring.append((self._steps, ev, slot))
        self.check_status()

    def check_status(self, final=False):
        """Raises RuntimeError if the health word of a step says that some rank's SyncBatchNorm exchange through peer memory
        failed (timed out).  ``final``: every queued step (synchronises) -- before a checkpoint, at the end of an epoch."""
        while self._status_ring and (final or self._status_ring[0][0] <= self._steps - self.STATUS_LAG):
            step, ev, slot = self._status_ring.pop(0)
            ev.synchronize()
            if float(self._status_host[slot]) != 0.0:      # (NaN included)
                self._status_ring = []
                raise RuntimeError(
                    f"coarse3d_amd.dist.DataParallel: a SyncBatchNorm exchange through peer memory failed on at least one rank in "
                    f"data-parallel step {step} (a peer did not arrive within the exchange timeout: a rank died, stalled longer "
                    f"than C3D_PEER_TIMEOUT_S, or the ranks' call sequences diverged).  The statistics of that step and "
                    f"everything computed since are invalid (NaN by construction) on every rank: restore the last checkpoint; "
                    f"C3D_SYNCBN_EXCHANGE=collective takes the mailboxes out of the picture")
        if final and self.peer is not None and self.flat.status_src is not None and self.peer.failed():
            raise RuntimeError("coarse3d_amd.dist.DataParallel: this rank's peer-memory exchange reports a timeout (status word set)")

    def finish_gradients(self):
        self.flat.finish()
        if not (self.flat.flat.is_cuda and torch.cuda.is_current_stream_capturing()):
            self.watch_status()
        live = getattr(self.module, "_grads_live", True)
        for n, p in self.module._trainable():      # autograd may have cloned the views
            # (no embedding branch in this step -- contrast warm-up --: the projector has no gradient, on any rank)
            p.grad = self.flat.views[n] if (live or not n.startswith("projector.")) else None
