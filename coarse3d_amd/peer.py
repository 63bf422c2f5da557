"""SyncBatchNorm statistics exchange through IPC-mapped peer memory (csrc/peer_ops.hip).

The reference wraps its model with ``torch.nn.SyncBatchNorm.convert_sync_batchnorm`` (tasks/weak_segmentation/trainer.py:54):
every BatchNorm layer all-reduces its per-channel sums in forward and backward -- 43 + 43 vectors of at most 704 x 2 fp64
per training step, each a RCCL launch of its own through ``torch.distributed`` (17 us of launch cost in a 1-rank group before
any xGMI latency, and they block the main stream).  ``PeerExchange`` sets up, ONCE, one mailbox per rank in device memory,
mapped into every other rank of the node (hipIpc handles travel over the existing process group); an exchange is then one
small kernel that writes the rank's vector into every peer's mailbox, publishes a sequence number, waits (bounded) for the
peers' and sums the slots in rank order -- the same bits on every rank.  The kernel is capturable (its sequence counter
lives in device memory), so a captured data-parallel step keeps only the gradient buckets and the bank on RCCL.

One process per GPU, all ranks on one node (IPC), at most ``C3D_PEER_MAX_RANKS`` = 8 of them.  Any transport works for the
control plane (nccl, gloo); two ranks may share one device (tests/test_gpu_dp.py runs exactly that on the one-GPU box)."""
import ctypes as C
import os
import socket

import torch
import torch.distributed as dist

from . import _lib as L

SELFTEST_EXCHANGES = 5 + 96   # what PeerExchange(selftest=True) adds to the exchange counter
CAP_DOUBLES = 8192          # slot capacity: SalsaNext's largest grouped exchange is 2 x (704 + 128) doubles, SqueezeSegV3's 2 x 2304


class PeerDesc(C.Structure):
    _fields_ = [("mailbox", C.c_void_p * 8), ("rank", C.c_int32), ("world", C.c_int32), ("cap_doubles", C.c_int32),
                ("timeout_s", C.c_float), ("one_device", C.c_int32), ("reserved", C.c_int32)]


def default_timeout_s():
    """How long an exchange waits for a peer before it gives up (status word set, NaN results, every later exchange a no-op).
    The collective it replaces waits for the process group's timeout (RCCL / NCCL: 10-30 minutes); ranks legitimately
    drift by more than seconds -- a checkpoint written by rank 0, a data-loader stall, first-touch compilation -- and this
    package's trainer has no per-step barrier (coarse3d_amd/trainer.py).  Hence minutes, not seconds: ``C3D_PEER_TIMEOUT_S``
    (default 600).  After rank-local work longer than that, barrier before the next step."""
    return float(os.environ.get("C3D_PEER_TIMEOUT_S", "600"))


class PeerExchange:
    def __init__(self, group=None, timeout_s=None, cap_doubles=CAP_DOUBLES, selftest=True, only_one_device=False):
        """Collective: every rank of ``group`` (default: the world) must construct it at the same point.  Raises
        RuntimeError -- on EVERY rank -- if any rank could not allocate, share or map a mailbox, if the ranks are not on one
        host, or if the self-test exchanges (``selftest``: five of known vectors, then 96 queued back to back with one rank
        late at a time, as in a training step) did not return the right sums on every rank within two seconds each.
        ``only_one_device``: also raise (everywhere) when the ranks sit on different devices -- the placement whose fenced
        form of the kernels has not run over xGMI yet; a library default takes the collectives there (dist.DataParallel)."""
        if not torch.cuda.is_available():
            raise RuntimeError("PeerExchange needs a GPU")
        self.group = group
        have_group = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if have_group else 0
        self.world = dist.get_world_size(group) if have_group else 1
        if self.world > 8:
            raise RuntimeError(f"PeerExchange handles at most 8 ranks (one node), got {self.world}")
        lib = L.lib()
        self._own = None
        self._mapped = []
        if timeout_s is None:
            timeout_s = default_timeout_s()
        self.desc = PeerDesc()
        self.desc.rank, self.desc.world, self.desc.cap_doubles, self.desc.timeout_s = self.rank, self.world, cap_doubles, timeout_s
        self.stream = torch.cuda.Stream()       # for the asynchronous form (begin / end)
        # One exchange of a rank at a time: every launch shares the sequence counter, the parity slots, the scratch and the
        # ticket word.  begin() puts an exchange on the side stream; until its end() nothing else of this object may be
        # launched (it would run on ANOTHER stream, unordered against the one in flight) -- enforced below, not assumed.
        self._in_flight = False
        self._scratch = self._ticket = None
        err = None
        handle = (C.c_ubyte * 64)()
        try:
            nbytes = (lib.c3d_peer_mailbox_bytes(cap_doubles) + 255) // 256 * 256
            ptr = C.c_void_p()
            # behind the mailbox, in the same fine-grained allocation (zeroed): the sums scratch (4 x cap doubles) and the ticket
            # word of the fused BatchNorm launches -- their blocks talk through write-through stores, like the exchange itself
            L.check(lib.c3d_peer_alloc(nbytes + 4 * cap_doubles * 8 + 256, C.byref(ptr), handle), "c3d_peer_alloc")
            self._own = ptr.value
            self._scratch = self._own + nbytes
            self._ticket = self._scratch + 4 * cap_doubles * 8
        except Exception as e:      # noqa: BLE001 -- the other ranks must learn about it below
            err = f"rank {self.rank}: {e}"
        props = torch.cuda.get_device_properties(torch.cuda.current_device())
        # which physical device: its UUID (or PCI address).  A device INDEX says nothing when every rank sees only its own GPU, so
        # without either the ranks count as being on different devices (the conservative, fenced form of the kernels)
        dev_id = ((getattr(props, "uuid", None) and str(props.uuid))
                  or (getattr(props, "pci_bus_id", None) is not None
                      and f"pci {getattr(props, 'pci_domain_id', 0)}:{props.pci_bus_id}:{getattr(props, 'pci_device_id', 0)}")
                  or f"unknown device of rank {self.rank}")
        infos = self._gather((socket.gethostname(), bytes(handle), err, str(dev_id)))
        errs = [i[2] for i in infos if i[2]]
        if not errs and len({i[0] for i in infos}) != 1:
            errs = [f"the ranks run on different hosts ({sorted({i[0] for i in infos})}): IPC needs one node"]
        # every rank on one device (the one-GPU test box; world == 1): no system-scope fences around the write-through payload
        self.desc.one_device = int(len({i[3] for i in infos}) == 1)
        same_device = bool(self.desc.one_device)
        if os.environ.get("C3D_PEER_FORCE_FENCES") == "1":
            # test hook: ranks that share one device take the multi-device form of the kernels (system-scope release / acquire
            # fences around the payload) -- the code path a node with several GPUs runs, exercised on a one-GPU box
            self.desc.one_device = 0
        if not errs and only_one_device and not same_device and self.world > 1:
            errs = ["the ranks run on different devices and the caller asked for the one-device form only"]
        if not errs:
            try:
                for r, (_, h, _, _) in enumerate(infos):
                    if r == self.rank:
                        self.desc.mailbox[r] = self._own
                        continue
                    p = C.c_void_p()
                    L.check(lib.c3d_peer_open((C.c_ubyte * 64).from_buffer_copy(h), C.byref(p)), "c3d_peer_open")
                    self._mapped.append(p.value)
                    self.desc.mailbox[r] = p.value
            except Exception as e:      # noqa: BLE001
                err = f"rank {self.rank}: {e}"
            errs = [e for e in self._gather(err) if e]       # (also the barrier: every mailbox is mapped before its first use)
        if not errs and selftest:
            # First use on this set of devices: a few exchanges of known vectors with a short timeout, checked on every rank.
            # A transport that does not deliver (or delivers stale data) shows here, in two seconds and with an exception
            # on EVERY rank, instead of as a hung or silently wrong training step -- the cross-GPU path has not run on
            # hardware in this repository's tests (one-GPU boxes), only the two-processes-one-device path has.
            err = self._selftest()
            errs = [e for e in self._gather(err) if e]
        if errs:
            self.close()
            raise RuntimeError("PeerExchange setup failed: " + "; ".join(errs))

    def _selftest(self):
        keep = self.desc.timeout_s
        self.desc.timeout_s = 2.0
        try:
            for it, n in enumerate((257, 1, 4096, 64, self.desc.cap_doubles)):
                t = torch.arange(n, dtype=torch.float64, device="cuda") * (self.rank + 1) + 0.25 * it
                self.allreduce_(t)
                want = torch.arange(n, dtype=torch.float64) * sum(r + 1 for r in range(self.world)) + 0.25 * it * self.world
                if not torch.equal(t.cpu(), want):
                    return f"rank {self.rank}: self-test exchange {it} ({n} values) returned a wrong sum"
            # the pattern of a training step: many exchanges queued back to back with no host synchronisation in between
            # (both parities reused ~50 times), one rank arriving late at some of them, sizes of the BatchNorm layers
            sizes = (64, 128, 256, 512, 1408, 64, 64, 1664)
            ts = []
            for it in range(96):
                n = sizes[it % len(sizes)]
                t = torch.full((n,), float(it), dtype=torch.float64, device="cuda") + self.rank
                if it % 16 == self.rank % 16:
                    torch.cuda._sleep(200_000)          # ~0.1 ms: this rank is the late one
                self.allreduce_(t)
                ts.append((it, t))
            base = sum(range(self.world))
            for it, t in ts:
                if not bool((t == float(it) * self.world + base).all()):
                    return f"rank {self.rank}: self-test exchange {it} of the back-to-back series returned a wrong sum"
            self.check()
            return None
        except Exception as e:      # noqa: BLE001 -- every rank must reach the gather below
            return f"rank {self.rank}: {e}"
        finally:
            self.desc.timeout_s = keep

    def _gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj, group=self.group)
        return out

    def allreduce_(self, t, stream=None):
        """In-place sum over ranks of a contiguous fp64 CUDA tensor (every rank: same call sequence, same sizes)."""
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("PeerExchange.allreduce_: contiguous fp64 CUDA tensor expected")
        if t.numel() > self.desc.cap_doubles:
            raise ValueError(f"PeerExchange.allreduce_: {t.numel()} values exceed the mailbox slot ({self.desc.cap_doubles})")
        if stream is None:
            self._require_idle("allreduce_")
        s = stream if stream is not None else torch.cuda.current_stream()
        L.check(L.lib().c3d_peer_allreduce_f64(C.byref(self.desc), t.data_ptr(), t.numel(), C.c_void_p(s.cuda_stream)),
                "c3d_peer_allreduce_f64")
        return t

    def _require_idle(self, what):
        if self._in_flight:
            raise RuntimeError(f"PeerExchange.{what}: an exchange started with begin() is still in flight on the side stream; the "
                               "exchanges of a rank share one sequence counter, parity slots and ticket word and must not overlap "
                               "-- call end() first (and never mix begin/end with the blocking calls inside one window)")

    def bn_finalize_partials(self, partial, count, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5):
        """SyncBatchNorm forward statistics of one layer in ONE launch on the current stream: fold the partials [C, 2, n],
        exchange the fp64 sums, finish -> (scale, shift, mean, invstd) and the running statistics.  ``count``: elements per
        channel over all ranks.  One exchange in every rank's call sequence (csrc/peer_ops.hip)."""
        c = gamma.shape[0]
        if 2 * c > self.desc.cap_doubles:
            raise ValueError(f"PeerExchange.bn_finalize_partials: {c} channels exceed the mailbox slot")
        self._require_idle("bn_finalize_partials")
        buf = torch.empty(4, c, device=gamma.device, dtype=torch.float32)
        s = torch.cuda.current_stream()
        L.check(L.lib().c3d_peer_bn_finalize_partials(
            C.byref(self.desc), partial.data_ptr(), partial.shape[2], float(count), gamma.data_ptr(), beta.data_ptr(),
            running_mean.data_ptr() if running_mean is not None else None, running_var.data_ptr() if running_var is not None else None,
            momentum, eps, c, buf[0].data_ptr(), buf[1].data_ptr(), buf[2].data_ptr(), buf[3].data_ptr(), C.c_void_p(self._scratch),
            C.c_void_p(self._ticket), C.c_void_p(s.cuda_stream)), "c3d_peer_bn_finalize_partials")
        return buf[0], buf[1], buf[2], buf[3]

    def bn_bwd_coeffs_partials(self, partial, count, mean, invstd, gamma, dgamma, dbeta):
        """... and the backward sums (sum dy, sum dy * a): -> the three input-gradient coefficients [3, C] from the global sums,
        dgamma / dbeta from this rank's (as torch.nn.SyncBatchNorm: parameter gradients are averaged with the others)."""
        c = gamma.shape[0]
        if 2 * c > self.desc.cap_doubles:
            raise ValueError(f"PeerExchange.bn_bwd_coeffs_partials: {c} channels exceed the mailbox slot")
        self._require_idle("bn_bwd_coeffs_partials")
        k = torch.empty(3, c, device=gamma.device, dtype=torch.float32)
        s = torch.cuda.current_stream()
        L.check(L.lib().c3d_peer_bn_bwd_coeffs_partials(
            C.byref(self.desc), partial.data_ptr(), partial.shape[2], float(count), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
            c, k[0].data_ptr(), k[1].data_ptr(), k[2].data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), C.c_void_p(self._scratch),
            C.c_void_p(self._ticket), C.c_void_p(s.cuda_stream)), "c3d_peer_bn_bwd_coeffs_partials")
        return k

    def begin(self, t):
        """Asynchronous form: the exchange runs on this object's side stream, ordered after what the current stream has
        queued; ``end`` makes the current stream wait for it.  Independent kernels queued in between run under the wait."""
        self._require_idle("begin")
        self.stream.wait_stream(torch.cuda.current_stream())
        t.record_stream(self.stream)
        self.allreduce_(t, self.stream)
        self._in_flight = True
        return self.stream

    def end(self, stream):
        torch.cuda.current_stream().wait_stream(stream)
        self._in_flight = False

    def status_to(self, dst, stream=None):
        """This rank's status word as a device float (0 / 1) into ``dst`` (one fp32 element) on the current stream -- no host
        synchronisation; coarse3d_amd.dist.DataParallel appends it to its last gradient bucket."""
        s = stream if stream is not None else torch.cuda.current_stream()
        L.check(L.lib().c3d_peer_status_to(C.byref(self.desc), dst.data_ptr(), C.c_void_p(s.cuda_stream)), "c3d_peer_status_to")

    def failed(self):
        """True if an exchange of this rank gave up waiting for a peer (synchronises; does not raise)."""
        st = C.c_int32(0)
        torch.cuda.synchronize()
        L.check(L.lib().c3d_peer_status(C.byref(self.desc), C.byref(st), None), "c3d_peer_status")
        return bool(st.value)

    def check(self):
        """Host check (synchronises): raises if an exchange gave up waiting for a peer -- its result was not a sum, and
        everything computed from it since is wrong.  Returns the number of exchanges made."""
        st, calls = C.c_int32(0), C.c_int64(0)
        torch.cuda.synchronize()
        L.check(L.lib().c3d_peer_status(C.byref(self.desc), C.byref(st), C.byref(calls)), "c3d_peer_status")
        if st.value:
            raise RuntimeError(f"PeerExchange: rank {self.rank} timed out waiting for a peer in one of its {calls.value} exchanges "
                               "(a rank died, or the ranks' call sequences diverged); results since then are invalid")
        return calls.value

    def close(self, collective=False):
        """Unmap the peers' mailboxes, free the own one.  ``collective`` (every rank calls it at the same point, the process
        group is alive): a barrier in between, so that no rank frees a mailbox another rank still has mapped."""
        lib = L.lib()
        torch.cuda.synchronize()
        for p in self._mapped:
            lib.c3d_peer_close(C.c_void_p(p))
        self._mapped = []
        if collective and self.world > 1 and dist.is_available() and dist.is_initialized():
            dist.barrier(group=self.group)
        if self._own is not None:
            lib.c3d_peer_free(C.c_void_p(self._own))
            self._own = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass
