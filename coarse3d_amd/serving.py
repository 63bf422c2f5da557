"""Low-latency inference: the eval-mode forward of SalsaNextProto captured in a hipGraph.

A single 64x2048 scan takes ~3 ms of GPU time in ~250 kernel launches; issued from Python the
forward is launch-bound (the host needs longer to enqueue the kernels than the GPU needs to run
them).  Every kernel of the C ABI is launched on the caller's stream with caller-provided
buffers and the library never allocates or synchronises, so the whole forward is capturable:
``GraphedInference`` records it once per input shape and replays it with one launch."""
import contextlib

import torch

from . import ops


@contextlib.contextmanager
def _own_packs(model, packs):
    saved, model._packs = model._packs, packs
    try:
        yield
    finally:
        model._packs = saved


class GraphedInference:
    """``out = GraphedInference(model)(x)`` -- x [B,5,H,W] float32 on the GPU.  Returns the dict of
    the eval forward (``pred_2d`` always, ``feat_2d`` if return_feat); the tensors are the graph's
    static outputs and are overwritten by the next call (clone them to keep them)."""

    def __init__(self, model, return_feat=False, warmup=2):
        if model.training:
            raise ValueError("GraphedInference captures the eval-mode forward: call model.eval() first")
        if warmup < 2:
            # the first eager pass records which weight repacks the plan needs, the second builds the
            # batched-repack table (a pageable H2D copy): neither may happen inside the capture
            raise ValueError("GraphedInference needs warmup >= 2")
        self.model, self.return_feat, self.warmup = model, return_feat, warmup
        self._graphs = {}
        # The captured graph bakes in raw device addresses of the weight-pack table and buffers.
        # They live in a PackCache owned by THIS object, which nothing else adds entries to: a
        # training step or a second GraphedInference on the same model (different pack set) can then
        # never free or move what a captured graph still points at.
        self._packs = ops.PackCache()

    def _storage_signature(self):
        """Addresses of everything the captured kernels read by raw pointer besides the pack table: parameters
        (conv biases, BatchNorm affine, the repack sources) and buffers (running statistics).  FlatAdamW rebinds every
        ``p.data`` into its flat buffer, ``model.float()/.to()`` and ``load_state_dict(assign=True)`` move storage too."""
        return tuple(t.data_ptr() for t in list(self.model.parameters()) + list(self.model.buffers()))

    def _capture(self, x):
        static_x = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with _own_packs(self.model, self._packs):
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):             # lazy initialisation (weight packs, ...) outside the graph
                    self.model(static_x, return_feat=self.return_feat)
            torch.cuda.current_stream().wait_stream(side)
            gen = self._packs.generation
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                out = self.model(static_x, return_feat=self.return_feat)
            if self._packs.generation != gen:
                raise RuntimeError("the weight-pack table was rebuilt during graph capture")
        return g, static_x, out, self._storage_signature()

    def __call__(self, x):
        if self.model.training:
            raise ValueError("GraphedInference replays the eval-mode forward: call model.eval() first")
        sig = self._storage_signature()
        if self._graphs and any(v[3] != sig for v in self._graphs.values()):
            # parameter / buffer storage moved since the capture (an optimiser flattened the parameters, the model was
            # cast or re-loaded): every captured graph -- and this object's weight packs -- point at stale or freed
            # memory.  Drop them all and re-capture from the live tensors.
            self._graphs.clear()
            self._packs = ops.PackCache()
        key = tuple(x.shape)
        if key not in self._graphs:
            self._graphs[key] = self._capture(x)
        g, static_x, out, _ = self._graphs[key]
        static_x.copy_(x)
        g.replay()
        return out
