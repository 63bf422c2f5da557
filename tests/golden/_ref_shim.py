"""Import shim for the upstream COARSE3D reference (THIS container only).

The reference (/root/reference) is pure Python/PyTorch with hard-coded ``.cuda()``
calls, a debug leftover that overwrites the forward inputs
(pc_processor/models/salsanext_proto.py:414-421) and package ``__init__`` files that
pull in heavy optional dependencies.  This module makes the hot-path classes importable
on CPU so golden vectors can be generated.  It never ships to the GPU box: only the
vectors it produces (tests/golden/*.npz) do.
"""
import ast
import importlib
import inspect
import sys
import textwrap
import types

import torch

REF_ROOT = "/root/reference"


def load_reference():
    """Return a namespace with the reference classes / functions of the hot path."""
    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        timm_layers = types.ModuleType("timm.models.layers")
        timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
        timm.models = timm_models
        timm_models.layers = timm_layers
        sys.modules["timm"] = timm
        sys.modules["timm.models"] = timm_models
        sys.modules["timm.models.layers"] = timm_layers

    # bare package objects: skip the heavy __init__ files
    for name, sub in (("pc_processor", "pc_processor"),
                      ("pc_processor.models", "pc_processor/models"),
                      ("pc_processor.loss", "pc_processor/loss"),
                      ("pc_processor.metrics", "pc_processor/metrics")):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [f"{REF_ROOT}/{sub}"]
            sys.modules[name] = pkg

    # .cuda() -> identity on CPU
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    rng_state = torch.random.get_rng_state()
    m_salsa = importlib.import_module("pc_processor.models.salsanext_proto")
    m_sink = importlib.import_module("pc_processor.models.sinkhorn")
    m_proj = importlib.import_module("pc_processor.models.projector")
    m_contrast = importlib.import_module("pc_processor.loss.contrast_pixel_loss")  # reseeds RNG
    m_focal = importlib.import_module("pc_processor.loss.focal_softmax")
    m_lovasz = importlib.import_module("pc_processor.loss.lovasz_softmax")
    m_iou = importlib.import_module("pc_processor.metrics.iou_eval")
    torch.random.set_rng_state(rng_state)

    # strip the debug lines that overwrite x/label/eval_mask
    cls = m_salsa.SalsaNextProto
    if not getattr(cls, "_golden_patched", False):
        src = textwrap.dedent(inspect.getsource(cls.forward)).splitlines()
        start = next(i for i, l in enumerate(src) if l.strip() == "bs = 1")
        end = next(i for i, l in enumerate(src) if "eval_mask = torch.ones((bs, h, w)).cuda()" in l)
        src = src[:start] + src[end + 1:]
        ns = {}
        exec(compile("\n".join(src), "<patched SalsaNextProto.forward>", "exec"), m_salsa.__dict__, ns)
        cls.forward = ns["forward"]
        cls._golden_patched = True

    # entropy_based_selection is a Trainer method; lift it out by AST
    tr_src = open(f"{REF_ROOT}/tasks/weak_segmentation/trainer.py").read()
    tree = ast.parse(tr_src)
    fn_node = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "entropy_based_selection":
            fn_node = node
    mod = ast.Module(body=[fn_node], type_ignores=[])
    ns = {"torch": torch}
    exec(compile(mod, "<entropy_based_selection>", "exec"), ns)

    out = types.SimpleNamespace(
        SalsaNextProto=m_salsa.SalsaNextProto,
        ResContextBlock=m_salsa.ResContextBlock,
        ResBlock=m_salsa.ResBlock,
        UpBlock=m_salsa.UpBlock,
        distributed_sinkhorn=m_sink.distributed_sinkhorn,
        ProjectionV1=m_proj.ProjectionV1,
        ContrastMEMLoss=m_contrast.ContrastMEMLoss,
        FocalSoftmaxLoss=m_focal.FocalSoftmaxLoss,
        Lovasz_softmax=m_lovasz.Lovasz_softmax,
        entropy_based_selection=ns["entropy_based_selection"],
        IOUEval=m_iou.IOUEval,
    )
    return out
