"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares, the
module mirror has the reference's state_dict surface, host-side helpers behave."""
import ctypes
import os

import numpy as np
import pytest
import torch

import weights as W
from oracle import coarse3d_oracle as oc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from coarse3d_amd import _lib as L
    protos = L.prototypes()
    assert len(protos) >= 40
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in include/coarse3d_hip.h but not exported"
    assert L.lib().c3d_version() >= 100
    assert L.lib().c3d_conv_num_mtiles(8, 64, 2048) == 8 * 8 * 64


def test_ctypes_struct_mirrors_match_the_c_layouts():
    """The ctypes Structures of coarse3d_amd/_lib.py must have exactly the size the C compiler
    gives the descriptor structs of include/coarse3d_hip.h (a silent mismatch would shift every
    field after the first divergence)."""
    from coarse3d_amd import _lib as L
    out = (ctypes.c_int32 * 4)()
    assert L.lib().c3d_abi_sizes(out) == 0
    assert list(out) == [ctypes.sizeof(L.Src), ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.WgradDesc),
                         ctypes.sizeof(L.PackEntry)]
    # a refused call reports why, without touching the GPU
    d = L.ConvDesc()
    d.nsrc = 7
    assert L.lib().c3d_conv_forward(ctypes.byref(d), None) != 0
    assert b"nsrc" in L.lib().c3d_last_error()
    d.nsrc = 1                                  # null pointers are refused on the host, not faulted on the GPU
    assert L.lib().c3d_conv_forward(ctypes.byref(d), None) != 0
    assert b"null" in L.lib().c3d_last_error()
    assert L.lib().c3d_conv_wgrad(ctypes.byref(L.WgradDesc()), None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    from coarse3d_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libcoarse3d_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.lib()


def test_state_dict_surface_matches_reference_names():
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    m = SalsaNextProto(5, 20, 20, 0, use_prototype=True)
    sd = m.state_dict()
    ref = oc.init_state()
    assert set(sd.keys()) == set(ref.keys())
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sum(p.numel() for p in m.parameters()) == 7492732
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 7390332
    m.load_state_dict(W.closed_form_state())
    # encoder key list of the reference's encoder_module.yaml: 198 keys downCntx.* .. resBlock5.bn4.*
    enc = [k for k in sd if k.split(".")[0] in ("downCntx", "downCntx2", "downCntx3", "resBlock1", "resBlock2",
                                                "resBlock3", "resBlock4", "resBlock5")]
    assert len(enc) == 198


def test_constructor_rejects_unsupported_modes():
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    with pytest.raises(ValueError):
        SalsaNextProto(classification=True)


def test_taps_and_select_ratio():
    from coarse3d_amd import ops
    from coarse3d_amd.trainer import select_ratio_for
    assert ops.conv_taps(3, 3, 2, 2)[0] == (-2, -2) and ops.conv_taps(3, 3, 2, 2)[8] == (2, 2)
    assert ops.conv_taps(2, 2, 2, 1) == [(-1, -1), (-1, 1), (1, -1), (1, 1)]
    assert ops.negate_taps([(1, -2)]) == [(-1, 2)]
    assert abs(select_ratio_for(10, 100) - oc.select_ratio_for(10, 100)) < 1e-15


def test_lovasz_and_focal_match_golden():
    """The sync-free batched Lovasz / focal restatements equal the reference's fixtures."""
    from coarse3d_amd.pc_processor.loss import FocalSoftmaxLoss, Lovasz_softmax
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "losses.npz")).items()}
    prob = g["prob"].clone().requires_grad_(True)
    tr = g["train_label"]
    lf = FocalSoftmaxLoss(prob.shape[1], gamma=2, alpha=g["alpha"].numpy(), softmax=False)(prob, tr, mask=tr > 0)
    ll = Lovasz_softmax(ignore=0, per_image=False, softmax=False)(prob, tr)
    torch.testing.assert_close(lf, g["focal"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ll, g["lovasz"], rtol=1e-5, atol=1e-7)
    gf, = torch.autograd.grad(lf, prob, retain_graph=True)
    gl, = torch.autograd.grad(ll, prob)
    torch.testing.assert_close(gf, g["grad_focal"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(gl, g["grad_lovasz"], rtol=1e-4, atol=1e-7)


def test_default_init_matches_reference_checksums():
    """Same constructor order => same RNG consumption => bit-identical default weights under one
    seed (golden: per-tensor checksums of the reference's SalsaNextProto under seed 7)."""
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    g = np.load(os.path.join(ROOT, "tests", "golden", "init_checksums.npz"))
    torch.manual_seed(7)
    sd = SalsaNextProto(5, 20, 20, 0).state_dict()
    assert list(sd.keys()) == list(g["keys"])
    for k, v in sd.items():
        v = v.double()
        assert float(v.sum()) == float(g[f"sum/{k}"]), k
        assert float((v * v).sum()) == float(g[f"sq/{k}"]), k
