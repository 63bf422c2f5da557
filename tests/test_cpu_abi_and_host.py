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
    out = (ctypes.c_int32 * 5)()
    assert L.lib().c3d_abi_sizes(out) == 0
    assert list(out) == [ctypes.sizeof(L.Src), ctypes.sizeof(L.ConvDesc), ctypes.sizeof(L.WgradDesc),
                         ctypes.sizeof(L.PackEntry), ctypes.sizeof(L.WgradFold)]
    from coarse3d_amd.peer import PeerDesc
    assert L.lib().c3d_peer_desc_bytes() == ctypes.sizeof(PeerDesc) == 88
    assert L.lib().c3d_peer_mailbox_bytes(8192) == 256 + 2 * 8 * 8 + 2 * 8 * 8192 * 8
    pd = PeerDesc()
    pd.rank, pd.world, pd.cap_doubles = 3, 2, 16          # refused on the host: rank outside the group
    assert L.lib().c3d_peer_allreduce_f64(ctypes.byref(pd), ctypes.c_void_p(8), 4, None) != 0
    assert b"rank" in L.lib().c3d_last_error()
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


def test_classification_mode_has_the_reference_state_dict_surface():
    """``classification=True`` (salsanext_proto.py:216-231, 308-309; built in round 5): the FC head's parameters appear
    where the reference registers them -- behind resBlock5, ahead of upBlock1 -- under the reference's names."""
    from coarse3d_amd.pc_processor.models import SalsaNextProto
    m = SalsaNextProto(classification=True)
    keys = list(m.state_dict().keys())
    i = keys.index("fc.linear.weight")
    assert keys[i - 1] == "resBlock5.bn4.num_batches_tracked" and keys[i + 1] == "fc.linear.bias" and keys[i + 2] == "upBlock1.conv1.weight"
    assert tuple(m.fc.linear.weight.shape) == (1000, 256)
    import weights as W
    sd = W.closed_form_state()
    sd.update(W.fc_state())
    m.load_state_dict(sd)


def test_the_other_backbones_carry_the_attributes_the_shared_forward_reads():
    """RangeNetProto / SqueezeSegV3Proto reuse SalsaNextProto.forward but run their own constructors: every attribute that
    forward reads must exist on them (a missing ``classification`` broke their data-parallel step once)."""
    from coarse3d_amd.pc_processor.models import RangeNetProto, SalsaNextProto, SqueezeSegV3Proto
    base = SalsaNextProto()
    for m in (RangeNetProto(layers=21, nclasses=20, use_prototype=True), SqueezeSegV3Proto(nclasses=20, layers=21, use_prototype=True)):
        assert m.classification is False and m.graph_backbone is False and m._gb is None
        missing = [k for k in vars(base) if not k.startswith("_") and not hasattr(m, k)
                   and k not in base._modules and k not in base._parameters and k not in base._buffers
                   and k not in ("softmax", "base_channels")]      # (stored as the reference does / read by SalsaNext's own mask draw only)
        assert not missing, missing


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


def test_module_caches_follow_the_parameters():
    """The module mirror builds its name -> tensor dictionaries once (host time at the step boundary).  They must
    alias the live parameters (in-place optimiser updates, load_state_dict), be dropped when storage moves (_apply)
    and the direct gradient binding must step aside whenever a gradient has to be accumulated."""
    from coarse3d_amd.pc_processor.models import SalsaNextProto, SqueezeSegV3Proto
    m = SalsaNextProto(5, 20, 20, 0, use_prototype=True)
    named, names, d = m._cached()
    assert m._cached() is m._cache and len(named) == len(names) == 192
    assert not any(k.startswith(("prototypes", "feat_norm", "mask_norm")) for k in names)
    w = dict(m.named_parameters())["downCntx.conv1.weight"]
    with torch.no_grad():
        w.add_(1.0)                                           # what an optimiser does
    assert d["downCntx.conv1.weight"].data_ptr() == w.data_ptr() and torch.equal(d["downCntx.conv1.weight"], w)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["downCntx.conv1.weight"].zero_()
    m.load_state_dict(sd)
    assert float(d["downCntx.conv1.weight"].abs().max()) == 0.0   # copied in place: the cache still aliases it
    m.double()                                                # storage moves: cache dropped and rebuilt
    assert m._cache is None and m._cached()[2]["downCntx.conv1.weight"].dtype == torch.float64
    m.float()
    # direct binding: off by default, on request one flat buffer in parameter order, off again while a grad is set
    assert m._bound_grad_views(m._cached()[1]) is None
    m._bind_grads = True
    v = m._bound_grad_views(m._cached()[1])
    flat = m._own_flat[1]
    assert flat.numel() == sum(p.numel() for _, p in m._cached()[0])
    assert all(v[n].shape == p.shape and v[n].data_ptr() >= flat.data_ptr() for n, p in m._cached()[0])
    assert m._bound_grad_views(m._cached()[1]) is v           # persistent across steps
    m._cached()[0][3][1].grad = torch.zeros_like(m._cached()[0][3][1])
    assert m._bound_grad_views(m._cached()[1]) is None        # something to accumulate into: autograd's path
    # subclasses keep their own trainable set (SqueezeSegV3: the reference's unused heads get no gradient)
    s = SqueezeSegV3Proto(nclasses=20, layers=21, use_prototype=True)
    assert not any(k.startswith(("head1.", "head2.", "head3.", "head4.")) for k in s._cached()[1])
    assert [k for k, _ in s._trainable()] == list(s._cached()[1])
    # dropout masks: contiguous views of one draw, right shapes and values
    masks = m._draw_masks(3, "cpu")
    assert all(t.is_contiguous() and t.shape[0] == 3 for t in masks.values())
    assert set(float(x) for x in torch.cat([t.flatten() for t in masks.values()]).unique()) <= {0.0, 1.0 / 0.8}


def test_flat_adamw_checkpoints_are_torch_adamw_checkpoints():
    """the reference saves ``optimizer.state_dict()`` (tasks/weak_segmentation/main.py:141,154) and
    resumes with ``optimizer.load_state_dict`` (trainer.py:129).  FlatAdamW must write and read that layout:
    state moves FlatAdamW -> torch.optim.AdamW -> FlatAdamW with identical continued updates; round 2's flat
    layout still loads; a parameter that left the flat buffer makes step() raise instead of training nothing."""
    import torch
    from coarse3d_amd.optim import FlatAdamW

    def make(seed):
        g = torch.Generator().manual_seed(seed)
        ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(4, 3), (7,), (2, 2, 2)]]
        return [(f"p{i}", p) for i, p in enumerate(ps)]

    def grads(step, shapes):
        g = torch.Generator().manual_seed(100 + step)
        return [torch.randn(s, generator=g) for s in shapes]

    named = make(0)
    shapes = [tuple(p.shape) for _, p in named]
    total = sum(p.numel() for _, p in named)
    flat_grad = torch.zeros(total)
    views, off = {}, 0
    for n, p in named:
        views[n] = flat_grad[off:off + p.numel()].view_as(p)
        off += p.numel()
    opt = FlatAdamW(named, views, flat_grad, lr=1e-2)
    assert opt.state_dict()["state"] == {}                               # like torch before the first step
    ref_params = [torch.nn.Parameter(p.detach().clone()) for _, p in make(0)]
    ref = torch.optim.AdamW(ref_params, lr=1e-2)
    for step in range(3):
        for (n, p), rp, gr in zip(named, ref_params, grads(step, shapes)):
            views[n].copy_(gr)
            p.grad = views[n]
            rp.grad = gr.clone()
        opt.step()
        ref.step()
    for (_, p), rp in zip(named, ref_params):
        assert torch.allclose(p, rp, rtol=1e-6, atol=1e-7)
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == [0, 1, 2]
    assert set(sd["state"][1]) == {"step", "exp_avg", "exp_avg_sq"} and sd["state"][2]["exp_avg"].shape == (2, 2, 2)
    # FlatAdamW checkpoint -> torch.optim.AdamW -> continue; and the torch checkpoint -> a fresh FlatAdamW -> continue
    cont_params = [torch.nn.Parameter(p.detach().clone()) for _, p in named]
    cont = torch.optim.AdamW(cont_params, lr=1e-2)
    cont.load_state_dict(sd)
    named2 = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in named]
    fg2 = torch.zeros(total)
    views2, off = {}, 0
    for n, p in named2:
        views2[n] = fg2[off:off + p.numel()].view_as(p)
        off += p.numel()
    opt2 = FlatAdamW(named2, views2, fg2, lr=1e-2)
    opt2.load_state_dict(ref.state_dict())
    for (n, p), cp, rp, gr in zip(named2, cont_params, ref_params, grads(7, shapes)):
        views2[n].copy_(gr)
        p.grad = views2[n]
        cp.grad = gr.clone()
        rp.grad = gr.clone()
    opt2.step()
    cont.step()
    ref.step()
    for (_, p), cp, rp in zip(named2, cont_params, ref_params):
        assert torch.allclose(p, rp, rtol=1e-6, atol=1e-7) and torch.allclose(cp, rp, rtol=1e-6, atol=1e-7)
    # round 2's flat layout is still accepted
    opt2.load_state_dict({"flat": True, "step": opt.step_t, "exp_avg": opt.exp_avg, "exp_avg_sq": opt.exp_avg_sq,
                          "param_groups": [{"lr": 5e-3}]})
    assert opt2.param_groups[0]["lr"] == 5e-3 and float(opt2.step_t) == 3
    # per-parameter step counts that differ (parameters that were not always trained together: the projector during the
    # reference's contrast warm-up) become separate segments of the flat update -- and keep matching torch
    odd = ref.state_dict()
    odd["state"][1]["step"] = odd["state"][1]["step"] + 1
    opt2.load_state_dict(odd)
    assert len(opt2._segments) == 3 and [float(s[2]) for s in opt2._segments] == [4.0, 5.0, 4.0]
    odd_params = [torch.nn.Parameter(p.detach().clone()) for _, p in named2]
    odd_ref = torch.optim.AdamW(odd_params, lr=opt2.param_groups[0]["lr"])
    odd_ref.load_state_dict(odd)
    for (n, p), op, gr in zip(named2, odd_params, grads(9, shapes)):
        views2[n].copy_(gr)
        p.grad = views2[n]
        op.grad = gr.clone()
    opt2.step()
    odd_ref.step()
    for (_, p), op in zip(named2, odd_params):
        assert torch.allclose(p, op, rtol=1e-6, atol=1e-7)
    back = opt2.state_dict()["state"]
    assert float(back[1]["step"]) == 6.0 and float(back[0]["step"]) == 5.0
    # a parameter WITHOUT a gradient is skipped like torch skips it: no decay, no step, no state movement
    before = named2[1][1].detach().clone()
    named2[1][1].grad = None
    opt2.step()
    assert torch.equal(named2[1][1].detach(), before) and float(opt2.state_dict()["state"][1]["step"]) == 6.0
    assert float(opt2.state_dict()["state"][0]["step"]) == 6.0
    # a parameter that left the flat buffer
    named2[0][1].data = named2[0][1].data.clone()
    with pytest.raises(RuntimeError):
        opt2.step()


def test_flat_adamw_round_trips_with_torch_adamw_on_the_real_module():
    """the reference builds ``AdamW(model.parameters())`` -- 197 parameters for SalsaNextProto, of which the
    frozen ``prototypes`` (index 0) and the ``feat_norm`` / ``mask_norm`` LayerNorms (193-196) never get state -- and
    checkpoints that layout (trainer.py:129,146-151).  FlatAdamW steps the 192 trained tensors as one buffer but must
    write and read the 197-entry layout: torch -> FlatAdamW -> torch on the real module, with identical continued
    updates; during a contrast warm-up (no gradient for the projector, config_semantic_kitti.yaml:20) the projector's
    parameters are skipped and come back without state, as torch leaves them."""
    import torch
    from coarse3d_amd.optim import FlatAdamW
    from coarse3d_amd.pc_processor.models import SalsaNextProto

    def model(seed):
        torch.manual_seed(seed)
        return SalsaNextProto(5, 20, 20, 0, use_prototype=True)

    def fake_grads(m, step, skip_projector):
        g = torch.Generator().manual_seed(500 + step)
        out = {}
        for n, p in m.named_parameters():
            if n in m._SKIP or (skip_projector and n.startswith("projector.")):
                continue
            out[n] = torch.randn(p.shape, generator=g) * 0.01
        return out

    # torch's optimiser on the reference's terms: two warm-up steps without projector gradients, one full step
    m_t = model(3)
    names_all = [n for n, _ in m_t.named_parameters()]
    assert len(names_all) == 197 and names_all[0] == "prototypes" and names_all[193].startswith("feat_norm")
    opt_t = torch.optim.AdamW(m_t.parameters(), lr=1e-3)
    m_f = model(3)
    m_f._bind_grads = True
    named, names, _ = m_f._cached()
    views = m_f._bound_grad_views(names)
    opt_f = FlatAdamW(named, views, m_f._own_flat[1], lr=1e-3, all_params=list(m_f.parameters()))
    assert len(opt_f.param_groups[0]["params"]) == 197
    for step, warm in enumerate((True, True, False)):
        gr = fake_grads(m_t, step, warm)
        P_t, P_f = dict(m_t.named_parameters()), dict(m_f.named_parameters())
        for n in names:
            P_t[n].grad = gr[n].clone() if n in gr else None
            if n in gr:
                views[n].copy_(gr[n])
                P_f[n].grad = views[n]
            else:
                P_f[n].grad = None
        opt_t.step()
        opt_f.step()
    for (n, a), (_, b) in zip(m_t.named_parameters(), m_f.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8), n
    sd_t, sd_f = opt_t.state_dict(), opt_f.state_dict()
    assert sd_f["param_groups"][0]["params"] == list(range(197)) == sd_t["param_groups"][0]["params"]
    assert set(sd_f["state"]) == set(sd_t["state"]) and 0 not in sd_f["state"] and 193 not in sd_f["state"]
    proj = [i for i, n in enumerate(names_all) if n.startswith("projector.")]
    assert all(float(sd_f["state"][i]["step"]) == 1.0 for i in proj) and float(sd_f["state"][1]["step"]) == 3.0
    for i in sd_t["state"]:
        assert float(sd_t["state"][i]["step"]) == float(sd_f["state"][i]["step"])
        assert torch.allclose(sd_t["state"][i]["exp_avg_sq"], sd_f["state"][i]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    # FlatAdamW checkpoint -> torch.optim.AdamW(net.parameters()); torch checkpoint -> a fresh FlatAdamW; one more step each
    m_t2 = model(3)
    m_t2.load_state_dict(m_f.state_dict())
    opt_t2 = torch.optim.AdamW(m_t2.parameters(), lr=1e-3)
    import copy
    opt_t2.load_state_dict(copy.deepcopy(sd_f))           # (torch aliases the step tensors of the dict it loads)
    m_f2 = model(3)
    m_f2.load_state_dict(m_t.state_dict())
    m_f2._bind_grads = True
    named2, names2, _ = m_f2._cached()
    views2 = m_f2._bound_grad_views(names2)
    opt_f2 = FlatAdamW(named2, views2, m_f2._own_flat[1], lr=1e-3, all_params=list(m_f2.parameters()))
    opt_f2.load_state_dict(copy.deepcopy(sd_t))
    assert len(opt_f2._segments) == 2                      # [backbone + head | projector]
    gr = fake_grads(m_t, 9, False)
    for mm, vv in ((m_t, None), (m_t2, None), (m_f2, views2)):
        for n, p in mm.named_parameters():
            if n in gr:
                if vv is None:
                    p.grad = gr[n].clone()
                else:
                    vv[n].copy_(gr[n])
                    p.grad = vv[n]
    opt_t.step()
    opt_t2.step()
    opt_f2.step()
    for (n, a), (_, b), (_, c) in zip(m_t.named_parameters(), m_t2.named_parameters(), m_f2.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8) and torch.allclose(a, c, rtol=1e-6, atol=1e-8), n
    # round 3's checkpoints (the 192 trained parameters only) still load
    old = {"state": {j: sd_f["state"][i] for j, i in enumerate(opt_f._index) if i in sd_f["state"]},
           "param_groups": [dict(sd_f["param_groups"][0], params=list(range(192)))]}
    opt_f2.load_state_dict(old)
    assert float(opt_f2.state_dict()["state"][1]["step"]) == 3.0


def test_tile_row_rule_of_the_host_mirror_matches_the_library():
    """ops._tile_rows (used to name the kernel a conv launch will run, for the bench's per-kernel timers) restates
    c3d_tile_rows of csrc/conv_mfma.hip, which decides the number of BatchNorm-statistic partials of every conv over an
    image (c3d_conv_num_mtiles, a host call): the two must agree for every height, and 8-row tiles -- the only ones the
    fused bf16x3 kernels exist for -- are chosen wherever they pad the image by at most a third."""
    from coarse3d_amd import ops
    for h in range(1, 200):
        tr = ops._tile_rows(h)
        assert tr in (8, 4, 2)
        assert ops.num_mtiles(3, h, 70) == 3 * ((h + tr - 1) // tr) * 3, h
        if ((h + 7) // 8 * 8) * 3 <= h * 4:
            assert tr == 8, h
    assert [ops._tile_rows(h) for h in (64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2)] == [8, 8, 8, 8, 8, 8, 8, 8, 4, 4, 2]
