"""SURVEY 8f N1, metrics: device argmax + un-projection + confusion matrix (csrc/metric_ops.hip,
pc_processor.metrics.IOUEval mirror) against the golden vectors of the reference IOUEval and
against the CPU oracle.  Integer work: bit-exact; the derived ratios to 1e-12."""
import os

import numpy as np
import pytest
import torch

from oracle import coarse3d_oracle as oc

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("tag,ncls,poss", [("kitti", 20, False), ("poss", 14, True)])
def test_iou_eval_matches_reference_golden(tag, ncls, poss):
    from coarse3d_amd.pc_processor.metrics import IOUEval
    d = np.load(os.path.join(GOLD, "metrics.npz"))
    pred = torch.from_numpy(d[f"{tag}/pred_2d"]).to(DEV)
    ev = IOUEval(ncls, ignore=[0])
    ev2 = IOUEval(ncls, ignore=[0])
    for ii in range(pred.shape[0]):
        uy = torch.from_numpy(d[f"{tag}/uy{ii}"])
        labels = torch.from_numpy(d[f"{tag}/labels{ii}"])
        ux = None if poss else torch.from_numpy(d[f"{tag}/ux{ii}"])
        # NCHW-contiguous input (copied to NHWC) and a channels-last view (read in place)
        un = ev.addBatchFromProbs(pred[ii], uy, ux, labels, labels.numel())
        nhwc = pred[ii].permute(1, 2, 0).contiguous()
        un2 = ev2.addBatchFromProbs(nhwc.permute(2, 0, 1), uy, ux, labels, labels.numel())
        want = torch.from_numpy(d[f"{tag}/unproj{ii}"])
        assert torch.equal(un.cpu().long(), want) and torch.equal(un2.cpu().long(), want)
    want_conf = torch.from_numpy(d[f"{tag}/conf"])
    assert torch.equal(ev.conf_matrix.cpu(), want_conf) and torch.equal(ev2.conf_matrix.cpu(), want_conf)
    for name, (mean, per) in (("iou", ev.getIoU()), ("acc", ev.getAcc()), ("recall", ev.getRecall())):
        assert abs(float(mean) - float(d[f"{tag}/{name}_mean"])) < 1e-12
        assert float((per.cpu() - torch.from_numpy(d[f"{tag}/{name}"])).abs().max()) < 1e-12


def test_add_batch_and_full_size_against_oracle():
    """bench-size scan (64x2048, 120k points, C=20) + plain addBatch, vs the CPU oracle; the
    confusion matrix sums to the number of points (nothing lost in the block-local histograms)."""
    from coarse3d_amd.pc_processor.metrics import IOUEval
    g = torch.Generator().manual_seed(3)
    ncls, h, w, n = 20, 64, 2048, 120_000
    prob = torch.softmax(torch.randn(ncls, h, w, generator=g) * 2, 0)
    uy = torch.randint(0, h, (n,), generator=g)
    ux = torch.randint(0, w, (n,), generator=g)
    labels = torch.randint(0, ncls, (n,), generator=g)
    ev = IOUEval(ncls, ignore=[0])
    for _ in range(3):
        un = ev.addBatchFromProbs(prob.to(DEV), uy, ux, labels)
    ref_un = oc.unproject_argmax(prob, uy, ux)
    assert torch.equal(un.cpu().long(), ref_un)
    conf = torch.zeros(ncls, ncls, dtype=torch.long)
    for _ in range(3):
        oc.confusion_add(conf, ref_un, labels)
    assert torch.equal(ev.conf_matrix.cpu(), conf) and int(ev.conf_matrix.sum()) == 3 * n
    ev.reset()
    ev.addBatch(ref_un.numpy(), labels.numpy())          # numpy inputs, as iou_eval.py:36-41 accepts
    ev.addBatch(ref_un.to(DEV), labels.to(DEV))
    assert torch.equal(ev.conf_matrix.cpu(), 2 * conf // 3)
    st = oc.iou_stats(ev.conf_matrix.cpu(), [0])
    assert abs(float(ev.getIoU()[0]) - float(st["iou"][0])) < 1e-12
    with pytest.raises(ValueError):
        ev.addBatch(ref_un[:-1], labels)
