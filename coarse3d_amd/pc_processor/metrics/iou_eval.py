"""Confusion-matrix metrics on the device (reference pc_processor/metrics/iou_eval.py:9-119).

Same class name, constructor and methods as the reference ``IOUEval``; the confusion matrix
lives in HBM (int64 [C,C]) and is accumulated by the HIP kernels of csrc/metric_ops.hip, so the
per-iteration metrics of the trainer (trainer.py:713-730) need no host synchronisation.
``addBatchFromProbs`` fuses the argmax over classes, the range-image -> point un-projection and
the accumulation (the reference does them as three indexing passes per scan)."""
import torch

from ... import ops


class IOUEval:
    def __init__(self, n_classes, device=None, ignore=None, is_distributed=False):
        self.n_classes = n_classes
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("IOUEval accumulates on the GPU (HIP kernel); there is no CPU fallback")
        ignore = [] if ignore is None else list(ignore)
        self.ignore = torch.tensor(ignore, dtype=torch.long, device=self.device)
        self.include = torch.tensor([n for n in range(n_classes) if n not in ignore], dtype=torch.long,
                                    device=self.device)
        self.is_distributed = is_distributed
        self.reset()

    def num_classes(self):
        return self.n_classes

    def reset(self):
        self.conf_matrix = torch.zeros(self.n_classes, self.n_classes, dtype=torch.long, device=self.device)

    # ---- accumulation
    def addBatch(self, x, y):
        """x = predictions, y = targets (any shape, integer classes): conf[x][y] += 1 (iou_eval.py:35-58)."""
        x = torch.as_tensor(x).to(self.device, torch.long).reshape(-1).contiguous()
        y = torch.as_tensor(y).to(self.device, torch.long).reshape(-1).contiguous()
        if x.numel() != y.numel():
            raise ValueError("predictions and targets differ in size: {} vs. {}".format(x.numel(), y.numel()))
        ops.confusion_add(x, y, self.conf_matrix)

    def addBatchFromProbs(self, pred_2d, uproj_y_idx, uproj_x_idx, labels, n_points=None):
        """One scan: pred_2d [C,H,W]-shaped probabilities (any strides; channels-last memory is
        read in place), pixel coordinates of every point and their labels.  ``uproj_x_idx=None``
        selects the SemanticPOSS convention (uproj_y_idx is the flat pixel index of the first
        ``len(uproj_y_idx)`` points; the remaining ``n_points - len`` points predict class 0,
        trainer.py:720-726).  Returns the un-projected argmax (int32 [n])."""
        return ops.unproject_confusion(pred_2d, uproj_y_idx, uproj_x_idx, labels, self.conf_matrix, n_points)

    # ---- statistics (tiny [C] vectors; torch ops on the device, no host sync)
    def getStats(self):
        conf = self.conf_matrix.clone().double()
        if self.is_distributed:
            import torch.distributed as dist
            dist.all_reduce(conf)
        conf[self.ignore] = 0
        conf[:, self.ignore] = 0
        tp = conf.diag()
        fp = conf.sum(dim=1) - tp
        fn = conf.sum(dim=0) - tp
        return tp, fp, fn

    def getIoU(self):
        tp, fp, fn = self.getStats()
        union = tp + fp + fn + 1e-15
        iou = tp / union
        return (tp[self.include] / union[self.include]).mean(), iou

    def getacc(self):
        tp, fp, fn = self.getStats()
        return tp.sum() / (tp[self.include].sum() + fp[self.include].sum() + 1e-15)

    def getAcc(self):
        tp, fp, fn = self.getStats()
        acc = tp / (tp + fp + 1e-15)
        return acc[self.include].mean(), acc

    def getRecall(self):
        tp, fp, fn = self.getStats()
        recall = tp / (tp + fn + 1e-15)
        return recall[self.include].mean(), recall
