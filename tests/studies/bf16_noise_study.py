"""Study (CPU, not a test): how far does a training step of this network move when every
convolution's operands are rounded to bf16 (fp32 accumulate/storage), using the CPU oracle?
Result (2x64x512, 2 % labels, seed 1): class probabilities move 3.6e-3 on average, focal loss
0.1 %, weight-gradient cosine vs fp32 0.88-0.91 in upBlock4, 0.5-0.6 in the encoder -- the same
figures the HIP bf16-operand mode shows (tests/test_gpu_configs.py), i.e. the decorrelation is a
property of the network at initialisation, not of the kernels.
Run from the repo root: python tests/studies/bf16_noise_study.py"""
import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch.nn.functional as F
from oracle import coarse3d_oracle as oc
import weights as W, bench
torch.set_num_threads(8)
def bf(t): return t.to(torch.bfloat16).to(torch.float32)
real_conv = F.conv2d
class BFConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation):
        ctx.save_for_backward(x, w); ctx.cfg = (stride, padding, dilation)
        return real_conv(bf(x), bf(w), b, stride=stride, padding=padding, dilation=dilation)
    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors; s, p, d = ctx.cfg
        dzb = bf(dz)
        dx = torch.nn.grad.conv2d_input(x.shape, bf(w), dzb, stride=s, padding=p, dilation=d)
        dw = torch.nn.grad.conv2d_weight(bf(x), w.shape, dzb, stride=s, padding=p, dilation=d)
        return dx, dw, dz.sum(dim=(0, 2, 3)), None, None, None
def fake_conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    return BFConv.apply(x, w, b, stride, padding, dilation)
def run(mode, dtype=torch.float32):
    b, h, w, ncls = 2, 64, 512, 20
    st = oc.init_state(nclasses=ncls, seed=1)
    st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    for k in oc.trainable_names(st): st[k].requires_grad_(True)
    x, tr, ev = bench.synth_batch(b, h, w, ncls, 1000, 'cpu', label_rate=2e-2)
    masks = W.dropout_masks_for(None, b, 11)
    masks = {k: v.to(dtype) for k, v in masks.items()}
    torch.manual_seed(7)
    oc.F.conv2d = fake_conv if mode == 'bf16' else real_conv
    info, grads = oc.train_step(st, x.to(dtype), tr, ev, None, epoch=10, num_anchor=64, dropout_masks=masks, w_contrast=0.0,
                                mean=torch.tensor(bench.FEATURE_MEAN, dtype=dtype), std=torch.tensor(bench.FEATURE_STD, dtype=dtype), use_prototype=False)
    oc.F.conv2d = real_conv
    return info, grads
i32, g32 = run('f32'); i16, g16 = run('bf16')
print('ce', float(i32['ce']), float(i16['ce']), 'pred mean abs diff', float((i32['pred_2d'] - i16['pred_2d']).abs().mean()))
for k in g32:
    a, b_ = g32[k], g16[k]
    if a is not None and a.numel() >= 4096:
        print(k, round(float((a * b_).sum() / (a.norm() * b_.norm())), 3))
