#!/usr/bin/env python3
"""CPU emulation: how accurate would R's convolutions be on the bf16 MFMA with 2-way (3 products) or 3-way (6 products)
operand splitting and fp32 accumulation?  Compares R forward outputs / flat gradients against float64 and against plain
fp32, at a cfg2-like size.  (Experiment only; informs DESIGN.md section 6.)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn.functional as F
from ganrev import models, synth
import torch_twin as tt

def split(x, n):
    if SplitConv.dtype is torch.float16:
        # fp16 terms need the tensor scaled into range: power-of-two scale putting max|x| into [2^14, 2^15)
        m = float(x.abs().max()); sc = 2.0 ** (14 - np.floor(np.log2(m))) if m > 0 else 1.0
        parts, r = [], x * sc
        for _ in range(n):
            p = r.to(torch.float16).to(torch.float32); parts.append(p / sc); r = r - p
        return parts
    parts, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32); parts.append(p); r = r - p
    return parts

class SplitConv(torch.autograd.Function):
    n = 2
    dtype = torch.bfloat16
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return SplitConv.conv(x, w, lambda a, c: F.conv2d(a, c, None, padding=1)) + b.view(1, -1, 1, 1)
    @staticmethod
    def conv(a, c, op):
        pa, pc = split(a, SplitConv.n), split(c, SplitConv.n)
        out = 0
        for i, ai in enumerate(pa):
            for j, cj in enumerate(pc):
                if i + j < SplitConv.n:      # 2-way: hh, hl, lh ; 3-way: the 6 terms of order < 3
                    out = out + op(ai, cj)
        return out
    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = SplitConv.conv(gy, w, lambda a, c: F.conv_transpose2d(a, c, None, padding=1))
        gw = SplitConv.conv(x, gy, lambda a, c: torch.nn.grad.conv2d_weight(a, w.shape, c, padding=1))
        return gx, gw, gy.sum((0, 2, 3))

class Twin32(tt.Twin):
    mode = "fp32"
    def forward(self, x):
        # same as Twin.forward but float32 and pluggable conv
        x = torch.tensor(np.asarray(x, np.float32)); B = x.shape[0]; bi = 0
        for li, (d, p) in enumerate(zip(self.descs, self.params)):
            k = d[0]
            if k == tt.CONV3:
                x = F.conv2d(x, p[0], p[1], padding=1) if self.mode == "fp32" else SplitConv.apply(x, p[0], p[1])
            elif k == tt.LINEAR: x = F.linear(x.reshape(B, -1), p[0], p[1])
            elif k == tt.BN:
                rm, rv = self.bn_running[bi]; bi += 1
                x = F.batch_norm(x, rm.float(), rv.float(), p[0], p[1], True, 0.1, 1e-5)
            elif k == tt.ELU: x = F.elu(x)
            elif k == tt.DROPOUT:
                keep = torch.tensor(self.masks[li].astype(np.float32)).reshape(x.shape); x = x * keep * 2.0
            elif k == tt.SPATIAL_DROPOUT:
                keep = torch.tensor(self.masks[li].astype(np.float32)).reshape(B, x.shape[1], 1, 1); x = x * keep
            elif k == tt.MAXPOOL2: x = F.max_pool2d(x, 2, 2)
            elif k == tt.VIEW: x = x.reshape(B, d[1])
        self.out = x
        return x.detach().numpy()

dims, nd, B = (1, 32, 32), 32, 32
R = models.create_R(dims, nd); synth.init_params(R, 3)
descs, index = R._descs(dims)
masks = {}
d_ = dims
for m in R.leaves():
    ds, nd_ = m.desc(d_)
    if m.typename in ("nn.Dropout", "nn.SpatialDropout"):
        n = B * (int(np.prod(d_)) if m.typename == "nn.Dropout" else d_[0]); masks[index[id(m)]] = synth.bernoulli_keep((n,), 7 + index[id(m)], m.p)
    d_ = nd_
running = [(m.running_mean.copy(), m.running_var.copy()) for m in R.leaves() if hasattr(m, "running_mean")]
x = synth.uniform((B,) + dims, 5, 0, 1)
ref = tt.Twin(descs, dims, R._flat_host(), running, True, masks)
out64 = ref.forward(x); gy = synth.normal(out64.shape, 9) * np.float32(0.1); g64 = ref.backward(gy)
for mode, n, dt in (("fp32", 0, None), ("split", 2, torch.bfloat16), ("split", 3, torch.bfloat16), ("split", 2, torch.float16)):
    t = Twin32(descs, dims, R._flat_host(), running, True, masks)
    t.params = [None if p is None else tuple(q.detach().float().requires_grad_(True) for q in p) for p in t.params]
    t.mode = mode; SplitConv.n = n; SplitConv.dtype = dt
    o = t.forward(x)
    flat_p = [q for p in t.params if p is not None for q in p]
    grads = torch.autograd.grad(t.out, flat_p, torch.tensor(gy), allow_unused=True)
    g = np.concatenate([(gg if gg is not None else torch.zeros_like(q)).reshape(-1).numpy() for gg, q in zip(grads, flat_p)])
    print(f"{mode:5s} n={n} {str(dt):15s}: out max|err| {np.abs(o - out64).max():.3e}   grad max|err| {np.abs(g - g64).max():.3e} (max|g| {np.abs(g64).max():.3f})  rel-to-max {np.abs(g-g64).max()/np.abs(g64).max():.3e}")
