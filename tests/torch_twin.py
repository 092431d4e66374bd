"""An independent PyTorch-CPU (float64) evaluation of a gr_layer_desc list — used only to cross-check the oracle.
PyTorch is the lineal descendant of TH/THNN and keeps the same operator definitions (conv2d, batch_norm, elu,
max_pool2d, linear, mse_loss); it is NOT the reference and never runs on the GPU box's product path."""
import numpy as np
import torch
import torch.nn.functional as F

CONV3, BN, ELU, RELU, LEAKYRELU, SIGMOID, TANH, DROPOUT, SPATIAL_DROPOUT, MAXPOOL2, UPSAMPLE2, VIEW, LINEAR, FULLCONV3 = range(1, 15)
CONVK, PRELU = 15, 16


class Twin:
    def __init__(self, descs, in_dims, flat_params, bn_running, training=True, masks=None, bn_groups=1):
        self.descs, self.in_dims, self.training = descs, in_dims, training
        self.masks = masks or {}
        self.groups = bn_groups
        self.params = []
        off = 0
        flat = torch.tensor(np.asarray(flat_params, np.float64))
        c, h, w = in_dims
        for d in descs:
            k, a, b = d[0], d[1], d[2]
            if k in (CONV3, FULLCONV3):
                shape = (b, a, 3, 3) if k == CONV3 else (a, b, 3, 3)
                wt = flat[off:off + a * b * 9].reshape(shape).clone().requires_grad_(True); off += a * b * 9
                bs = flat[off:off + b].clone().requires_grad_(True); off += b
                self.params.append((wt, bs))
            elif k == CONVK:
                kk = d[3]
                wt = flat[off:off + a * b * kk * kk].reshape(b, a, kk, kk).clone().requires_grad_(True); off += a * b * kk * kk
                bs = flat[off:off + b].clone().requires_grad_(True); off += b
                self.params.append((wt, bs))
            elif k == PRELU:
                wt = flat[off:off + 1].clone().requires_grad_(True); off += 1
                self.params.append((wt,))
            elif k == LINEAR:
                wt = flat[off:off + a * b].reshape(b, a).clone().requires_grad_(True); off += a * b
                bs = flat[off:off + b].clone().requires_grad_(True); off += b
                self.params.append((wt, bs))
            elif k == BN:
                wt = flat[off:off + a].clone().requires_grad_(True); off += a
                bs = flat[off:off + a].clone().requires_grad_(True); off += a
                self.params.append((wt, bs))
            else:
                self.params.append(None)
        assert off == flat.numel()
        self.bn_running = [(torch.tensor(np.asarray(m, np.float64)), torch.tensor(np.asarray(v, np.float64))) for m, v in bn_running]

    def forward(self, x):
        x = torch.tensor(np.asarray(x, np.float64))
        B = x.shape[0]
        bi = 0
        for li, (d, p) in enumerate(zip(self.descs, self.params)):
            k = d[0]
            if k == CONV3:
                x = F.conv2d(x, p[0], p[1], padding=1)
            elif k == FULLCONV3:
                x = F.conv_transpose2d(x, p[0], p[1], stride=1, padding=1)
            elif k == CONVK:
                x = F.conv2d(x, p[0], p[1], padding=(d[3] - 1) // 2)
            elif k == PRELU:
                x = F.prelu(x, p[0])
            elif k == LINEAR:
                x = F.linear(x.reshape(B, -1), p[0], p[1])
            elif k == BN:
                rm, rv = self.bn_running[bi]; bi += 1
                if self.training:
                    outs = []
                    for g in range(self.groups):
                        xs = x[g * (B // self.groups):(g + 1) * (B // self.groups)]
                        outs.append(F.batch_norm(xs, rm, rv, p[0], p[1], True, 0.1, 1e-5))
                    x = torch.cat(outs, 0)
                else:
                    x = F.batch_norm(x, rm, rv, p[0], p[1], False, 0.1, 1e-5)
            elif k == ELU:
                x = F.elu(x)
            elif k == RELU:
                x = F.relu(x)
            elif k == LEAKYRELU:
                x = F.leaky_relu(x, d[4])
            elif k == SIGMOID:
                x = torch.sigmoid(x)
            elif k == TANH:
                x = torch.tanh(x)
            elif k == DROPOUT:
                v2, always = d[5] & 1, d[5] & 2
                if self.training or always:
                    keep = torch.tensor(self.masks[li].astype(np.float64)).reshape(x.shape)
                    x = x * keep * (1.0 / (1.0 - d[4]) if v2 else 1.0)
                elif not v2:
                    x = x * (1.0 - d[4])
            elif k == SPATIAL_DROPOUT:
                if self.training:
                    keep = torch.tensor(self.masks[li].astype(np.float64)).reshape(B, x.shape[1], 1, 1)
                    x = x * keep
                else:
                    x = x * (1.0 - d[4])
            elif k == MAXPOOL2:
                x = F.max_pool2d(x, 2, 2)
            elif k == UPSAMPLE2:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            elif k == VIEW:
                x = x.reshape((B, d[1]) if (d[2] <= 1 and d[3] <= 1) else (B, d[1], d[2], d[3]))
        self.out = x
        return x.detach().numpy()

    def backward(self, gout):
        flat_p = [q for p in self.params if p is not None for q in p]
        grads = torch.autograd.grad(self.out, flat_p, torch.tensor(np.asarray(gout, np.float64)), allow_unused=True)
        return np.concatenate([(g if g is not None else torch.zeros_like(q)).reshape(-1).numpy() for g, q in zip(grads, flat_p)])
