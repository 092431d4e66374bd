#!/usr/bin/env python3
"""Soak run of gr_train_r_step (the head kernel's grid barriers, the free-running weight-gradient halves, the persistent convolution): N steps at cfg2 /
cfg3 geometry, the loss read every `every` steps - it must stay finite (a grid barrier that timed out poisons it with NaN) and fall.
   python tools/soak_train.py [cfg2|cfg3] [steps] [every]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT]
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from ganrev.parallel import DeviceTrainer
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dims, nd, B = ((1, 32, 32), 32, 256) if wl == "cfg2" else ((3, 64, 64), 100, 512)
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
G = models.create_G(dims, nd); synth.init_params(G, 1)
R = models.create_R(dims, nd, seed=1)
G._ctx = R._ctx = ctx
G.evaluate(); G.forward(synth.normal((2, nd), 1))
R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
R._net.set_seed(1); R._net.adam_reset()
tr = DeviceTrainer(ctx, G._net, R._net, L.Hyper(), B)
t0 = time.perf_counter(); losses = []
for i in range(1, steps + 1):
    tr.new_noise(i)
    want = i % every == 0 or i == 1
    l = tr.step(want_loss=want)
    if want:
        losses.append(l)
        print(f"step {i}: loss {l:.6f}  ({(time.perf_counter() - t0) / i * 1e3:.3f} ms/step incl. host)", flush=True)
        assert np.isfinite(l), "loss is not finite: a grid barrier timed out or the step diverged"
assert losses[-1] < losses[0], (losses[0], losses[-1])
print(f"ok: {steps} steps, loss {losses[0]:.4f} -> {losses[-1]:.4f}, f16x3 guard fallbacks {ctx.range_guard_stats()[1]}, context arithmetic now {ctx.conv_mode()}")
