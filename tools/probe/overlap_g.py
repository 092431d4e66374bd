#!/usr/bin/env python3
"""Probe (round 6): would running G's forward of step t + 1 beside R's forward / backward of step t pay?  Two contexts (two streams) on one GPU:
B runs whole train_r steps, A runs extra G forwards.  T1 = a step alone, T2 = a G forward alone, T3 = one of each enqueued together.
If the kernels of the two streams overlapped usefully, T3 would be well under T1 + T2 (its floor is max(T1, T2) ~ T1).
   python tools/probe/overlap_g.py [cfg2|cfg3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT]
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from ganrev.parallel import DeviceTrainer
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dims, nd, B = ((1, 32, 32), 32, 256) if wl == "cfg2" else ((3, 64, 64), 100, 512)
ctxB = L.default_context(); ctxB.set_conv_mode("f16x3")
G = models.create_G(dims, nd); synth.init_params(G, 1)
R = models.create_R(dims, nd, seed=1)
G.evaluate(); G.forward(synth.normal((2, nd), 1))
R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params(); R._net.adam_reset()
tr = DeviceTrainer(ctxB, G._net, R._net, L.Hyper(), B)
ctxA = L.Context(0); ctxA.set_conv_mode("f16x3")
GA = models.create_G(dims, nd); GA._ctx = ctxA; synth.init_params(GA, 1)
GA.evaluate(); GA.forward(synth.normal((2, nd), 1))
noiseA = ctxA.upload(synth.normal((B, nd), 9))
GA._net.set_training(False)
N = 40
def run(step_b, fwd_a):
    for _ in range(5):
        if fwd_a: GA._net.forward_dev(noiseA, B)
        if step_b: tr.new_noise(3); tr.step()
    ctxA.synchronize(); ctxB.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        if fwd_a: GA._net.forward_dev(noiseA, B)
        if step_b: tr.new_noise(100 + i); tr.step()
    ctxA.synchronize(); ctxB.synchronize()
    return (time.perf_counter() - t0) / N * 1e3
for rep in range(3):
    t1, t2, t3 = run(True, False), run(False, True), run(True, True)
    print(f"{wl}: step alone {t1:.3f} ms, G forward alone {t2:.3f} ms, both enqueued together {t3:.3f} ms per pair  (sum {t1 + t2:.3f}; overlap recovered {(t1 + t2 - t3) / t2 * 100:.0f} % of the G forward)", flush=True)
