"""Does DeviceGame's parameter scan trip the f16x3 range guard on the synthetic G / D2 of the GAN tests? (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganrev._lib as L
from ganrev import adversarial, models, nn_utils, synth
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
dims, nd, B = (1, 32, 32), 16, 8
G = models.create_G(dims, nd, seed=1); synth.init_params(G, 2)
D = models.create_D2(dims, seed=2); synth.init_params(D, 3)
env = adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd, N_epoch=1, seed=5)
print("before", ctx.conv_mode(), ctx.range_guard_stats())
game = adversarial.DeviceGame(env)
print("after compile", ctx.conv_mode(), ctx.range_guard_stats())
for n in [game.gnet] + game.dg.nets:
    print("  scan ->", n.range_guard_scan(), ctx.conv_mode(), ctx.range_guard_stats())
for m in G.leaves():
    if hasattr(m, "running_mean"):
        g = np.abs(m.weight); print("  BN", m.weight.size, "gamma min/max", g.min(), g.max(), "spread bits", np.log2(g.max() / g[g > 0].min()))
