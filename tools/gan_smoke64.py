"""Smoke run of the device-resident GAN batch at 64x64 RGB (multi-tile 5x5 planes, 16x16 / 8x8 planes in D's deep tower)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gan-reverser_amd"))
import numpy as np
import ganrev._lib as L
from ganrev import adversarial, models, synth
dims, nd, B = (3, 64, 64), 100, int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = L.default_context()
G = models.create_G(dims, nd, seed=1); synth.init_params(G, 2)
D = models.create_D2(dims, seed=2); synth.init_params(D, 3)
env = adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd)
game = adversarial.DeviceGame(env)
real = synth.uniform((B // 2,) + dims, 40, 0, 1)
for _ in range(3):
    ld, lg = game.batch(real, want_loss=True)
ctx.event_record(61000)
for _ in range(5):
    game.batch(real)
ctx.event_record(61001)
game.sync_to_host()
print("losses", ld, lg, "ms per batch", ctx.event_elapsed_ms(61000, 61001) / 5, "finite", bool(np.isfinite(env.PARAMETERS_D).all() and np.isfinite(env.PARAMETERS_G).all()))
