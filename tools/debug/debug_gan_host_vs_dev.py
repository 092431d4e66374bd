"""test_device_resident_gan_batch_matches_the_host_mirror, per module: where do the host mirror and the device game differ? (diagnostic)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganrev._lib as L
from ganrev import adversarial, models, nn_utils, synth
from helpers import param_segments
ctx = L.default_context()
for mode in (sys.argv[1:] or ["f16x3", "bf16x6"]):
    ctx.set_conv_mode(mode)
    dims, nd, B = (1, 32, 32), 16, 8
    envs = []
    for _ in range(2):
        G = models.create_G(dims, nd, seed=1); synth.init_params(G, 2)
        D = models.create_D2(dims, seed=2); synth.init_params(D, 3)
        envs.append(adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd, N_epoch=1, seed=5))
    host, dev = envs
    game = adversarial.DeviceGame(dev)
    host.MODEL_D.forward(host.MODEL_G.forward(nn_utils.createNoiseInputs(2, nd, "normal", seed=1)))
    real = synth.uniform((B // 2,) + dims, 40, 0, 1)
    noise_d = nn_utils.createNoiseInputs(B // 2, nd, "normal", seed=5 * 100003 + 1)
    noise_g = nn_utils.createNoiseInputs(B, nd, "normal", seed=5 * 100003 + 2)
    pg0 = host.PARAMETERS_G.copy()
    adversarial.train(host, real)
    ld, lg = game.batch(real, noise_d, noise_g, want_loss=True)
    game.sync_to_host()
    print(mode, "losses", ld, lg, host.last_losses["D"][0], host.last_losses["G"][0], ctx.conv_mode(), ctx.range_guard_stats())
    for name, model, a, b in (("G", host.MODEL_G, dev.PARAMETERS_G, host.PARAMETERS_G), ("D", host.MODEL_D, dev.PARAMETERS_D, host.PARAMETERS_D)):
        d = np.abs(a.astype(np.float64) - b)
        print(f" {name}: median {np.median(d):.2e} share>1e-5 {(d > 1e-5).mean():.2e} max {d.max():.2e}")
        off = 0
        for chunk, lo, hi in model._param_chunks():
            for mod, nm, l2, h2 in param_segments(chunk):
                dd = d[lo + l2:lo + h2]
                if dd.size and (dd > 1e-5).mean() > 1e-3:
                    print(f"    {mod.typename}.{nm} [{lo + l2}:{lo + h2}] share>1e-5 {(dd > 1e-5).mean():.2e} max {dd.max():.2e}")
