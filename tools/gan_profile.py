"""rocprofv3 target: N device-resident GAN batches (ganrev.adversarial.DeviceGame) at one batch size.
usage: rocprofv3 --kernel-trace --stats -d gpurun_out/gan_prof -- python3 tools/gan_profile.py [batch] [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gan-reverser_amd"))
import ganrev._lib as L
from ganrev import adversarial, models, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dims, nd = (1, 32, 32), 100
ctx = L.default_context()
ctx.set_conv_mode(os.environ.get("GR_CONV_MODE", "f16x3"))
G = models.create_G(dims, nd, seed=1); synth.init_params(G, 2)
D = models.create_D2(dims, seed=2); synth.init_params(D, 3)
env = adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd)
game = adversarial.DeviceGame(env)
real = synth.uniform((B // 2,) + dims, 40, 0, 1)
for _ in range(3):
    game.batch(real)
ctx.event_record(61000)
for _ in range(reps):
    game.batch(real)
ctx.event_record(61001)
print("ms per batch", ctx.event_elapsed_ms(61000, 61001) / reps)
