"""phase stamps of head_fwd_bwd_kernel (ablation build: GANREV_LIB=.../libganrev_ablate.so): wall-clock (100 MHz) per workgroup at
0 start, 1 end of phase 1, 2 past barrier 1, 3 end of fc2 forward, 4 end of phase 2, 5 past barrier 2, 6 end of phase 3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd")); sys.path.insert(0, ROOT)
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
dims, nd, B = ((1, 32, 32), 32, 256) if len(sys.argv) < 2 or sys.argv[1] == "cfg2" else ((3, 64, 64), 100, 512)
G = models.create_G(dims, nd); synth.init_params(G, 1)
R = models.create_R(dims, nd); synth.init_params(R, 2)
G.evaluate(); G.forward(synth.normal((2, nd), 1))
R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
dn = ctx.upload(synth.normal((B, nd), 5))
st = ctx.malloc(8 * 8 * 64)
ctx.lib.gr_debug_stamps(ctx.h, st)
for i in range(4):
    L.train_r_step(G._net, R._net, dn, B, B, L.Hyper(), i + 1)
ctx.synchronize()
t = ctx.download(st, (64, 8), np.uint64).astype(np.int64)
t0 = t[:, 0].min()
rel = (t - t0) / 100.0          # us at 100 MHz
names = ["start", "ph1 end", "bar1 out", "fc2 fwd", "ph2 end", "bar2 out", "ph3 end"]
for i, nm in enumerate(names):
    print(f"{nm:9s} min {rel[:, i].min():7.2f}  median {np.median(rel[:, i]):7.2f}  max {rel[:, i].max():7.2f} us")
