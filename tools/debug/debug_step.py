import sys, os
sys.path.insert(0, 'gan-reverser_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from oracle import oracle
from helpers import inject_noise, maxdiff
ctx = L.default_context()
dims, nd, B = (3, 16, 16), 12, 8
G = models.create_G(dims, nd); synth.init_params(G, 5)
R = models.create_R(dims, nd); synth.init_params(R, 6)
oG, oR = oracle.from_model(G, (nd,1,1)), oracle.from_model(R, dims)
G.evaluate(); G.forward(synth.normal((B, nd), 1))
R.training(); inject_noise(R, oR, B, 0); R.forward(synth.uniform((B,) + dims, 2, 0, 1)); R.push_params()
gnet, rnet = G._net, R._net
m = np.zeros(rnet.n_params, np.float32); v = np.zeros_like(m)
rnet.adam_reset()
dn = ctx.malloc(4*B*nd)
# param segment names
segs = []
off = 0
for mod in R.leaves():
    for nm, a in zip(('w','b'), mod.param_arrays()):
        segs.append((f"{mod.typename}.{nm}", off, off + a.size)); off += a.size
for t in (1,2,3):
    noise = synth.normal((B, nd), 100+t); ctx.upload(noise, dn)
    inject_noise(R, oR, B, 10+t)
    for module, keep in R._pending_masks.values(): rnet.set_mask(R._leaf_layer(module), keep)
    R._pending_masks = {}
    loss = L.train_r_step(gnet, rnet, dn, B, B, L.Hyper(), t)
    rloss, rimg = oracle.train_r_step(oG, oR, noise, oracle.GoHyper(), m, v, t, want_images=True)
    g, th = rnet.get_grads(), rnet.get_params()
    print(f"t={t} loss {loss:.8f} ref {rloss:.8f}  grad maxdiff {maxdiff(g, oR.grads):.3e} (max|g| {np.abs(oR.grads).max():.3g})  param maxdiff {maxdiff(th, oR.params):.3e} frac>1e-4 {np.mean(np.abs(th-oR.params)>1e-4):.5f}")
    for nm, a, b in segs:
        dg = maxdiff(g[a:b], oR.grads[a:b]); dp = maxdiff(th[a:b], oR.params[a:b])
        if dg > 1e-5 or dp > 1e-4:
            print(f"     {nm:40s} n={b-a:8d} grad diff {dg:.3e} (max {np.abs(oR.grads[a:b]).max():.3e})  param diff {dp:.3e}")
