"""Debug: one cfg2 step at batch B on the GPU vs the oracle, per-parameter-tensor gradient differences and pool near-ties."""
import sys, os
sys.path.insert(0, 'gan-reverser_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from oracle import oracle
from helpers import inject_noise, maxdiff
ctx = L.default_context()
oracle.set_threads(min(32, os.cpu_count()))
dims, nd = (1, 32, 32), 32
for B in [int(a) for a in sys.argv[1:]] or [64, 256]:
    G = models.create_G(dims, nd); synth.init_params(G, 21)
    R = models.create_R(dims, nd); synth.init_params(R, 22)
    oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
    G.evaluate(); G.forward(synth.normal((8, nd), 1))
    R.training(); inject_noise(R, oR, B, 0); R.forward(synth.uniform((B,) + dims, 2, 0, 1))
    gnet, rnet = G._net, R._net
    theta0 = oR.params.copy()
    m = np.zeros(rnet.n_params, np.float32); v = np.zeros_like(m)
    segs = []; off = 0
    for mod in R.leaves():
        for nm, a in zip(('w', 'b'), mod.param_arrays()):
            segs.append((f"{mod.typename}.{nm}", off, off + a.size)); off += a.size
    noise = synth.normal((B, nd), 77); inject_noise(R, oR, B, 78)
    rloss, rimg = oracle.train_r_step(oG, oR, noise, oracle.GoHyper(), m, v, 1, want_images=True)
    rnet.set_params(theta0); rnet.set_adam_state(np.zeros_like(m), np.zeros_like(m))
    dn = ctx.malloc(4 * B * nd); ctx.upload(noise, dn)
    for module, keep in R._pending_masks.values(): rnet.set_mask(R._leaf_layer(module), keep)
    R._pending_masks = {}
    loss = L.train_r_step(gnet, rnet, dn, B, B, L.Hyper(), 1)
    g = rnet.get_grads()
    print(f"B={B} mode={ctx.conv_mode if hasattr(ctx,'conv_mode') else '?'} loss {loss:.8f} ref {rloss:.8f} grad maxdiff {maxdiff(g, oR.grads):.3e} max|g| {np.abs(oR.grads).max():.3e}")
    for nm, a, b in segs:
        dg = maxdiff(g[a:b], oR.grads[a:b]); mx = np.abs(oR.grads[a:b]).max()
        print(f"     {nm:40s} n={b-a:8d} grad diff {dg:.3e} (max {mx:.3e}) rel {dg/max(mx,1e-30):.2e}  n>1e-4rel {(np.abs(g[a:b]-oR.grads[a:b])>1e-4*mx).sum()}")
    # per-layer activations: where do they start to differ?
    for li in range(oR.n_layers):
        try:
            ro = oR.layer_output(li)
            go = rnet.layer_output(li, ro.size) if hasattr(rnet, 'layer_output') else None
        except Exception as e:
            go = None
        if go is not None:
            d = np.abs(go.ravel() - ro.ravel())
            print(f"     layer {li:2d} out n={ro.size:9d} maxdiff {d.max():.3e} (max {np.abs(ro).max():.3e})  n>1e-5: {(d>1e-5).sum()}")
    ctx.free(dn)
