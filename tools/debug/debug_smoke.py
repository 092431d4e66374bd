import sys, os
ROOT="/root/repo"
for p in (os.path.join(ROOT,"gan-reverser_amd"), ROOT, os.path.join(ROOT,"tests")): sys.path.insert(0,p)
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from oracle import oracle
import helpers
ctx = L.default_context()
for mode in ("f32","bf16x6","f16x3"):
    ctx.set_conv_mode(mode)
    dims, nd, B = (1, 32, 32), 32, 4
    G = models.create_G(dims, nd); synth.init_params(G, 1)
    R = models.create_R(dims, nd); synth.init_params(R, 2)
    oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
    G.evaluate(); G.forward(synth.normal((B, nd), 1))
    R.training(); R.forward(synth.uniform((B,) + dims, 2, 0, 1))
    R.push_params()
    noise = synth.normal((B, nd), 3)
    for m in R.leaves():
        if m.typename in ("nn.Dropout", "nn.SpatialDropout"):
            li = oR.layer_index[id(m)]
            keep = synth.bernoulli_keep((oR.mask_size(li, B),), 50 + li, m.p)
            oR.set_mask(li, keep); R._net.set_mask(li, keep)
    R._net.adam_reset()
    dn = ctx.upload(noise)
    loss = L.train_r_step(G._net, R._net, dn, B, B, L.Hyper(), 1)
    m = np.zeros(oR.n_params, np.float32); v = np.zeros_like(m)
    rloss, rimg = oracle.train_r_step(oG, oR, noise, oracle.GoHyper(), m, v, 1, want_images=True)
    img = ctx.download(G._net.lib.gr_net_output_dev(G._net.h), rimg.shape)
    g = R._net.get_grads()
    d = np.abs(g - oR.grads)
    print(mode, "dimg", float(np.abs(img-rimg).max()), "loss", loss, rloss, "dg", float(d.max()), "argmax", int(d.argmax()), "n>1e-4", int((d>1e-4).sum()),
          "well-conditioned pools (gap 2e-6):", helpers.pools_well_conditioned(R, oR, B), " (gap 2e-5):", helpers.pools_well_conditioned(R, oR, B, gap=2e-5))
