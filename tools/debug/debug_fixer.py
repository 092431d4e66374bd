import sys
sys.path.insert(0, 'gan-reverser_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from oracle import oracle
from helpers import inject_noise, maxdiff
for fixer in (True, False):
    dims, nd, B = (1, 16, 16), 8, 16
    R = models.create_R(dims, nd, "normal", fixer); synth.init_params(R, 3)
    onet = oracle.from_model(R, dims)
    x = synth.uniform((B,) + dims, 5, 0, 1)
    R.training(); onet.set_training(True)
    flat, grads = R.getParameters()
    inject_noise(R, onet, B, 7)
    out = R.forward(x); ref = onet.forward(x)
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = R.backward(x, gy); rgin = onet.backward(x, gy)
    print("fixer", fixer, "out diff", maxdiff(out, ref), "gin diff", maxdiff(gin, rgin), "max gin", np.abs(rgin).max())
    off = 0
    for mod in R.leaves():
        for nm, a in zip(('w','b'), mod.param_arrays()):
            a0, b0 = off, off + a.size; off = b0
            dg = maxdiff(grads[a0:b0], onet.grads[a0:b0]); mx = np.abs(onet.grads[a0:b0]).max()
            print(f"   {mod.typename:32s}.{nm} n={a.size:8d} diff {dg:.3e} max {mx:.3e} rel {dg/max(mx,1e-30):.2e}")
