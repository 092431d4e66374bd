#!/usr/bin/env python3
"""Per-parameter-tensor gradient differences vs the oracle for one R case in every arithmetic mode (diagnostic for pooling
near-tie flips: a flip shows up only in tensors upstream of the pooling layer and in every mode-independent position)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from oracle import oracle
from helpers import inject_noise, pools_well_conditioned
dims, nd, B = (3, 64, 64), 100, 2
if len(sys.argv) > 1: dims, nd, B = tuple(int(v) for v in sys.argv[1].split("x")), int(sys.argv[2]), int(sys.argv[3])
gap = float(os.environ.get("GAP", "1e-5"))
ctx = L.default_context()
for mode in ("f32", "bf16x6", "f16x3"):
    ctx.set_conv_mode(mode)
    R = models.create_R(dims, nd, "normal", False); synth.init_params(R, 3)
    flat, grads = R.getParameters()
    onet = oracle.from_model(R, dims)
    R.training(); onet.set_training(True)
    for seed in range(5, 12):
        x = synth.uniform((B,) + dims, seed, 0, 1)
        inject_noise(R, onet, B, seed + 2)
        ref = onet.forward(x)
        if pools_well_conditioned(R, onet, B, gap=gap): break
    out = R.forward(x)
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = R.backward(x, gy); ref_gin = onet.backward(x, gy)
    gmax = float(np.abs(onet.grads).max())
    print(f"== {mode} seed {seed} fwd diff {np.abs(out-ref).max():.2e} gin diff {np.abs(gin-ref_gin).max():.2e} (max {np.abs(ref_gin).max():.2e}) grad diff {np.abs(grads-onet.grads).max():.2e} (max {gmax:.3g})")
    off = 0
    for m in R.leaves():
        for nm in ("weight", "bias"):
            w = getattr(m, nm, None)
            if w is None: continue
            n = w.size
            d = np.abs(grads[off:off+n] - onet.grads[off:off+n]); 
            print(f"   {m.typename:28s} {nm:6s} n={n:9d} max|g| {np.abs(onet.grads[off:off+n]).max():9.3e}  max diff {d.max():9.3e}  n(diff>1e-4*gmax) {(d > 1e-4*gmax).sum()}")
            off += n
