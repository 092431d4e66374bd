"""fused head kernel vs stage-by-stage step: per-quantity differences (diagnostic for tests/test_gpu_parity.py::test_head_kernel_equals_the_stage_by_stage_step)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd")); sys.path.insert(0, ROOT)
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
ctx = L.default_context(); ctx.set_conv_mode(sys.argv[1] if len(sys.argv) > 1 else "f32")
dims, nd, B = (1, 32, 32), 32, 64
G = models.create_G(dims, nd); synth.init_params(G, 1)
R = models.create_R(dims, nd); synth.init_params(R, 2)
G.evaluate(); G.forward(synth.normal((2, nd), 1))
R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
theta0 = R._net.get_params()
dn = ctx.upload(synth.normal((B, nd), 5))
hyper = L.Hyper(l2=0.0, clamp=1e30)
res = []
nl = R._net.lib.gr_net_num_layers(R._net.h) if hasattr(R._net.lib, "gr_net_num_layers") else None
for fused in (1, 0):
    ctx.set_tuning("fused_head", fused)
    R._net.set_params(theta0); R._net.adam_reset(); R._net.set_seed(11)
    loss = L.train_r_step(G._net, R._net, dn, B, B, hyper, 1)
    out = ctx.download(R._net.lib.gr_net_output_dev(R._net.h), (B, nd))
    layers = {}
    lv = R.leaves(); nl_ = len(lv)
    for li in range(nl_ - 5, nl_):
        try:
            layers[li] = (lv[li].typename, R._net.layer_output(li, (B, 512) if li < nl_ - 1 else (B, nd)))
        except Exception as e:  # noqa
            layers[li] = (lv[li].typename + " " + str(e)[:80], None)
    res.append((loss, out, R._net.get_grads(), layers))
(l1, o1, g1, y1), (l0, o0, g0, y0) = res
print("loss", l1, l0, "out diff", np.abs(o1 - o0).max(), "grads diff", np.abs(g1 - g0).max(), np.abs(g0).max())
for li in y1:
    a, b = y1[li][1], y0[li][1]
    if a is not None and b is not None:
        print(li, y1[li][0], a.shape, "diff", float(np.abs(a - b).max()), "max", float(np.abs(b).max()))
off = 0
for m in R.leaves():
    for t in m.param_arrays():
        n_ = t.size
        a, b = g1[off:off + n_], g0[off:off + n_]
        print(m.typename, n_, "grad diff", float(np.abs(a - b).max()), "max", float(np.abs(b).max()))
        off += n_
