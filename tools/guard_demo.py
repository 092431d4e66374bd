"""What the f16x3 range guard prevents: the hostile-range convolution of tests/test_gpu_parity.py::test_f16x3_hostile_channel_ranges
(channels spanning 2^-e .. 2^e, every channel contributing equally) with the guard on and off.  Prints the worst error of an
output channel relative to that channel's largest entry.  python tools/guard_demo.py [e]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gan-reverser_amd")); sys.path.insert(0, ROOT)
import ganrev._lib as L
from ganrev import nn, synth
from oracle import oracle
oracle.build()
ctx = L.default_context(); ctx.set_conv_mode("f16x3")
e = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Cin = Cout = 64; B = 4; shape = (64, 16, 16)
for guard in (1, 0):
    ctx.set_tuning("range_guard", guard)
    net = nn.Sequential(); lay = nn.SpatialConvolution(Cin, Cout, 3, 3, 1, 1, 1, 1); net.add(lay)
    synth.init_params(net, 3)
    rng = np.random.default_rng(11)
    s_c = np.exp2(rng.permutation(np.linspace(-e, e, Cin))).astype(np.float32)
    t_o = np.exp2(rng.permutation(np.linspace(-e, e, Cout))).astype(np.float32)
    net.getParameters()
    lay.weight *= (t_o[:, None] / s_c[None, :])[:, :, None, None]; lay.bias *= t_o
    x = synth.normal((B,) + shape, 5) * s_c.reshape(1, -1, 1, 1)
    onet = oracle.from_model(net, shape)
    net.training(); onet.set_training(True)
    ref = onet.forward(x).reshape(B, Cout, 16, 16); out = net.forward(x)
    err = (np.abs(out.astype(np.float64) - ref) / np.abs(ref).max(axis=(0, 2, 3), keepdims=True)).max()
    print(f"e = {e}  range_guard = {guard}: worst output-channel error {err:.3e} of the channel's largest entry; guard stats (scans, fallbacks) = {ctx.range_guard_stats()}")
ctx.set_tuning("range_guard", 1)
