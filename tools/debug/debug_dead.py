"""Dead-channel case of tests/test_gpu_parity.py::test_dead_channels_between_two_guard_scans_in_the_device_loop in every arithmetic:
per-module gradient error against the oracle (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "gan-reverser_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import ganrev._lib as L
from ganrev import models, synth
from oracle import oracle
from helpers import adopt_device_argmax, inject_noise, param_segments, release_argmax

dims, nd, B = (1, 32, 32), 32, 8
ctx = L.default_context()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-7
for mode in ("f32", "bf16x6", "f16x3"):
    ctx.set_tuning("range_guard", 0)
    ctx.set_conv_mode(mode)
    G = models.create_G(dims, nd); synth.init_params(G, 9)
    R = models.create_R(dims, nd); synth.init_params(R, 10)
    convs = [m for m in R.leaves() if m.typename == "nn.SpatialConvolution"]
    for conv, dead_out, dead_in in ((convs[1], (3, 17, 40), 5), (convs[4], (0, 64, 127), 77)):
        conv.weight[list(dead_out)] *= np.float32(scale)
        conv.weight[:, dead_in] *= np.float32(scale)
    oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
    G.evaluate(); G.forward(synth.normal((2, nd), 1))
    R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params(); R._pending_masks = {}
    gnet, rnet = G._net, R._net
    theta0 = oR.params.copy(); zeros = np.zeros_like(theta0)
    noise = synth.normal((B, nd), 321)
    inject_noise(R, oR, B, 322)
    for module, keep in R._pending_masks.values():
        rnet.set_mask(R._leaf_layer(module), keep)
    R._pending_masks = {}
    rnet.set_params(theta0); rnet.set_adam_state(zeros, zeros)
    dn = ctx.upload(noise)
    loss = L.train_r_step(gnet, rnet, dn, B, B, L.Hyper(l2=0.0, clamp=0.0), 2)
    g = rnet.get_grads()
    oG.set_training(False); rimg = oG.forward(noise)
    oR.set_training(True); oR.zero_grads(); oR.forward(rimg)
    adopt_device_argmax(R, oR, B, 8, mode=mode)
    oR.zero_grads(); preds = np.array(oR.forward(rimg), copy=True)
    rloss, dfdo = oracle.mse(preds, noise)
    oR.backward(rimg, dfdo, want_gin=False); release_argmax(R, oR)
    rg = oR.grads
    print(mode, "loss", loss, rloss)
    for mod, nm, lo, hi in param_segments(R):
        d = np.abs(g[lo:hi] - rg[lo:hi]); mx = np.abs(rg[lo:hi]).max()
        if mod.typename.endswith("SpatialConvolution") and nm == "weight":
            w = d.reshape(mod.weight.shape); r = rg[lo:hi].reshape(mod.weight.shape)
            per_o = w.max(axis=(1, 2, 3)); o = int(per_o.argmax())
            print(f"  {mod.typename}.{nm} {mod.weight.shape}: max|d| {d.max():.3e} max|g| {mx:.3e} rel {d.max() / max(mx, 1e-30):.2e}; worst out-channel {o} (|g| there {np.abs(r[o]).max():.2e})")
