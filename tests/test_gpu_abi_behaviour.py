"""-m gpu: behaviour of the C ABI beyond numerics — error codes instead of aborts, determinism, noise (Philox) statistics
and read-back, module-level helpers."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_error_codes_not_aborts(ctx):
    import ganrev._lib as L
    lib = ctx.lib
    h = C.c_void_p()
    assert lib.gr_init(9999, C.byref(h)) == -1                               # GR_ERR_INVALID: no such device
    bad = (L.LayerDesc * 2)(L.LayerDesc(L.CONV3, 5, 8, 0, 0.0, 0), L.LayerDesc(L.ELU, 0, 0, 0, 0.0, 0))
    net = C.c_void_p()
    assert lib.gr_net_create(ctx.h, bad, 2, 3, 8, 8, C.byref(net)) == -1      # conv expects 5 planes, input has 3
    assert b"expects 5 input planes" in lib.gr_last_error(ctx.h)
    lone_up = (L.LayerDesc * 1)(L.LayerDesc(L.UPSAMPLE2, 0, 0, 0, 0.0, 0))
    assert lib.gr_net_create(ctx.h, lone_up, 1, 3, 8, 8, C.byref(net)) == -2  # GR_ERR_UNSUPPORTED
    with pytest.raises(L.GanrevError):
        ctx.cosine_topk(np.zeros((4, 3), np.float32), [7], 2)                 # query row out of range
    from ganrev import nn, synth
    m = nn.Sequential().add(nn.SpatialConvolution(2, 4)).add(nn.SpatialBatchNormalization(4))
    with pytest.raises(L.GanrevError):
        m.backward(synth.normal((2, 2, 8, 8), 1), synth.normal((2, 4, 8, 8), 2))     # backward before forward
    m.evaluate(); m.forward(synth.normal((2, 2, 8, 8), 1))
    with pytest.raises(L.GanrevError, match="training mode"):
        m.backward(synth.normal((2, 2, 8, 8), 1), synth.normal((2, 4, 8, 8), 2))     # BN backward in evaluate()
    with pytest.raises(L.GanrevError):
        m.forward(synth.normal((2, 3, 8, 8), 1))                                      # wrong channel count -> new net fails


def test_step_is_deterministic(ctx, conv_mode):
    """No float atomics anywhere: two runs from the same state and seeds are bit-identical (losses, parameters)."""
    import ganrev._lib as L
    from ganrev import models, synth
    from ganrev.parallel import DeviceTrainer
    runs = []
    for _ in range(2):
        G = models.create_G((1, 32, 32), 16); synth.init_params(G, 1)
        R = models.create_R((1, 32, 32), 16); synth.init_params(R, 2)
        G.evaluate(); G.forward(synth.normal((2, 16), 1))
        R.training(); R.forward(synth.uniform((2, 1, 32, 32), 2, 0, 1)); R.push_params()
        R._net.set_seed(77); R._net.adam_reset()
        tr = DeviceTrainer(ctx, G._net, R._net, L.Hyper(), 16)
        losses = []
        for t in range(4):
            tr.new_noise(50 + t); losses.append(tr.step(want_loss=True))
        runs.append((losses, R._net.get_params()))
    assert runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])


def test_philox_noise_statistics_and_readback(ctx):
    from ganrev import models, synth
    R = models.create_R((1, 32, 32), 8); synth.init_params(R, 1)
    R.training(); R.manualSeed(5)
    B = 16
    x = synth.uniform((B, 1, 32, 32), 3, 0, 1)
    out1 = R.forward(x).copy()
    drops = [m for m in R.leaves() if m.typename in ("nn.Dropout", "nn.SpatialDropout")]
    keeps1 = [R.getNoise(m, B) for m in drops]
    for m, k in zip(drops, keeps1):
        rate = k.mean()
        tol = 0.01 if k.size > 10000 else 0.06
        assert abs(rate - (1 - m.p)) < tol, (m.typename, m.p, rate)           # Bernoulli(1-p) keep rate
    out2 = R.forward(x).copy()                                               # fresh noise on the next forward
    keeps2 = [R.getNoise(m, B) for m in drops]
    assert any(not np.array_equal(a, b) for a, b in zip(keeps1, keeps2)) and not np.array_equal(out1, out2)
    # feeding the recorded noise back reproduces the forward bit for bit
    for m, k in zip(drops, keeps2):
        R.setNoise(m, k)
    assert np.array_equal(R.forward(x), out2)
    # evaluate(): dropout off, deterministic
    R.evaluate()
    assert np.array_equal(R.forward(x).copy(), R.forward(x))


def test_fill_normal_statistics(ctx):
    n = 1 << 20
    d = ctx.malloc(4 * n)
    ctx.fill_normal(d, n, 9)
    a = ctx.download(d, (n,))
    assert abs(a.mean()) < 5e-3 and abs(a.std() - 1) < 5e-3 and np.isfinite(a).all()
    ctx.fill_normal(d, n, 10)
    assert not np.array_equal(a, ctx.download(d, (n,)))
    ctx.free(d)


def test_layer_output_and_module_level_calls(ctx, oracle):
    from ganrev import nn, synth
    import ganrev._lib as L
    x = synth.normal((3, 4, 8, 8), 1)
    conv = nn.SpatialConvolution(4, 6); synth.init_params(nn.Sequential().add(conv), 2)
    y = conv.forward(x)                                                      # a leaf module on its own is a one-layer net
    assert np.max(np.abs(y - oracle.conv3_forward(x, conv.weight, conv.bias))) < 1e-4
    for act, ref in ((nn.ReLU(), lambda v: np.maximum(v, 0)), (nn.Tanh(), np.tanh), (nn.LeakyReLU(0.333), lambda v: np.where(v > 0, v, v * np.float32(0.333)))):
        assert np.max(np.abs(act.forward(x) - ref(x))) < 1e-6
    up = nn.Sequential().add(nn.SpatialUpSamplingNearest(2)).add(nn.SpatialConvolution(4, 6))
    up.modules[1].weight[...] = conv.weight; up.modules[1].bias[...] = conv.bias
    yu = up.forward(x)
    assert np.max(np.abs(yu - oracle.conv3_forward(np.repeat(np.repeat(x, 2, 2), 2, 3), conv.weight, conv.bias))) < 1e-4
    seq = nn.Sequential().add(nn.SpatialConvolution(4, 6)).add(nn.SpatialBatchNormalization(6)).add(nn.ELU())
    seq.training(); out = seq.forward(x)
    raw = seq._net.layer_output(0, (3, 6, 8, 8))                             # the conv's own output is materialised
    assert raw.shape == (3, 6, 8, 8) and np.isfinite(raw).all()
    with pytest.raises(L.GanrevError):
        seq._net.layer_output(1, (3, 6, 8, 8))                               # BN output is fused away
    assert np.array_equal(seq._net.layer_output(2, (3, 6, 8, 8)), out)
