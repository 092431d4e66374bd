"""-m gpu: the HIP path (through the C ABI, libganrev.so) against the CPU oracle on identical inputs.

Tolerance 1e-4 absolute on fp32 outputs (BASELINE.json north_star); integer results (top-k indices, pool argmax via
gradients) bit-exact; element-wise fp32 kernels (Adam) bit-exact given identical inputs."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL, assert_close, inject_noise, maxdiff, rel_close

pytestmark = pytest.mark.gpu


def _dev(ctx, a):
    return ctx.upload(np.ascontiguousarray(a, np.float32))


CONV_SHAPES = [
    # B, Cin, Cout, H, W
    (2, 64, 64, 32, 32),     # R.conv2/3 (cfg2 shape, small batch)
    (2, 1, 64, 32, 32),      # R.conv1 gray
    (2, 3, 64, 16, 16),      # R.conv1 RGB
    (2, 64, 128, 16, 16),    # R.conv4
    (2, 128, 128, 16, 16),   # R.conv5/6
    (3, 128, 3, 32, 32),     # G.convC RGB
    (2, 128, 1, 32, 32),     # G.convC gray
    (2, 24, 3, 40, 40),      # few output channels on a plane that is neither small nor a multiple of 64 wide: the generic 32-row tiles of conv3x3_fewout_kernel (ragged last tile column and row)
    (1, 16, 2, 96, 96),      # ... with 3 x 3 whole tiles
    (2, 16, 40, 8, 8),       # ragged: Cout not a multiple of 32
    (9, 32, 40, 8, 8),       # 8x8 planes, four images stacked per tile (conv3x3_split_kernel<8, 1, *, 4>), ragged last tile, ragged Cout
    (8, 48, 128, 8, 8),      # ... and the 64-channel workgroups (<8, 2, *, 4>): the D network's deep tower (models.lua:304-317)
    (2, 20, 64, 4, 4),       # tiny spatial (tile mostly masked)
    (1, 8, 32, 64, 64),      # W = 64 (two column tiles)
    (2, 12, 32, 24, 20),     # non power-of-two H, W
    (257, 16, 128, 16, 16),  # enough workgroups for the stacked two-image 512-pixel tile; odd batch -> half-empty last tile
    (64, 8, 128, 32, 32),    # enough workgroups for the 512-pixel tile on 32-wide planes
    (32, 8, 64, 64, 64),     # 512-pixel tiles, two column tiles per row (cfg3 geometry)
    (150, 16, 64, 32, 32),   # 300 tiles on 256 persistent workgroups: 44 of them walk two tiles
    (37, 8, 128, 64, 64),    # 592 tiles (two channel blocks per pixel tile): two or three tiles per workgroup
]


@pytest.mark.parametrize("B,Cin,Cout,H,W", CONV_SHAPES)
def test_conv3_kernels_vs_oracle(ctx, oracle, conv_mode, B, Cin, Cout, H, W):
    from ganrev import synth
    x = synth.normal((B, Cin, H, W), 11)
    w = synth.uniform((Cout, Cin, 3, 3), 12, -1, 1) / np.float32(np.sqrt(Cin * 9))
    b = synth.uniform((Cout,), 13, -0.5, 0.5)
    gy = synth.normal((B, Cout, H, W), 14)
    dx, dw, db, dgy = _dev(ctx, x), _dev(ctx, w), _dev(ctx, b), _dev(ctx, gy)
    dout = ctx.malloc(4 * B * Cout * H * W)
    lib = ctx.lib
    if H == 8 and W == 8 and B >= 4:
        ctx.set_tuning("stack8_min_wgs", 1)          # four 8x8 images per tile on these small grids too (the library stacks from 128 workgroups on)
    try:
        ctx.check(lib.gr_conv3_forward_dev(ctx.h, dx, dw, db, dout, B, Cin, Cout, H, W, 0), "fwd")
        y = ctx.download(dout, (B, Cout, H, W))
        assert_close(y, oracle.conv3_forward(x, w, b), TOL, "conv forward")
        # backward-data
        dgin = ctx.malloc(4 * B * Cin * H * W)
        ctx.check(lib.gr_conv3_backward_data_dev(ctx.h, dgy, dw, dgin, B, Cin, Cout, H, W), "bwd-data")
    finally:
        ctx.set_tuning("stack8_min_wgs", 128)
    gin = ctx.download(dgin, (B, Cin, H, W))
    assert_close(gin, oracle.conv3_backward_data(gy, w), TOL * 3, "conv backward-data")
    # backward-weight (accumulating)
    gw0 = synth.uniform((Cout, Cin, 3, 3), 15, -1, 1)
    dgw = _dev(ctx, gw0)
    ctx.check(lib.gr_conv3_backward_weight_dev(ctx.h, dx, dgy, dgw, B, Cin, Cout, H, W), "bwd-weight")
    ctx.synchronize()
    gw = ctx.download(dgw, (Cout, Cin, 3, 3))
    ref, _ = oracle.conv3_backward_weight(x, gy)
    rel_close(gw - gw0, ref, 2e-5, "conv backward-weight")
    for p in (dx, dw, db, dgy, dout, dgin, dgw):
        ctx.free(p)


def test_f16x3_scale_tracking_is_exact_and_range_safe(ctx, oracle):
    """f16x3 puts every operand tensor into fp16 range with a power-of-two scale taken from its device-tracked maximum and
    scales the result back with ldexp.  Consequences checked here, on forward / backward-data / backward-weight and on the
    up-sampling kernel: (1) multiplying the inputs by powers of two - 2^-60, 2^40: far outside fp16's exponent range -
    changes the output by exactly that power of two, bit for bit; (2) one element 2^20 times larger than the rest (every other
    element's low term goes subnormal) keeps fp32-level accuracy relative to the largest output, and 1e-4 absolute away from it."""
    from ganrev import synth
    prev = ctx.conv_mode()
    ctx.set_conv_mode("f16x3")
    try:
        B, Cin, Cout, H, W = 4, 64, 64, 16, 16
        x = synth.normal((B, Cin, H, W), 91)
        w = synth.uniform((Cout, Cin, 3, 3), 92, -1, 1) / np.float32(np.sqrt(Cin * 9))
        gy = synth.normal((B, Cout, H, W), 93)
        xs = synth.normal((B, Cin, H // 2, W // 2), 94)
        lib = ctx.lib

        def run(xv, wv, gv, xsv):
            dx, dw, dg, dxs = _dev(ctx, xv), _dev(ctx, wv), _dev(ctx, gv), _dev(ctx, xsv)
            dout, dgin = ctx.malloc(4 * B * Cout * H * W), ctx.malloc(4 * B * Cin * H * W)
            dgw = _dev(ctx, np.zeros((Cout, Cin, 3, 3), np.float32))
            ctx.check(lib.gr_conv3_forward_dev(ctx.h, dx, dw, None, dout, B, Cin, Cout, H, W, 0), "fwd")
            y = ctx.download(dout, (B, Cout, H, W))
            ctx.check(lib.gr_conv3_forward_dev(ctx.h, dxs, dw, None, dout, B, Cin, Cout, H, W, 1), "fwd-up")
            yu = ctx.download(dout, (B, Cout, H, W))
            ctx.check(lib.gr_conv3_backward_data_dev(ctx.h, dg, dw, dgin, B, Cin, Cout, H, W), "bwd-data")
            gin = ctx.download(dgin, (B, Cin, H, W))
            ctx.check(lib.gr_conv3_backward_weight_dev(ctx.h, dx, dg, dgw, B, Cin, Cout, H, W), "bwd-weight")
            ctx.synchronize()
            gw = ctx.download(dgw, (Cout, Cin, 3, 3))
            for q in (dx, dw, dg, dxs, dout, dgin, dgw):
                ctx.free(q)
            return y, yu, gin, gw

        base = run(x, w, gy, xs)
        for kx, kw, kg in ((-60, 20, 30), (40, -50, -45)):
            sx, sw, sg = np.float32(2.0) ** kx, np.float32(2.0) ** kw, np.float32(2.0) ** kg
            y, yu, gin, gw = run(x * sx, w * sw, gy * sg, xs * sx)
            assert np.array_equal(y, base[0] * (sx * sw)), "forward is not exactly scale-equivariant"
            assert np.array_equal(yu, base[1] * (sx * sw)), "up-sampling forward is not exactly scale-equivariant"
            assert np.array_equal(gin, base[2] * (sg * sw)), "backward-data is not exactly scale-equivariant"
            assert np.array_equal(gw, base[3] * (sx * sg)), "backward-weight is not exactly scale-equivariant"
        xo = x.copy(); xo[1, 3, 5, 7] = np.float32(2.0 ** 20)
        y = run(xo, w, gy, xs)[0]
        import torch
        ref = torch.nn.functional.conv2d(torch.from_numpy(xo).double(), torch.from_numpy(w).double(), padding=1).numpy()
        top = float(np.abs(ref).max())
        # fp32 accumulation next to a 4e4 term rounds at ulp(4e4) = 4e-3 per add (the fp32 oracle is off by as much):
        # the bar is fp32-level error relative to the largest output, and outputs away from the outlier keep their own accuracy
        assert maxdiff(y, ref) <= 1e-5 * top
        far = np.ones_like(ref, bool); far[1, :, 3:8, 5:10] = False
        assert maxdiff(y[far], ref[far]) <= 1e-4
    finally:
        ctx.set_conv_mode(prev)


# source planes 8x8 (eight stacked images per tile, odd batch -> a partly empty tile), 16x16 (two stacked images), 24x40 (ragged
# 32-wide tiles, two column tiles), channels not a multiple of the 16-channel chunk / the 32-channel block
@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 32, 64, 16, 16), (11, 20, 40, 16, 16), (3, 48, 33, 32, 32), (2, 16, 64, 48, 80)])
def test_conv3_upsample_fused(ctx, oracle, conv_mode, B, Cin, Cout, H, W):
    """SpatialUpSamplingNearest(2) + SpatialConvolution (models.lua:121-122,127-128).  In f16x3 mode this is the four-2x2-
    convolutions kernel; the other modes fold the up-sampling into the 3x3 kernel's input addressing."""
    from ganrev import synth
    xs = synth.normal((B, Cin, H // 2, W // 2), 21)
    w = synth.uniform((Cout, Cin, 3, 3), 22, -0.1, 0.1)
    b = synth.uniform((Cout,), 23)
    dx, dw, db = _dev(ctx, xs), _dev(ctx, w), _dev(ctx, b)
    dout = ctx.malloc(4 * B * Cout * H * W)
    ctx.check(ctx.lib.gr_conv3_forward_dev(ctx.h, dx, dw, db, dout, B, Cin, Cout, H, W, 1), "fwd-up")
    y = ctx.download(dout, (B, Cout, H, W))
    xu = np.repeat(np.repeat(xs, 2, axis=2), 2, axis=3)
    assert_close(y, oracle.conv3_forward(xu, w, b), TOL, "conv forward with fused upsample")


R_CASES = [((1, 8, 8), 6, 8, "normal", False), ((1, 32, 32), 32, 8, "normal", False),
           ((3, 16, 16), 10, 6, "uniform", False), ((1, 16, 16), 8, 16, "normal", True),
           ((2, 12, 20), 5, 3, "normal", False),      # ragged: H, W not powers of two, odd batch -> scalar / non-vector fallbacks
           # cfg3 geometry (two column tiles per row).  Batch 4, not 2: nn.BatchNormalization over two samples normalises them
           # to exactly +-1 and its backward amplifies rounding noise ~50x (exact-fp32 mode: 7e-5 of the largest gradient entry)
           ((3, 64, 64), 100, 4, "normal", False)]


@pytest.mark.parametrize("dims,nd,B,method,fixer", R_CASES)
def test_R_forward_backward_vs_oracle(oracle, conv_mode, dims, nd, B, method, fixer):
    from ganrev import models, synth
    from helpers import pools_well_conditioned
    R = models.create_R(dims, nd, method, fixer)
    synth.init_params(R, 3)
    flat, grads = R.getParameters()
    onet = oracle.from_model(R, dims)
    R.training()
    onet.set_training(True)
    from helpers import GAPS
    found = False
    for gap in GAPS:                    # skip inputs whose pooling has a rounding-level near-tie (see helpers)
        for seed in range(5, 12):
            x = synth.uniform((B,) + dims, seed, 0, 1)
            inject_noise(R, onet, B, seed + 2)
            ref = onet.forward(x)
            if pools_well_conditioned(R, onet, B, gap):
                found = True
                break
        if found:
            break
    out = R.forward(x)
    assert_close(out, ref, TOL, "R forward (training)")
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0
    onet.zero_grads()
    gin = R.backward(x, gy)
    ref_gin = onet.backward(x, gy)
    # gradients are not O(1) quantities (tiny batches make BN gradients large and ill-conditioned): 1e-4 relative to
    # the largest entry, 1e-4 absolute when that is below 1
    gmax = float(np.abs(onet.grads).max())
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "R gradInput")
    assert_close(grads, onet.grads, 2 * TOL * max(1.0, gmax), f"R flat gradient (max |g| = {gmax:.3g})")


@pytest.mark.parametrize("dims,nd,B", [((1, 32, 32), 8, 29), ((1, 32, 32), 8, 50), ((3, 64, 64), 16, 72), ((1, 16, 16), 8, 305)])
def test_R_batch_sizes_with_uneven_channel_slices(oracle, conv_mode, dims, nd, B):
    """The float4 pipeline kernels (BatchNorm statistics, pipeline backward passes A and B) slice the BATCH over up to 64
    workgroups per channel, `per = ceil(B / splits)` images each.  At these batch sizes (splits - 1) * per exceeds B (29 images
    in 14 slices of 3: slices 10..13 would start past the batch), which used to wrap an unsigned count and read far out of
    bounds.  Training-mode forward, gradInput and every gradient tensor against the oracle (device argmax adopted)."""
    from ganrev import models, synth
    from helpers import adopt_device_argmax, assert_grads_close
    R = models.create_R(dims, nd)
    synth.init_params(R, 3)
    flat, grads = R.getParameters()
    onet = oracle.from_model(R, dims)
    R.training(); onet.set_training(True)
    x = synth.uniform((B,) + dims, 5, 0, 1)
    inject_noise(R, onet, B, 7)
    ref = onet.forward(x)
    out = R.forward(x)
    assert_close(out, ref, TOL, "R forward (training)")
    adopt_device_argmax(R, onet, B, 8)
    ref = onet.forward(x)
    assert_close(out, ref, TOL, "R forward vs the argmax-forced oracle")
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = R.backward(x, gy)
    ref_gin = onet.backward(x, gy)
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "R gradInput")
    assert_grads_close(R, grads, onet.grads, 1e-4, 1e-3)


@pytest.mark.parametrize("dims,nd,B", [((1, 32, 32), 8, 8), ((1, 32, 32), 8, 5), ((3, 64, 64), 16, 3)])
def test_R_operand_ready_path_vs_oracle(oracle, f16_path, dims, nd, B):
    """f16x3 with the operand-ready (P16) pipeline forced onto small shapes (conftest.f16_path): the pipeline kernels of a
    stage write the next convolution's input - and pass B of the backward the data-gradient convolution's input - as
    [8-channel group][term][pixel] fp16 hi/lo vectors scaled by an A-PRIORI bound of the tensor's maximum (BatchNorm
    statistics + max|y|; K * max|dz|), and conv3x3_p16_wide_kernel stages them by LDS-DMA.  32x32 planes (one image = two
    tiles), 16x16 planes (two stacked images per tile; odd batch: a half-empty tile), 64x64 (two column tiles).  Same bars as
    every other R case; the default-selection run of the same shapes is the control."""
    from ganrev import models, synth
    from helpers import adopt_device_argmax, assert_grads_close
    R = models.create_R(dims, nd)
    synth.init_params(R, 3)
    flat, grads = R.getParameters()
    onet = oracle.from_model(R, dims)
    R.training(); onet.set_training(True)
    x = synth.uniform((B,) + dims, 5, 0, 1)
    inject_noise(R, onet, B, 7)
    ref = onet.forward(x)
    out = R.forward(x)
    assert_close(out, ref, TOL, "R forward (training)")
    adopt_device_argmax(R, onet, B, 8)
    ref = onet.forward(x)
    assert_close(out, ref, TOL, "R forward vs the argmax-forced oracle")
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = R.backward(x, gy)
    ref_gin = onet.backward(x, gy)
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "R gradInput")
    assert_grads_close(R, grads, onet.grads, 1e-4, 1e-3)


@pytest.mark.parametrize("B,Cin,Cmid,Cout,H,W", [(6, 16, 24, 20, 12, 12), (4, 64, 64, 32, 32, 32), (3, 8, 40, 3, 16, 16)])
def test_fullconv_bn_leakyrelu_dropout_stage_vs_oracle(oracle, conv_mode, B, Cin, Cmid, Cout, H, W):
    """BASELINE.json north_star names nn.SpatialFullConvolution and a SpatialBatchNormalization + LeakyReLU + Dropout epilogue
    (the live reference has neither: models.lua:121-122 up-samples with UpSamplingNearest + SpatialConvolution, LeakyReLU(0.333)
    only appears in the never-called models.lua:8-55).  Two such stages back to back - FullConv(3,3,1,1,1,1) -> SBN ->
    LeakyReLU(0.333) -> Dropout - in training mode (forward, gradInput, every gradient tensor: the weight [nIn][nOut][3][3],
    bias, BN) and in evaluate() mode (BN + LeakyReLU ride in the convolution's epilogue), all three arithmetics."""
    from ganrev import nn, synth
    from helpers import adopt_device_kinks, assert_grads_close
    net = nn.Sequential()
    net.add(nn.SpatialFullConvolution(Cin, Cmid, 3, 3, 1, 1, 1, 1)); net.add(nn.SpatialBatchNormalization(Cmid)); net.add(nn.LeakyReLU(0.333)); net.add(nn.Dropout(0.5))
    net.add(nn.SpatialFullConvolution(Cmid, Cout, 3, 3, 1, 1, 1, 1)); net.add(nn.SpatialBatchNormalization(Cout)); net.add(nn.LeakyReLU(0.333)); net.add(nn.Dropout(0.5))
    synth.init_params(net, 23)
    flat, grads = net.getParameters()
    onet = oracle.from_model(net, (Cin, H, W))
    x = synth.normal((B, Cin, H, W), 31)
    net.training(); onet.set_training(True)
    inject_noise(net, onet, B, 5)
    ref = onet.forward(x)
    out = net.forward(x)
    assert_close(out, ref, TOL * max(1.0, float(np.abs(ref).max())), "forward (training)")
    adopt_device_kinks(net, onet, B, 8)      # a LeakyReLU input within rounding of zero: use the side the device took
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = net.backward(x, gy)
    ref_gin = onet.backward(x, gy)
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "gradInput")
    assert_grads_close(net, grads, onet.grads, 1e-4, 1e-3)
    net.evaluate(); onet.set_training(False)
    ref_e = onet.forward(x)
    assert_close(net.forward(x), ref_e, TOL * max(1.0, float(np.abs(ref_e).max())), "forward (evaluate)")


@pytest.mark.parametrize("dims,nd,B", [((1, 16, 16), 8, 6), ((3, 32, 32), 16, 4)])
def test_G_training_forward_backward_vs_oracle(oracle, conv_mode, dims, nd, B):
    """First slice of the GAN step (adversarial.lua:37-205 trains G through D): MODEL_G:forward / :backward in TRAINING mode -
    batch-statistics BatchNorm, and the backward of the fused SpatialUpSamplingNearest(2) + SpatialConvolution stages
    (models.lua:121-122,127-128): weight gradient against the up-sampled input, data gradient folded back over 2x2 blocks.
    Images, gradInput (w.r.t. the noise) and every gradient tensor of G against the oracle, three arithmetics."""
    from ganrev import models, synth
    from helpers import adopt_device_kinks, assert_grads_close
    G = models.create_G(dims, nd); synth.init_params(G, 2)
    flat, grads = G.getParameters()
    onet = oracle.from_model(G, (nd, 1, 1))
    z = synth.normal((B, nd), 8)
    G.training(); onet.set_training(True)
    ref = onet.forward(z)
    img = G.forward(z)
    assert_close(img, ref, TOL, "G images (training mode)")
    adopt_device_kinks(G, onet, B, 8)        # a ReLU input within rounding of zero: use the side the device took
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = G.backward(z, gy)
    ref_gin = onet.backward(z, gy)
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "gradInput (noise)")
    assert_grads_close(G, grads, onet.grads, 1e-4, 1e-3)


@pytest.mark.parametrize("kind", ["conv", "fullconv", "linear"])
@pytest.mark.parametrize("e", [4, 20])
def test_f16x3_hostile_channel_ranges(ctx, oracle, kind, e):
    """f16x3 scales a tensor by one power of two, so hostile per-channel ranges are where it could lose digits: x[:, c] *= s_c,
    W[o][c] *= t_o / s_c, bias_o *= t_o, gradOutput[:, o] /= t_o with s, t spanning 2^-e .. 2^e - every channel still
    contributes equally to every output, so a channel that lost its digits shows.  Forward, gradInput and the weight
    gradient are held to 1e-4 of the largest reference entry OF THEIR OWN CHANNEL (weight gradient: of the natural size of
    their own (out, in) pair).  e = 4: the activation-side and the weight-side spread add up to 18 bits, inside the range guard's 20-bit budget - the f16x3
    kernels run and must meet the bar.  e = 20 (channels spanning 2^-20 .. 2^20): the guard must send both passes to
    bf16x6 (gr_range_guard_stats counts them) and the same bar holds."""
    from ganrev import nn, synth
    prev = ctx.conv_mode(); ctx.set_conv_mode("f16x3")
    try:
        if kind == "linear":
            Cin, Cout, B, shape = 1024, 1024, 64, (1024,)
            net = nn.Sequential(); lay = nn.Linear(Cin, Cout); net.add(lay)
        else:
            Cin, Cout, B, shape = 64, 64, 4, (64, 16, 16)
            net = nn.Sequential()
            lay = (nn.SpatialFullConvolution if kind == "fullconv" else nn.SpatialConvolution)(Cin, Cout, 3, 3, 1, 1, 1, 1); net.add(lay)
        synth.init_params(net, 3)
        rng = np.random.default_rng(11)
        s_c = np.exp2(rng.permutation(np.linspace(-e, e, Cin))).astype(np.float32)
        t_o = np.exp2(rng.permutation(np.linspace(-e, e, Cout))).astype(np.float32)
        flat, grads = net.getParameters()
        w = lay.weight
        if kind == "linear": w *= t_o[:, None] / s_c[None, :]
        elif kind == "fullconv": w *= (t_o[None, :] / s_c[:, None])[:, :, None, None]
        else: w *= (t_o[:, None] / s_c[None, :])[:, :, None, None]
        lay.bias *= t_o
        bshape = (1, -1) + (1,) * (len(shape) - 1)
        x = synth.normal((B,) + shape, 5) * s_c.reshape(bshape)
        onet = oracle.from_model(net, shape if len(shape) == 3 else (shape[0], 1, 1))
        net.training(); onet.set_training(True)
        scans0, falls0 = ctx.range_guard_stats()
        ref = onet.forward(x)
        out = net.forward(x)
        gy = (synth.normal(ref.shape, 9) / t_o.reshape(bshape)).astype(np.float32)
        grads[...] = 0; onet.zero_grads()
        gin = net.backward(x, gy)
        ref_gin = onet.backward(x, gy)
        scans1, falls1 = ctx.range_guard_stats()
        assert scans1 > scans0, "the range guard did not look at this pass"
        assert falls1 - falls0 == (2 if e == 20 else 0), f"range guard sent {falls1 - falls0} passes to bf16x6"

        def per_channel(a, r, axis_keep, what):
            a = np.asarray(a, np.float64); r = np.asarray(r, np.float64)
            red = tuple(i for i in range(r.ndim) if i not in axis_keep)
            scale = np.abs(r).max(axis=red, keepdims=True)
            err = np.abs(a - r) / np.maximum(scale, 1e-300)
            assert err.max() <= TOL, f"{kind} e={e} {what}: {err.max():.3e} of the channel's largest entry"
        per_channel(out, ref.reshape(out.shape), (1,), "output, per output channel")
        per_channel(gin, ref_gin.reshape(gin.shape), (1,), "gradInput, per input channel")
        nW = w.size
        gw_dev, gw_ref = grads[:nW].reshape(w.shape), onet.grads[:nW].reshape(w.shape)
        # an entry of the weight gradient is a sum of N = B * H * W products gradOutput_o * x_c of random sign: its own value can
        # cancel to anything, its natural size is rms(gradOutput_o) * rms(x_c) * sqrt(N) - the bar is 1e-4 of THAT, per (o, c)
        red = tuple(i for i in range(x.ndim) if i != 1)
        g_rms = np.sqrt((gy.astype(np.float64) ** 2).mean(axis=red)); x_rms = np.sqrt((x.astype(np.float64) ** 2).mean(axis=red))
        nat = np.outer(x_rms, g_rms) if kind == "fullconv" else np.outer(g_rms, x_rms)
        nat = nat * np.sqrt(x.size / Cin)
        err = np.abs(gw_dev.astype(np.float64) - gw_ref) / nat.reshape(nat.shape + (1,) * (gw_ref.ndim - 2))
        assert err.max() <= TOL, f"{kind} e={e} weight gradient: {err.max():.3e} of the (out, in) pair's natural size"
    finally:
        ctx.set_conv_mode(prev)


def test_range_guard_trips_in_the_device_resident_loop(ctx):
    """gr_train_r_step cannot synchronise, so it samples: parameter scans on the step's own stream every 64th step, verdict read
    by a later call.  A BatchNorm whose gammas span 2^-12 .. 2^12 (24 bits of spread > the 20-bit budget) must move the
    context to bf16x6 within a few steps and be counted once; a well-conditioned R must not trip it."""
    import ganrev._lib as L
    from ganrev import models, synth
    from ganrev.parallel import DeviceTrainer
    prev = ctx.conv_mode(); ctx.set_conv_mode("f16x3")
    try:
        dims, nd, B = (1, 16, 16), 8, 8
        for hostile in (False, True):
            G = models.create_G(dims, nd); synth.init_params(G, 1)
            R = models.create_R(dims, nd); synth.init_params(R, 2)
            if hostile:
                bn = [m for m in R.leaves() if m.typename == "nn.SpatialBatchNormalization"][1]
                bn.weight[...] = np.exp2(np.linspace(-12, 12, bn.weight.size)).astype(np.float32)
            G.evaluate(); G.forward(synth.normal((2, nd), 1))
            R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
            ctx.set_conv_mode("f16x3")
            _, falls0 = ctx.range_guard_stats()
            tr = DeviceTrainer(ctx, G._net, R._net, L.Hyper(), B)
            for i in range(4):
                tr.new_noise(i + 1); tr.step(); ctx.synchronize()
            _, falls1 = ctx.range_guard_stats()
            if hostile:
                assert ctx.conv_mode() == "bf16x6" and falls1 - falls0 == 1, (ctx.conv_mode(), falls1 - falls0)
            else:
                assert ctx.conv_mode() == "f16x3" and falls1 == falls0
    finally:
        ctx.set_tuning("range_guard", 0); ctx.set_tuning("range_guard", 1)     # clears the tripped state
        ctx.set_conv_mode(prev)


def test_default_initialised_nets_stay_on_f16x3(ctx, tmp_path):
    """Torch's BatchNorm reset draws gamma ~ U(0, 1): over 64-512 channels that is 8-14 bits of spread per layer, and until round 3
    the guard added the two largest spreads of ANY two tensors - two of five default-initialised R nets (seeds 1 and 3 here) went
    to bf16x6 at their first step for nothing.  The guard adds what a kernel multiplies (largest activation-side + largest
    weight-side spread): train_r's loop on default-initialised nets must stay on f16x3, no pass sent to bf16x6."""
    from ganrev import models, train_r
    for seed in (1, 3):
        gam = [m.weight for m in models.create_R((1, 32, 32), 32, "normal", False, seed=seed).leaves() if "BatchNorm" in m.typename]   # what train_r.main builds
        spread = sorted(float(np.log2(np.abs(g).max() / np.abs(g)[g != 0].min())) for g in gam)
        assert spread[-1] + spread[-2] > 20, "this seed no longer has the spreads the old rule tripped on"
        _, falls0 = ctx.range_guard_stats()
        _, R, losses = train_r.main(["--nbBatches", "3", "--batchSize", "8", "--quiet", "--height", "32", "--width", "32", "--channels", "1",
                                     "--seed", str(seed), "--save", str(tmp_path / f"r{seed}.net")])
        assert ctx.conv_mode() == "f16x3" and ctx.range_guard_stats()[1] == falls0, (seed, ctx.conv_mode(), ctx.range_guard_stats())
        assert len(losses) == 3 and np.all(np.isfinite(losses))


def test_dead_channels_between_two_guard_scans_in_the_device_loop(ctx, oracle, conv_mode):
    """the device-resident loop scans the parameters only every 64th gr_train_r_step and never the
    activations.  The case that could slip through between two scans is a TRAINED-LOOKING weight tensor with dead channels: a few
    output channels (and one input channel's column) of R's convolutions at |w| ~ 1e-7 next to channels at ~1 - 23 bits of
    spread, over the 20-bit budget; f16x3 then keeps only ~17 bits of the dead channels' weights.  What saves the result is the
    reference's own BatchNorm: var + 1e-5 in the denominator caps the gain of a channel whose outputs are ~1e-7 at 316, so what a
    dead channel's lost bits change downstream is below 1e-10.  Asserted: (1) a step that runs with such weights BETWEEN two scans
    (t = 2: no scan due; in f16x3 mode the context must still be on f16x3 afterwards) meets every parity bar against the oracle -
    images, recovered noise, loss, and every RAW gradient tensor at 1e-4 of its module's largest entry (measured on MI355X: 4-7e-6
    in f32, bf16x6 and f16x3 alike - tools/debug/debug_dead.py; the dead channels' own gradients are ~35, amplified by that 316, so the
    comparison is made before the clamp, which would otherwise hide them all at +-1); (2) f16x3 only: the next due scan (t = 65)
    sees the spread and moves the context to bf16x6, counted once."""
    import ganrev._lib as L
    from ganrev import synth
    from helpers import adopt_device_argmax, assert_grads_close, release_argmax
    dims, nd, B = (1, 32, 32), 32, 8
    try:
        G, R, oG, oR = _make_pair(oracle, dims, nd, 9)
        convs = [m for m in R.leaves() if m.typename == "nn.SpatialConvolution"]
        for conv, dead_out, dead_in in ((convs[1], (3, 17, 40), 5), (convs[4], (0, 64, 127), 77)):
            conv.weight[list(dead_out)] *= np.float32(1e-7)          # output channels that died in training
            conv.weight[:, dead_in] *= np.float32(1e-7)              # an input channel nothing listens to any more
        oR = oracle.from_model(R, dims)
        G.evaluate(); G.forward(synth.normal((2, nd), 1))
        R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
        R._pending_masks = {}
        ctx.set_tuning("range_guard", 0); ctx.set_tuning("range_guard", 1); ctx.set_conv_mode(conv_mode)   # (the compile forwards above were host calls: guarded)
        gnet, rnet = G._net, R._net
        theta0 = oR.params.copy()
        zeros = np.zeros_like(theta0)
        noise = synth.normal((B, nd), 321)
        inject_noise(R, oR, B, 322)
        for module, keep in R._pending_masks.values():
            rnet.set_mask(R._leaf_layer(module), keep)
        R._pending_masks = {}
        rnet.set_params(theta0); rnet.set_adam_state(zeros, zeros)
        _, falls0 = ctx.range_guard_stats()
        dn = ctx.upload(noise)
        t = 2                                                          # t % 64 != 1: no parameter scan is due in this call
        free = L.Hyper(l1=0.0, l2=0.0, clamp=0.0)                      # the step leaves the raw gradient readable
        loss = L.train_r_step(gnet, rnet, dn, B, B, free, t)
        assert ctx.conv_mode() == conv_mode and ctx.range_guard_stats()[1] == falls0, "the step under test changed arithmetic"
        img = ctx.download(gnet.lib.gr_net_output_dev(gnet.h), (B,) + dims)
        rec = ctx.download(rnet.lib.gr_net_output_dev(rnet.h), (B, nd))
        g = rnet.get_grads()
        # the oracle on the same state (argmax adopted as everywhere else)
        oG.set_training(False); rimg = oG.forward(noise)
        oR.set_training(True); oR.zero_grads()
        preds = np.array(oR.forward(rimg), copy=True)
        assert_close(img, rimg, TOL, "G images"); assert_close(rec, preds, TOL, "recovered noise with dead channels, between scans")
        flips = adopt_device_argmax(R, oR, B, 8)
        oR.zero_grads(); preds = np.array(oR.forward(rimg), copy=True)
        rloss, dfdo = oracle.mse(preds, noise)
        oR.backward(rimg, dfdo, want_gin=False); release_argmax(R, oR)
        assert abs(loss - rloss) <= 1e-5 * max(1.0, abs(rloss))
        assert_grads_close(R, g, oR.grads, 1e-4, 1e-3, f"(dead channels, {conv_mode}, argmax flips {flips})")
        if conv_mode == "f16x3":                                       # (2) the next due scan catches it
            rnet.set_params(theta0)
            for t2 in (65, 66, 67):
                L.train_r_step(gnet, rnet, dn, B, B, L.Hyper(), t2); ctx.synchronize()
            assert ctx.conv_mode() == "bf16x6" and ctx.range_guard_stats()[1] - falls0 == 1, (ctx.conv_mode(), ctx.range_guard_stats())
        ctx.free(dn)
    finally:
        ctx.set_tuning("range_guard", 0); ctx.set_tuning("range_guard", 1)     # clears the tripped state


@pytest.mark.parametrize("dims,nd,B,method", [((1, 32, 32), 32, 64, "normal"), ((1, 32, 32), 32, 37, "uniform"), ((1, 32, 32), 32, 256, "normal")])
def test_head_kernel_equals_the_stage_by_stage_step(ctx, conv_mode, dims, nd, B, method):
    """gr_train_r_step runs R's last two stages, the criterion and their backward (models.lua:446-454, train_r.lua:147-151) in ONE launch (head_fwd_bwd_kernel,
    gr_set_tuning "fused_head" 1 = default) instead of 14.  Same operations per value, only the order of the sums differs: one step from the same state with
    the kernel on and off must agree - loss to 1e-6 relative, recovered noise to 1e-6, every raw gradient tensor to 1e-4 of its module's largest entry (a 1e-7 difference in
    fc1's dy passes through six BatchNorm backwards on its way to conv1) - and the kernel must really have run (and not run when switched off).  The oracle comparison of the
    step itself is test_train_r_steps_vs_oracle / test_full_size_step_vs_oracle (which now run the head kernel)."""
    import ganrev._lib as L
    from ganrev import models, synth
    G = models.create_G(dims, nd); synth.init_params(G, 1)
    R = models.create_R(dims, nd, noiseMethod=method); synth.init_params(R, 2)
    G.evaluate(); G.forward(synth.normal((2, nd), 1))
    R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
    theta0 = R._net.get_params()
    noise = synth.normal((B, nd), 5) if method == "normal" else synth.uniform((B, nd), 5, -1, 1)
    dn = ctx.upload(noise)
    hyper = L.Hyper(l2=0.0, clamp=1e30)            # raw gradients: penalty and clamp would hide differences
    res = []
    try:
        for fused in (1, 0):
            ctx.set_tuning("fused_head", fused)
            R._net.set_params(theta0); R._net.adam_reset(); R._net.set_seed(11)
            ctx.set_timing(2)
            loss = L.train_r_step(G._net, R._net, dn, B, B, hyper, 1)
            names = {k["kernel"] for k in ctx.kernel_times()}
            ctx.set_timing(0)
            assert ("head_fwd_bwd_kernel" in names) == bool(fused), sorted(names)
            out = ctx.download(R._net.lib.gr_net_output_dev(R._net.h), (B, nd))
            res.append((loss, out, R._net.get_grads(), R._net.get_params()))
        (l1, o1, g1, p1), (l0, o0, g0, p0) = res
        assert abs(l1 - l0) <= 1e-6 * max(1.0, abs(l0)), (l1, l0)
        assert np.abs(o1 - o0).max() <= 1e-6, np.abs(o1 - o0).max()
        off = 0
        for m in R.leaves():
            sizes = [t.size for t in m.param_arrays()]
            if not sizes:
                continue
            scale = max(float(np.abs(g0[off:off + sum(sizes)]).max()), 1e-12)      # the module's largest entry (a bias in front of a BatchNorm holds rounding residue only)
            for n_ in sizes:
                a, b = g1[off:off + n_], g0[off:off + n_]
                assert np.abs(a - b).max() <= 1e-4 * scale, (m.typename, n_, float(np.abs(a - b).max()), scale)
                off += n_
        assert off == g0.size
    finally:
        ctx.set_tuning("fused_head", 1)
        ctx.free(dn)


def test_side_stream_weight_gradients_change_nothing(ctx):
    """gr_set_tuning "side_wgrad" 1 runs R's convolution weight gradients on a second stream beside the rest of backward (dy
    double-buffered, events both ways).  Same kernels, same operands, same order inside each kernel: three training steps
    from the same state must leave bit-identical parameters and gradients with the knob on and off (cfg2 geometry, f16x3)."""
    import ganrev._lib as L
    from ganrev import models, synth
    from ganrev.parallel import DeviceTrainer
    prev = ctx.conv_mode(); ctx.set_conv_mode("f16x3")
    try:
        dims, nd, B = (1, 32, 32), 32, 64
        G = models.create_G(dims, nd); synth.init_params(G, 1)
        R = models.create_R(dims, nd); synth.init_params(R, 2)
        G.evaluate(); G.forward(synth.normal((2, nd), 1))
        R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
        theta0 = R._net.get_params()
        res = []
        # (third arm, round 5: side stream AND a one-rank RCCL communicator - at cfg3's sizes the side stream is on by default, so under data parallelism the
        #  bucketed all-reduce on the comm stream has to wait for weight gradients that finish on the side stream)
        for side, comm in ((0, False), (1, False), (1, True)):
            ctx.set_tuning("side_wgrad", side)
            R._net.set_params(theta0); R._net.adam_reset(); R._net.set_seed(7)
            if comm:
                ctx.comm_init(ctx.comm_unique_id(), 1, 0)
            try:
                tr = DeviceTrainer(ctx, G._net, R._net, L.Hyper(), B)
                for i in range(3):
                    tr.new_noise(i + 1); tr.step()
                ctx.synchronize()
                res.append((R._net.get_params(), R._net.get_grads()))
            finally:
                if comm:
                    ctx.comm_destroy()
        for other in res[1:]:
            assert np.array_equal(res[0][0], other[0]) and np.array_equal(res[0][1], other[1])
    finally:
        ctx.set_tuning("side_wgrad", -1)     # the library default: by stage size
        ctx.set_conv_mode(prev)


def test_operand_ready_kernels_are_selected(ctx):
    """At the benchmark geometry (cfg2: batch 256) the f16x3 training step must take the operand-ready kernels: R's five
    512-pixel-tile forward convolutions and at least four of its data-gradient convolutions run as conv3x3_p16_*_kernel, fed by
    post_forward_g8_kernel / post_backward_b_g8_kernel; the kernel table of a step says so."""
    import ganrev._lib as L
    from ganrev import models, synth
    from ganrev.parallel import DeviceTrainer
    prev = ctx.conv_mode(); ctx.set_conv_mode("f16x3")
    try:
        dims, nd, B = (1, 32, 32), 32, 256
        G = models.create_G(dims, nd); synth.init_params(G, 1)
        R = models.create_R(dims, nd); synth.init_params(R, 2)
        G.evaluate(); G.forward(synth.normal((2, nd), 1))
        R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
        tr = DeviceTrainer(ctx, G._net, R._net, L.Hyper(), B)
        tr.new_noise(1); tr.step()
        ctx.set_timing(2)
        tr.new_noise(2); tr.step()
        kt = ctx.kernel_times(); ctx.set_timing(0)
        count = lambda prefix: sum(k["launches"] for k in kt if k["kernel"].startswith(prefix))
        assert count("conv3x3_p16_") >= 9, sorted((k["kernel"], k["launches"]) for k in kt if k["kernel"].startswith("conv3x3"))
        assert count("post_forward_g8_kernel") == 5 and count("post_backward_b_g8_kernel") == 5
        assert count("conv3x3_wgrad_p16_") == 5, "R.conv2 .. conv6: weight gradients from the operand-ready x and dy images (plain or ping-pong kernel)"
        assert count("conv3x3_split_wide_kernel") == 0
    finally:
        ctx.set_conv_mode(prev)


@pytest.mark.parametrize("B,nin,nmid,nout", [(130, 1030, 1100, 37), (257, 2052, 640, 129), (128, 1024, 1024, 128)])
def test_large_linear_ragged_shapes_vs_oracle(oracle, conv_mode, B, nin, nmid, nout):
    """nn.Linear layers big enough for the f16x3 GEMM (>= 2^20 weights) at sizes that are not multiples of anything: partial
    128-row tiles in M and N, a K tail (1030 = 32 * 32 + 6, not a multiple of 4: scalar tile loads), split-K and unsplit
    plans, training-mode BatchNormalization after the first layer (models.lua:447-451 pattern), evaluate()-mode epilogue
    (models.lua:115-117 pattern).  Forward, gradInput and the flat gradient against the oracle."""
    from ganrev import nn, synth
    net = nn.Sequential()
    net.add(nn.Linear(nin, nmid)); net.add(nn.BatchNormalization(nmid)); net.add(nn.ReLU())
    net.add(nn.Linear(nmid, nout))
    synth.init_params(net, 17)
    flat, grads = net.getParameters()
    onet = oracle.from_model(net, (nin, 1, 1))
    x = synth.normal((B, nin), 5)
    net.training(); onet.set_training(True)
    ref = onet.forward(x)
    out = net.forward(x)
    assert_close(out, ref, TOL * max(1.0, float(np.abs(ref).max())), "forward (training)")
    gy = synth.normal(ref.shape, 9) * np.float32(0.1)
    grads[...] = 0; onet.zero_grads()
    gin = net.backward(x, gy)
    ref_gin = onet.backward(x, gy)
    gmax = float(np.abs(onet.grads).max())
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "gradInput")
    assert_close(grads, onet.grads, 2 * TOL * max(1.0, gmax), "flat gradient")
    net.evaluate(); onet.set_training(False)
    ref_e = onet.forward(x)
    assert_close(net.forward(x), ref_e, TOL * max(1.0, float(np.abs(ref_e).max())), "forward (evaluate)")


def test_R_eval_forward_and_running_stats(oracle, conv_mode):
    from ganrev import models, synth
    dims, nd, B = (1, 16, 16), 8, 6
    R = models.create_R(dims, nd)
    synth.init_params(R, 4)
    onet = oracle.from_model(R, dims)
    x = synth.uniform((B,) + dims, 6, 0, 1)
    R.training(); onet.set_training(True)
    inject_noise(R, onet, B, 3)
    assert_close(R.forward(x), onet.forward(x), TOL, "R forward (training)")
    R.pull_params()
    bi = 0
    for m in R.leaves():
        if hasattr(m, "running_mean"):
            rm, rv = onet.bn_running(bi)
            assert_close(m.running_mean, rm, 1e-5, f"running_mean[{bi}]")
            assert_close(m.running_var, rv, 1e-5, f"running_var[{bi}]")
            bi += 1
    R.evaluate(); onet.set_training(False)
    assert_close(R.forward(x), onet.forward(x), TOL, "R forward (evaluate)")


@pytest.mark.parametrize("dims,nd,B", [((1, 32, 32), 32, 4), ((3, 16, 16), 10, 3), ((3, 64, 64), 100, 2)])
def test_G_forward_vs_oracle(oracle, conv_mode, dims, nd, B):
    from ganrev import models, synth
    G = models.create_G(dims, nd)
    synth.init_params(G, 2)
    onet = oracle.from_model(G, (nd, 1, 1))
    z = synth.normal((B, nd), 8)
    G.evaluate(); onet.set_training(False)
    img = G.forward(z)
    assert img.shape == (B,) + dims
    assert_close(img, onet.forward(z), TOL, "G images")


def test_mse_and_adam_bit_exact(ctx, oracle):
    from ganrev import synth
    import ganrev._lib as L
    x, t = synth.normal((16, 32), 1), synth.normal((16, 32), 2)
    loss, g = ctx.mse(x, t)
    rl, rg = oracle.mse(x, t)
    assert abs(loss - rl) <= 1e-12 * max(1, abs(rl))
    assert np.array_equal(g, rg), "MSE gradient must be bit-exact"
    # Adam on a one-layer net's flat vector
    from ganrev import nn
    lin = nn.Linear(300, 70)
    xin = synth.normal((4, 300), 3)
    lin.forward(xin)
    net = lin._net
    n = net.n_params
    theta = synth.normal((n,), 4) * np.float32(0.05)
    m = synth.normal((n,), 5) * np.float32(0.01)
    v = np.abs(synth.normal((n,), 6)) * np.float32(1e-3)
    for hyp in (dict(), dict(l1=1e-5, l2=1e-4, clamp=0.5), dict(l1=0, l2=0, clamp=0)):
        for t_ in (1, 2, 1000):
            g = synth.normal((n,), 7 + t_) * np.float32(0.7)
            net.set_params(theta); net.set_grads(g); net.set_adam_state(m, v)
            net.adam_step(L.Hyper(**hyp), t_)
            th2, (m2, v2) = net.get_params(), net.adam_state()
            rt, rg_, rm, rv = theta.copy(), g.copy(), m.copy(), v.copy()
            oracle.penalty_clamp_adam(rt, rg_, rm, rv, oracle.GoHyper(**hyp), t_)
            assert np.array_equal(th2, rt) and np.array_equal(m2, rm) and np.array_equal(v2, rv), f"Adam not bit-exact {hyp} t={t_}"


def _make_pair(oracle, dims, nd, seed):
    from ganrev import models, synth
    G = models.create_G(dims, nd); synth.init_params(G, seed)
    R = models.create_R(dims, nd); synth.init_params(R, seed + 1)
    return G, R, oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)


_STEP_SEEDS = {}


@pytest.mark.parametrize("dims,nd,B", [((1, 32, 32), 32, 8), ((3, 16, 16), 12, 8)])
def test_train_r_steps_vs_oracle(ctx, oracle, conv_mode, dims, nd, B):
    """train_r.lua:138-170, three iterations.  Each iteration starts from the ORACLE's state (theta, m, v): Adam's
    normalised update turns rounding-level gradient differences on entries whose gradient is itself rounding noise
    (every conv bias under BatchNorm has an exactly-zero true gradient) into +-lr parameter differences, and a 1e-3
    parameter difference can flip a max-pool argmax, so free-running trajectories of two correct implementations
    diverge.  Per-step quantities from identical state are well-conditioned and are what is compared."""
    import ganrev._lib as L
    from ganrev import synth
    from helpers import pools_well_conditioned
    G, R, oG, oR = _make_pair(oracle, dims, nd, 5)
    G.evaluate(); G.forward(synth.normal((B, nd), 1))
    R.training(); inject_noise(R, oR, B, 0); R.forward(synth.uniform((B,) + dims, 2, 0, 1))
    gnet, rnet = G._net, R._net
    m = np.zeros(rnet.n_params, np.float32); v = np.zeros_like(m)
    hyper, ohyper = L.Hyper(), oracle.GoHyper()
    dn = ctx.malloc(4 * B * nd)
    seed = 100
    for t in (1, 2, 3):
        theta0, m0, v0 = oR.params.copy(), m.copy(), v.copy()
        # the search below runs on the oracle alone, so the batch it settles on does not depend on the arithmetic mode under test:
        # the seed found by the first mode is reused by the others (40 oracle steps per search were most of this test's time)
        found = _STEP_SEEDS.get((dims, nd, B, t))
        for k in ([39] if found else range(40)):     # pick a batch without pooling near-ties (helpers): wide gap first
            seed = found if found else seed + 1
            noise = synth.normal((B, nd), seed)
            inject_noise(R, oR, B, seed)
            oR.params[...] = theta0; m[...] = m0; v[...] = v0
            rloss, rimg = oracle.train_r_step(oG, oR, noise, ohyper, m, v, t, want_images=True)
            if pools_well_conditioned(R, oR, B, 1e-5 if k < 30 else 2e-6):
                _STEP_SEEDS[(dims, nd, B, t)] = seed
                break
        else:
            raise AssertionError("no batch without a max-pool near-tie among 40 seeds: the conditioning filter is broken")
        rnet.set_params(theta0); rnet.set_adam_state(m0, v0)
        ctx.upload(noise, dn)
        for module, keep in R._pending_masks.values():
            rnet.set_mask(R._leaf_layer(module), keep)
        R._pending_masks = {}
        loss = L.train_r_step(gnet, rnet, dn, B, B, hyper, t)
        img = ctx.download(gnet.lib.gr_net_output_dev(gnet.h), rimg.shape)
        assert_close(img, rimg, TOL, f"G images step {t}")
        assert abs(loss - rloss) <= 1e-5 * max(1.0, abs(rloss)), f"loss step {t}: {loss} vs {rloss}"
        g = rnet.get_grads()                          # penalised + clamped gradient, as fevalR returns it
        assert_close(g, oR.grads, 2e-5, f"clamped gradient step {t}")
        theta = rnet.get_params()
        well = np.abs(oR.grads) > 1e-4                # entries whose Adam step is well-conditioned
        assert_close(theta[well], oR.params[well], TOL, f"parameters after Adam, step {t}")
        assert maxdiff(theta, oR.params) <= 2.1e-3    # the rest moved by at most one lr-sized step either way
        m2, v2 = rnet.adam_state()
        assert_close(m2, m, 1e-5, "adam m"); assert_close(v2, v, 1e-5, "adam v")


_FULL_CACHE = {}


def _full_size_case(oracle, dims, nd, B):
    """Models, oracle twins and the oracle's own ("natural") forward of one full-size train_r iteration; computed once and
    shared by the three arithmetic modes (it does not depend on them)."""
    import os
    from ganrev import synth
    key = (dims, nd, B)
    if key not in _FULL_CACHE:
        # (cfg2 and cfg3 both stay resident: the modes alternate between them, and rebuilding cfg3's G forward on the host cores
        # for every mode was 30 s of the suite; the lean oracle net of cfg3 holds about 6 GB of host memory)
        oracle.set_threads(max(1, min(32, os.cpu_count() or 1)))
        G, R, oG, oR = _make_pair(oracle, dims, nd, 21)
        G.evaluate(); G.forward(synth.normal((8, nd), 1))
        R.training(); inject_noise(R, oR, 8, 0); R.forward(synth.uniform((8,) + dims, 2, 0, 1))
        noise = synth.normal((B, nd), 77)
        oG.set_training(False)
        rimg = oG.forward(noise)                     # train_r.lua:139
        del oG
        oR.set_lean(True)
        oR.set_training(True)
        _FULL_CACHE[key] = dict(G=G, R=R, oR=oR, noise=noise, rimg=rimg, theta0=oR.params.copy())
    return _FULL_CACHE[key]


# allowed argmax flips per pool layer: 4 x the documented counts (helpers.DOCUMENTED_FLIPS: 1-5 at cfg2, 100-110 at cfg3), and
# never more than round 2 allowed (256 at cfg3)
FULL_STEP_SIZES = [pytest.param((1, 32, 32), 32, 256, 20, id="cfg2"), pytest.param((3, 64, 64), 100, 512, 256, id="cfg3")]


@pytest.mark.parametrize("dims,nd,B,max_flips", FULL_STEP_SIZES)
def test_full_size_step_vs_oracle(ctx, oracle, conv_mode, dims, nd, B, max_flips):
    """BASELINE.json configs[1] (32x32 grayscale, noise 32, batch 256) and configs[2] (64x64 RGB, noise 100, batch 512) at
    their full sizes: one train_r.lua:138-170 iteration on the GPU against the oracle run on the host cores.  G images,
    recovered noise and loss at the north-star tolerance against the oracle's own forward.  The gradient - ALL of R's
    parameter tensors - at 2e-4 of its module's largest entry: the pool argmax the device took is read back
    (gr_net_get_pool_index), the windows where it differs from the oracle's must be a handful of rounding-level near-ties
    (gap below 2 x the forward error MEASURED in this run at that pool's inputs - helpers.pool_input_error - capped at the 1e-4
    forward tolerance; 6.5M / 52M windows: measured 1-5 / 100-110 flips, at most 4 x that is accepted and the counts are printed),
    and the oracle's backward is run with the device's argmax (helpers.adopt_device_argmax).
    Why 2e-4 and not 1e-4 at this size (the small cases hold 1e-4): BatchNorm's backward makes sum(dy) vanish per channel in
    exact arithmetic; in fp32 a residue of ~1e-7 |dy| per element survives on either side, and the first convolution's weight
    gradient multiplies it with NON-NEGATIVE pixels summed over B*H*W = 2.1M positions, where the signal itself (random signs)
    only grows like the square root: 1e-7 * sqrt(2.1M) = 1.4e-4 of the result.  The exact-fp32 mode measures 1.09e-4 there."""
    import ganrev._lib as L
    from helpers import adopt_device_argmax, assert_grads_close, release_argmax
    case = _full_size_case(oracle, dims, nd, B)
    G, R, oR, noise, rimg, theta0 = (case[k] for k in ("G", "R", "oR", "noise", "rimg", "theta0"))
    gnet, rnet = G._net, R._net
    release_argmax(R, oR)
    inject_noise(R, oR, B, 78)
    oR.zero_grads()
    preds = oR.forward(rimg)                          # the oracle's own forward: train_r.lua:146
    rloss, _ = oracle.mse(preds, noise)
    rnet.set_params(theta0); rnet.set_adam_state(np.zeros_like(theta0), np.zeros_like(theta0))
    dn = ctx.upload(noise)
    for module, keep in R._pending_masks.values():
        rnet.set_mask(R._leaf_layer(module), keep)
    R._pending_masks = {}
    loss = L.train_r_step(gnet, rnet, dn, B, B, L.Hyper(), 1)
    ctx.free(dn)
    img = ctx.download(gnet.lib.gr_net_output_dev(gnet.h), rimg.shape)
    assert_close(img, rimg, TOL, "G images, full batch")
    rec = ctx.download(rnet.lib.gr_net_output_dev(rnet.h), (B, nd))
    assert_close(rec, preds, TOL, "recovered noise, full batch")
    assert abs(loss - rloss) <= 1e-5 * max(1.0, abs(rloss)), f"loss {loss} vs {rloss}"
    flips = adopt_device_argmax(R, oR, B, max_flips)
    # the oracle again, routed through the device's argmax: forward (the few re-routed windows change its activations by
    # < 1e-5), criterion, backward, penalty + clamp + Adam
    oR.zero_grads()
    preds_f = oR.forward(rimg)
    assert_close(rec, preds_f, TOL, "recovered noise vs the argmax-forced oracle")
    _, dfdo = oracle.mse(preds_f, noise)
    oR.backward(rimg, dfdo, want_gin=False)           # train_r.lua:151
    release_argmax(R, oR)
    rg, rtheta = oR.grads.copy(), theta0.copy()
    rm, rv = np.zeros_like(rg), np.zeros_like(rg)
    oracle.penalty_clamp_adam(rtheta, rg, rm, rv, oracle.GoHyper(), 1)    # :153-170
    g, theta = rnet.get_grads(), rnet.get_params()
    assert_grads_close(R, g, rg, 2e-4, 1e-3, f"(argmax flips {flips})")
    well = np.abs(rg) > 1e-4                          # entries whose Adam step is well-conditioned
    assert_close(theta[well], rtheta[well], TOL, "parameters after Adam")
    assert maxdiff(theta, rtheta) <= 2.1e-3           # nothing moved by more than one lr-sized Adam step either way
    m2, v2 = rnet.adam_state()
    assert_close(m2, rm, 1e-5, "adam m"); assert_close(v2, rv, 1e-5, "adam v")


FULL_SIZES = [pytest.param((1, 32, 32), 32, 256, id="cfg2"), pytest.param((3, 64, 64), 100, 512, id="cfg3")]


@pytest.mark.parametrize("dims,nd,B", FULL_SIZES + [pytest.param((1, 32, 32), 32, 200, id="cfg2-ragged-batch")])
def test_full_size_backward_is_linear(ctx, conv_mode, dims, nd, B):
    """Size-independent property at the full sizes of BASELINE.json configs[1] and configs[2]: with the forward state fixed (batch statistics,
    dropout masks, pool argmax) nn.Sequential:backward is linear in gradOutput, for gradInput and for every gradParameter
    (train_r.lua:150-152).  Exercises the full-size launch geometry of every backward kernel - data gradients, weight
    gradients with their split-and-reduce, BatchNorm and bias reductions - with no oracle and no pooling ambiguity."""
    from ganrev import models, synth
    R = models.create_R(dims, nd); synth.init_params(R, 31)
    x = synth.uniform((B,) + dims, 32, 0, 1)
    R.training(); R.manualSeed(5)
    R.forward(x)
    _, grads = R.getParameters()

    def backward(go):
        R.zeroGradParameters()
        gi = R.backward(x, go).copy()
        return gi, grads.copy()

    u, v = synth.normal((B, nd), 33), synth.normal((B, nd), 34)
    a, b = np.float32(0.75), np.float32(-1.5)
    gi_u, gp_u = backward(u)
    gi_v, gp_v = backward(v)
    gi_w, gp_w = backward(a * u + b * v)
    for what, got, ref in (("gradInput", gi_w, a * gi_u + b * gi_v), ("gradParameters", gp_w, a * gp_u + b * gp_v)):
        scale = float(np.abs(ref).max())
        assert scale > 0
        # f16x3 carries 22-bit operands under data-dependent power-of-two scales (a*u + b*v has its own): measured 2.2e-5
        lin_tol = 4e-5 if conv_mode == "f16x3" else 2e-5
        assert maxdiff(got, ref) <= lin_tol * scale, f"{what}: backward not linear at full size ({maxdiff(got, ref)} vs max {scale})"
    gi_u2, gp_u2 = backward(u)                        # and deterministic: the same bits on a second run
    assert np.array_equal(gi_u, gi_u2) and np.array_equal(gp_u, gp_u2)


@pytest.mark.parametrize("dims,nd,B", FULL_SIZES + [pytest.param((1, 32, 32), 32, 200, id="cfg2-ragged-batch")])   # 200: a partial 128-row GEMM tile
def test_full_size_forward_is_batch_consistent(ctx, oracle, conv_mode, dims, nd, B):
    """In evaluate() mode every sample is independent (running statistics, no dropout), so rows of a full-size batch must
    equal the same rows pushed through as a small batch - bit for bit through the conv stack, whose per-pixel accumulation
    order does not depend on the batch tiling, and to fp32 reordering noise after nn.Linear, whose split-K plan follows the
    batch size - and the small batch is checked against the oracle.  Covers G (train_r.lua:139) and R's
    evaluate() forward (apply_r.lua:120-140) at the full sizes of configs[1] and configs[2]."""
    from ganrev import models, synth
    G = models.create_G(dims, nd); synth.init_params(G, 41)
    R = models.create_R(dims, nd); synth.init_params(R, 42)
    oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
    G.evaluate(); R.evaluate(); oG.set_training(False); oR.set_training(False)
    z = synth.normal((B, nd), 43)
    img = G.forward(z).copy()
    rec = R.forward(img).copy()
    rows = np.array([0, 1, B // 2, B - 1])
    img_s = G.forward(z[rows]).copy()
    rec_s = R.forward(img[rows]).copy()
    assert np.array_equal(img[rows], img_s), "G: full-batch rows differ from the same rows as a small batch"
    assert_close(rec[rows], rec_s, 1e-5, "R: full-batch rows vs the same rows as a small batch")
    assert_close(img_s, oG.forward(z[rows]), TOL, "G images vs oracle")
    assert_close(rec_s, oR.forward(img[rows]), TOL, "recovered noise vs oracle")


def test_cosine_similarity_and_topk_bit_exact(ctx, oracle):
    from ganrev import synth
    N, d, k = 10000, 32, 100
    emb = synth.normal((N, d), 42)
    emb[777] = emb[100]            # exact duplicate of a needle -> score tie, resolved by index
    emb[5000] = 0                  # zero row: score 0 against everything
    q = np.array([100, 200, 300, 400, 500], dtype=np.int64)      # apply_r.lua:267  face_i_idx = i*100
    for accf in (False, True):
        idx, sc = ctx.cosine_topk(emb, q, k, accumulate_in_float=accf)
        ridx, rsc = oracle.cosine_topk(emb, q, k, accumulate_in_float=accf)
        assert np.array_equal(idx, ridx), f"top-{k} indices differ (accf={accf})"
        assert np.array_equal(sc, rsc), "scores must be bit-exact"
    idx50, sc50 = ctx.cosine_topk(emb, q, 50)                   # BASELINE.json: top-50 exact match
    r50, rs50 = oracle.cosine_topk(emb, q, 50)
    assert np.array_equal(idx50, r50) and np.array_equal(sc50, rs50)
    idx100, _ = ctx.cosine_topk(emb, q, 100)
    assert np.array_equal(idx50, idx100[:, :50]), "top-50 must be the prefix of top-100 (strict total order: score desc, index asc)"
    assert ctx.cosine_similarity(emb[1], emb[2]) == oracle.cosine_similarity(emb[1], emb[2])


def test_cosine_topk_edges(ctx, oracle):
    from ganrev import synth
    # ragged sizes: N not a multiple of the tile, d not a multiple of the staging width, k == N, single row
    for (N, d, k, qs) in [(1, 5, 1, [0]), (37, 100, 37, [0, 36]), (2049, 33, 50, [5, 2048]), (4500, 7, 1024, [1, 2, 3, 4, 5, 6, 7, 8, 9])]:
        emb = synth.normal((N, d), N + d)
        q = np.array(qs, dtype=np.int64)
        idx, sc = ctx.cosine_topk(emb, q, k)
        ridx, rsc = oracle.cosine_topk(emb, q, k)
        assert np.array_equal(idx, ridx), (N, d, k)
        assert np.array_equal(sc, rsc), (N, d, k)
    # pixel-wise variant of apply_r.lua:308-314 (d = C*H*W)
    imgs = synth.uniform((300, 3 * 32 * 32), 9, 0, 1)
    idx, sc = ctx.cosine_topk(imgs, np.array([100, 200]), 100)
    ridx, rsc = oracle.cosine_topk(imgs, np.array([100, 200]), 100)
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)


def test_full_size_cfg5_search_bit_exact(ctx, oracle):
    """BASELINE.json configs[4] at its full size: 1M x 100-d embeddings, top-50 for the face_i_idx = i*100 needles of
    apply_r.lua:266-293.  The oracle's scan of 1M rows takes about a second per needle on one core, so the full-size case is
    compared directly: indices and scores bit-exact, plus the order/tie properties that hold at any size."""
    import os
    from ganrev import synth
    N, d, k = 1_000_000, 100, 50
    oracle.set_threads(max(1, min(32, os.cpu_count() or 1)))
    emb = synth.normal((N, d), 4242)
    emb[123456] = emb[100]                    # exact duplicate of a needle: a score tie resolved towards the lower index
    emb[999_999] = emb[300] * np.float32(2)   # parallel vector in the very last row: cosine 1 up to rounding
    q = np.array([100, 200, 300, 999_900], dtype=np.int64)
    idx, sc = ctx.cosine_topk(emb, q, k)
    ridx, rsc = oracle.cosine_topk(emb, q, k)
    assert np.array_equal(idx, ridx), "top-50 indices differ at 1M rows"
    assert np.array_equal(sc, rsc), "top-50 scores differ at 1M rows"
    assert np.all(sc[:, :-1] >= sc[:, 1:]), "scores must be sorted descending"
    tie = sc[:, :-1] == sc[:, 1:]
    assert np.all(idx[:, :-1][tie] < idx[:, 1:][tie]), "ties are ordered by ascending index"
    assert set(idx[0, :2]) == {100, 123456} and idx[0, 0] == 100
    assert 999_999 in idx[2, :2]


def test_search_small_needle_path_on_a_fresh_context(oracle):
    """the five-needle path carves its candidate lists (8 x 256 lists of 96 entries per needle, ~1.58 MB) out of the key area of the
    workspace, which below ~197 K rows used to be SMALLER than the lists - out-of-bounds device writes on a context whose workspace no bigger search had grown.
    A fresh context per table, d = 100 / 32 / 128, Q = 5 (and 8, the most lists), k = 50: bit-exact against the oracle and no rerun."""
    import os
    import ganrev._lib as L
    from ganrev import synth
    oracle.set_threads(max(1, min(32, os.cpu_count() or 1)))
    for N, d, Q, k in ((131_072, 100, 5, 50), (160_000, 100, 5, 50), (131_073, 32, 8, 128), (140_001, 128, 8, 50)):
        c = L.Context(0)
        try:
            emb = synth.normal((N, d), N % 1000 + d)
            q = (np.arange(Q, dtype=np.int64) * 100 + 100) % N
            q[-1] = N - 1
            r0 = c.search_reruns()
            idx, sc = c.cosine_topk(emb, q, k)
            ridx, rsc = oracle.cosine_topk(emb, q, k)
            assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc), (N, d, Q, k)
            assert c.search_reruns() == r0
            # device-resident table, the polled five-needle call (gr_cosine_topk_dev) on the same fresh context
            idx2, sc2 = c.cosine_topk(emb, q[:5], k)
            assert np.array_equal(idx2, ridx[:5]) and np.array_equal(sc2, rsc[:5])
        finally:
            c.close()


def test_bce_criterion_vs_oracle(ctx, oracle):
    """nn.BCECriterion (train.lua:173's CRITERION, used by adversarial.lua): loss to 1e-12 relative (the device's log), gradInput bit-exact (IEEE +, -, x, /
    in double on both sides), through the criterion class the reference scripts use."""
    from ganrev import nn, synth
    for n, seed in ((1, 1), (37, 2), (4096, 3)):
        x = synth.uniform((n,), seed, 0.001, 0.999); t = (synth.uniform((n,), seed + 9, 0, 1) < 0.5).astype(np.float32)
        if n > 2: x[:2] = (0.0, 1.0); t[:2] = (0.0, 1.0)
        crit = nn.BCECriterion()
        loss = crit.forward(x, t); g = crit.backward(x, t)
        rl, rg = oracle.bce(x, t)
        assert abs(loss - rl) <= 1e-12 * max(1.0, abs(rl)), (n, loss, rl)
        assert np.array_equal(g, rg), f"BCE gradient must be bit-exact (n = {n})"


def test_search_1024_needles_bit_exact(ctx, oracle):
    """Q = 1024 needles in one call (128 passes of 8 needles over the table, the needle groups ragged at the end of the list:
    1021 is prime): indices and scores bit-exact against the oracle, on a table small enough for the unfiltered path and
    on one large enough for the sample-bound filter."""
    import os
    from ganrev import synth
    oracle.set_threads(max(1, min(32, os.cpu_count() or 1)))
    for N, d, Q, k in ((20_000, 32, 1024, 50), (140_000, 32, 1021, 50)):
        emb = synth.normal((N, d), 77 + d)
        q = (np.arange(Q, dtype=np.int64) * 19 + 3) % N
        idx, sc = ctx.cosine_topk(emb, q, k)
        ridx, rsc = oracle.cosine_topk(emb, q, k)
        assert np.array_equal(idx, ridx), (N, Q)
        assert np.array_equal(sc, rsc), (N, Q)
        assert np.array_equal(idx[:, 0], q), "every needle is its own best match"


def test_search_batched_mfma_path_bit_exact_and_its_rerun(ctx, oracle):
    """Q >= 32 needles on a table of >= 2^17 rows take the batched path: approximate cosines on the fp16 MFMA (bf16 until round 6) select candidates
    (two cuts with a margin of twice the proven error bound 2^-10 + 2^-13), the exact op order re-scores them.  (1) 1M x 100, 48
    needles incl. a duplicated row and a parallel vector: indices and scores bit-exact vs the oracle, no rerun; (2) d = 30 (not
    a multiple of 4: scalar staging, k padding); (3) a table built against the sample overflows the per-workgroup entries:
    rerun on the unbatched path, still exact."""
    import os
    from ganrev import synth
    oracle.set_threads(max(1, min(32, os.cpu_count() or 1)))
    r0 = ctx.search_reruns()
    N, d, k = 1_000_000, 100, 50
    emb = synth.normal((N, d), 4242)
    emb[123456] = emb[100]; emb[999_999] = emb[300] * np.float32(2)
    q = np.concatenate([np.array([100, 200, 300, 999_900], dtype=np.int64), (np.arange(44, dtype=np.int64) * 20011 + 7) % N])
    idx, sc = ctx.cosine_topk(emb, q, k)
    ridx, rsc = oracle.cosine_topk(emb, q, k)
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc), "batched search differs from the oracle at 1M x 100"
    del emb
    N, d = 150_000, 30
    emb = synth.normal((N, d), 31)
    q = (np.arange(40, dtype=np.int64) * 3001 + 5) % N
    idx, sc = ctx.cosine_topk(emb, q, k)
    ridx, rsc = oracle.cosine_topk(emb, q, k)
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc), "batched search differs from the oracle at d = 30"
    assert ctx.search_reruns() == r0, "well-spread tables must not overflow the candidate entries"
    stride = N // 16384
    hostile = emb[q[0]][None, :] + np.float32(1e-3) * synth.normal((N, d), 5)
    hostile[::stride] = emb[::stride]
    hostile[q] = emb[q]
    idx, sc = ctx.cosine_topk(hostile, q, k)
    ridx, rsc = oracle.cosine_topk(hostile, q, k)
    assert ctx.search_reruns() == r0 + 1, "the overflow must be detected and the search rerun unbatched"
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)


def test_search_batched_fp16_candidate_pass_bound_and_range_guard(ctx, oracle):
    """Round 6: the batched path's candidate pass runs on the fp16 MFMA (rows and needles rounded to fp16 while staged, error bound 2^-10 + 2^-13 instead of
    bf16's 2^-7 + 2^-10 - search.hip).  (1) a cluster of 300 rows within 2e-4 of each other in cosine around the needle, scattered over a 200 000 x 64
    table: fp16 rounding (up to 1e-3) reorders them at will, the two cuts must keep every one of them and the exact re-score must order them as the oracle
    does - no rerun; (2) the same table scaled by 2^9 (norms inside the fp16 range: still no rerun); (3) tables fp16 cannot carry within the proven bound -
    rows scaled up past 65504 (norm^2 > 65504^2), scaled down into the subnormal range (norm^2 < 2^-10), ONE zero row in an ordinary table, and one needle
    of huge norm among ordinary rows - must be noticed and rerun on the unbatched path, exact."""
    import os
    from ganrev import synth
    oracle.set_threads(max(1, min(32, os.cpu_count() or 1)))
    N, d, k = 200_000, 64, 50
    emb = synth.normal((N, d), 606)
    needle = 777
    cluster = (np.arange(300, dtype=np.int64) * 661 + 13) % N
    emb[cluster] = emb[needle][None, :] + np.float32(0.02) * synth.normal((300, d), 607)
    q = np.concatenate([np.array([needle], dtype=np.int64), (np.arange(39, dtype=np.int64) * 4999 + 11) % N])
    ridx, rsc = oracle.cosine_topk(emb, q, k)
    assert np.isin(ridx[0][1:], cluster).all() and float(rsc[0][1] - rsc[0][k - 1]) < 3e-4, "the case: the needle's top k is the tight cluster"
    r0 = ctx.search_reruns()
    for scale in (1.0, 512.0):
        idx, sc = ctx.cosine_topk(emb * np.float32(scale), q, k)
        oidx, osc = (ridx, rsc) if scale == 1.0 else oracle.cosine_topk(emb * np.float32(scale), q, k)
        assert np.array_equal(idx, oidx) and np.array_equal(sc, osc), f"fp16 candidate pass, scale {scale}"
    assert ctx.search_reruns() == r0, "tables inside the fp16 range must not rerun"
    hostile = []
    hostile.append(("rows past 65504", emb * np.float32(1e5)))
    hostile.append(("rows in the subnormal range", emb * np.float32(2.0 ** -20)))
    z = emb.copy(); z[5] = 0.0
    hostile.append(("one zero row", z))
    h = emb.copy(); h[q[3]] *= np.float32(1e6)
    hostile.append(("one needle of huge norm", h))
    for i, (what, tab) in enumerate(hostile):
        idx, sc = ctx.cosine_topk(tab, q, k)
        oidx, osc = oracle.cosine_topk(tab, q, k)
        assert ctx.search_reruns() == r0 + i + 1, f"{what}: the fp16 range guard must send the call to the unbatched path"
        assert np.array_equal(idx, oidx) and np.array_equal(sc, osc), what


def test_search_filter_bound_and_its_overflow_rerun(ctx, oracle):
    """Tables of >= 2^17 rows are searched through a bound from a strided 16384-row sample (search.hip).  (1) random order: the
    filtered result equals the oracle's bit for bit and no rerun happens; (2) a table built against the sample - every
    sampled row (multiples of n / 16384) random, every other row within 1e-3 of a needle - overflows the candidate lists:
    the library must notice, run again on every key, and still return the oracle's answer."""
    from ganrev import synth
    N, d, k = 150_000, 16, 50
    stride = N // 16384
    q = np.array([7, 77_777], dtype=np.int64)
    emb = synth.normal((N, d), 99)
    r0 = ctx.search_reruns()
    idx, sc = ctx.cosine_topk(emb, q, k)
    ridx, rsc = oracle.cosine_topk(emb, q, k)
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)
    assert ctx.search_reruns() == r0, "random order must not overflow the candidate lists"
    hostile = emb[7][None, :] + np.float32(1e-3) * synth.normal((N, d), 5)
    hostile[::stride] = emb[::stride]
    hostile[7] = emb[7]; hostile[77_777] = emb[77_777]
    idx, sc = ctx.cosine_topk(hostile, q, k)
    ridx, rsc = oracle.cosine_topk(hostile, q, k)
    assert ctx.search_reruns() == r0 + 1, "the overflow must be detected and the search rerun unfiltered"
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)


def test_rccl_single_rank_allreduce(ctx):
    """The RCCL path with nranks=1 (the only world size a 1-GPU box can run): init, all-reduce, destroy."""
    from ganrev import synth
    uid = ctx.comm_unique_id()
    ctx.comm_init(uid, 1, 0)
    try:
        a = synth.normal((1000,), 3)
        d = ctx.upload(a)
        ctx.allreduce(d, a.size)
        ctx.synchronize()
        assert np.array_equal(ctx.download(d, a.shape), a)
    finally:
        ctx.comm_destroy()


def test_sharded_search_over_the_rccl_communicator(ctx):
    """the sharded search's two exchanges - the needle vectors (sum all-reduce) and the candidates (ONE
    ncclAllGather of Q * k * 12 bytes per rank) - through ganrev.parallel.RcclCommunicator, i.e. through libganrev.so's RCCL
    communicator instead of host arrays.  A 1-GPU box can only form a one-rank communicator: the
    collectives still run through RCCL (gr_allreduce_dev / gr_allgather_dev do not bypass it), and the result must equal the
    plain search bit for bit."""
    from ganrev import synth
    from ganrev.parallel import RcclCommunicator, sharded_cosine_topk
    N, d, k = 20011, 100, 50
    emb = synth.normal((N, d), 91); emb[12345] = emb[150]
    needles = [100, 150, N - 1]
    comm = RcclCommunicator(ctx, world=1, rank=0)
    try:
        assert ctx.comm_ranks() == (1, 0)
        got = comm.allgather(np.arange(12, dtype=np.int64).reshape(3, 4))
        assert len(got) == 1 and np.array_equal(got[0], np.arange(12).reshape(3, 4))
        idx, sc = sharded_cosine_topk(ctx.cosine_topk, emb, 0, needles, k, comm)
    finally:
        comm.close()
    ridx, rsc = ctx.cosine_topk(emb, needles, k)
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)


# ------------------------------------------------------------------ committed golden vectors (tests/golden/golden_v1.npz)
import os as _os

from golden_cases import CASES as _CASES, build_case as _build_case, step_inputs as _step_inputs, STRIDE as _STRIDE

_GOLD = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "golden_v1.npz"))


@pytest.mark.parametrize("name", [n for n, c in sorted(_CASES.items()) if c["kind"] in ("R", "G")])
def test_nets_vs_golden(conv_mode, name):
    from ganrev import synth
    case = _CASES[name]
    model, in_dims, x, masks = _build_case(case)
    index = model._descs(tuple(in_dims))[1]
    by_layer = {index[id(m)]: m for m in model.leaves()}
    if case["training"]:
        model.training()
    else:
        model.evaluate()
    flat, grads = model.getParameters()
    for li, keep in masks.items():
        model.setNoise(by_layer[li], keep)
    out = model.forward(x)
    assert_close(out, _GOLD[f"{name}/out"], TOL, f"{name} forward")
    if case["kind"] == "R" and case["training"]:
        gy = synth.normal(out.shape, case["seed"] + 9) * np.float32(0.1)
        grads[...] = 0
        gin = model.backward(x, gy)
        assert_close(gin, _GOLD[f"{name}/gin"], TOL * max(1.0, float(np.abs(_GOLD[f'{name}/gin']).max())), f"{name} gradInput")
        gs = _GOLD[f"{name}/grads_sample"]
        assert_close(grads[::_STRIDE], gs, 2 * TOL * max(1.0, float(np.abs(gs).max())), f"{name} gradient sample")
        rel = abs(float(np.abs(grads.astype(np.float64)).sum()) - float(_GOLD[f"{name}/grads_abs"])) / float(_GOLD[f"{name}/grads_abs"])
        assert rel < 1e-4, f"{name} |grad| checksum off by {rel:.2e}"
        model.pull_params()
        bn0 = [m for m in model.leaves() if hasattr(m, "running_mean")][0]
        assert_close(bn0.running_mean, _GOLD[f"{name}/running_mean0"], 1e-5); assert_close(bn0.running_var, _GOLD[f"{name}/running_var0"], 1e-5)


def test_search_vs_golden(ctx):
    from ganrev import synth
    for name in ("search_10k_32", "search_pixel"):
        case = _CASES[name]
        emb = synth.normal((case["N"], case["d"]), case["seed"])
        emb[case["needles"][0] + 7] = emb[case["needles"][0]]
        idx, sc = ctx.cosine_topk(emb, case["needles"], case["k"])
        assert np.array_equal(idx, _GOLD[f"{name}/idx"]), f"{name}: top-k indices must be bit-exact"
        assert np.array_equal(sc, _GOLD[f"{name}/scores"]), f"{name}: scores must be bit-exact"
        assert np.array_equal(idx[:, :50], _GOLD[f"{name}/idx"][:, :50])      # BASELINE.json: top-50 exact match


def test_train_step_vs_golden(ctx, oracle, conv_mode):
    """First iteration of the golden 3-step trajectory (identical initial state): images, loss, clamped gradient."""
    import ganrev._lib as L
    from ganrev import models, synth
    case = _CASES["step_gray32"]
    dims, nd, B = case["dims"], case["nd"], case["B"]
    G = models.create_G(dims, nd); synth.init_params(G, case["seed"])
    R = models.create_R(dims, nd); synth.init_params(R, case["seed"] + 1)
    oR = oracle.from_model(R, dims)      # only used for mask sizes / layer indices
    G.evaluate(); G.forward(synth.normal((B, nd), 1))
    R.training(); inject_noise(R, oR, B, 0); R.forward(synth.uniform((B,) + dims, 2, 0, 1)); R.push_params()
    R._net.adam_reset()
    inject_noise(R, oR, B, case["seed"] + 1)
    for module, keep in R._pending_masks.values():
        R._net.set_mask(R._leaf_layer(module), keep)
    R._pending_masks = {}
    dn = ctx.upload(_step_inputs(case, 1))
    loss = L.train_r_step(G._net, R._net, dn, B, B, L.Hyper(), 1)
    img = ctx.download(G._net.lib.gr_net_output_dev(G._net.h), (B,) + dims)
    assert_close(img, _GOLD["step_gray32/images1"], TOL, "G images")
    assert abs(loss - float(_GOLD["step_gray32/losses"][0])) < 1e-5 * max(1.0, abs(loss))
    g = R._net.get_grads()
    assert_close(g[::_STRIDE], _GOLD["step_gray32/grads_sample1"], 2e-5, "clamped gradient sample")


def test_fused_step_equals_decomposed_and_comm_path(ctx, oracle, conv_mode):
    """gr_train_r_step (a) == the same iteration through the individual ABI calls, and (b) is unchanged when an RCCL
    communicator (nranks = 1, all a 1-GPU box can run) is active, i.e. with the bucketed all-reduce overlapped with backward
    on the comm stream."""
    import ganrev._lib as L
    from ganrev import models, synth
    from ganrev.parallel import DeviceTrainer
    dims, nd, B = (1, 32, 32), 32, 8
    results = []
    # bit for bit: gr_train_r_step on its stage-by-stage path (fused_head 0).  With the head kernel (the default) the same values come out of sums in another
    # order: that path is held to this one in test_head_kernel_equals_the_stage_by_stage_step and to the oracle in test_train_r_steps_vs_oracle.
    ctx.set_tuning("fused_head", 0)
    try:
        _fused_decomposed_comm(ctx, dims, nd, B, results)
    finally:
        ctx.set_tuning("fused_head", 1)
    for other in results[1:]:
        assert results[0][0] == other[0], "loss trajectories differ"
        assert np.array_equal(results[0][1], other[1]) and np.array_equal(results[0][2], other[2])
    # and with the head kernel: fused == fused + comm, bit for bit (the communicator must not change what the kernel computes)
    results = []
    _fused_decomposed_comm(ctx, dims, nd, B, results, modes=("fused", "fused+comm"))
    assert results[0][0] == results[1][0] and np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])


def _fused_decomposed_comm(ctx, dims, nd, B, results, modes=("fused", "decomposed", "fused+comm")):
    import ganrev._lib as L
    from ganrev import models, synth
    from ganrev.parallel import DeviceTrainer
    for mode in modes:
        G = models.create_G(dims, nd); synth.init_params(G, 5)
        R = models.create_R(dims, nd); synth.init_params(R, 6)
        G.evaluate(); G.forward(synth.normal((2, nd), 1))
        R.training(); R.forward(synth.uniform((2,) + dims, 2, 0, 1)); R.push_params()
        R._net.set_seed(123); R._net.adam_reset()
        if mode == "fused+comm":
            ctx.comm_init(ctx.comm_unique_id(), 1, 0)
        try:
            tr = DeviceTrainer(ctx, G._net, R._net, L.Hyper(), B)
            losses = []
            for t in range(3):
                tr.new_noise(900 + t)
                losses.append(tr.step_decomposed() if mode == "decomposed" else tr.step(want_loss=True))
            results.append((losses, R._net.get_params(), R._net.get_grads()))
        finally:
            if mode == "fused+comm":
                ctx.comm_destroy()


def test_apply_r_pipeline_vs_oracle(oracle, conv_mode):
    """apply_r.lua:145-153 (embed), :265-318 (search on attributes and on pixels), :324-352 (fix faces), :355-390 (anomalies):
    the Python mirror over libganrev against the same composition on the oracle."""
    from ganrev import apply_r, models, nn_utils, synth
    dims, nd, N = (1, 16, 16), 8, 600
    G = models.create_G(dims, nd); synth.init_params(G, 3)
    R = models.create_R(dims, nd); synth.init_params(R, 4)
    Rf = models.create_R(dims, nd, "normal", True); synth.init_params(Rf, 5)
    oG, oR, oRf = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims), oracle.from_model(Rf, dims)
    for o in (oG, oR, oRf):
        o.set_training(False)
    noise = nn_utils.createNoiseInputs(N, nd, "normal", seed=7)
    # the fixer's always-on dropout noise (one mask per forwardBatched chunk of 32 rows) is injected on both sides
    drop = Rf.modules[1]
    li = oRf.layer_index[id(drop)]
    G.evaluate(); images = nn_utils.forwardBatched(G, noise, 32)
    ref_images = np.concatenate([oG.forward(noise[s:s + 32]) for s in range(0, N, 32)])
    assert_close(images, ref_images, TOL, "G images (forwardBatched)")
    R.evaluate(); attributes = nn_utils.forwardBatched(R, images, 32)
    ref_attr = np.concatenate([oR.forward(ref_images[s:s + 32]) for s in range(0, N, 32)])
    assert_close(attributes, ref_attr, TOL, "attributes")
    Rf.evaluate()
    att_fix, ref_fix = [], []
    for s in range(0, N, 32):
        chunk = images[s:s + 32]
        keep = synth.bernoulli_keep((chunk.size,), 1000 + s, 0.5)
        Rf.setNoise(drop, keep); oRf.set_mask(li, keep)
        att_fix.append(Rf.forward(chunk).copy()); ref_fix.append(oRf.forward(ref_images[s:s + 32]))
    att_fix, ref_fix = np.concatenate(att_fix), np.concatenate(ref_fix)
    assert_close(att_fix, ref_fix, TOL, "attributesFixer (always-on v1 dropout)")
    # search: identical inputs on both sides -> indices bit-exact
    by_attr, by_pix = apply_r.createSimilaritySearch(5, 100, images, attributes)
    needles = np.array([99, 199, 299, 399, 499])
    ra, _ = oracle.cosine_topk(attributes, needles, 100)
    rp, _ = oracle.cosine_topk(images.reshape(N, -1), needles, 100)
    assert np.array_equal(by_attr, ra) and np.array_equal(by_pix, rp)
    # fix faces + anomalies
    fixed = apply_r.fixFaces(64, G, att_fix)
    ref_fixed = np.concatenate([oG.forward(ref_fix[s:s + 32]) for s in range(0, 64, 32)])
    assert_close(fixed, ref_fixed, TOL, "fixed faces G(R_fixer(G(z)))")
    dist, below, is_anom = apply_r.detectAnomalies(256, 0.15, images, G, att_fix)
    ref_fixed_all = np.concatenate([oG.forward(ref_fix[s:s + 32]) for s in range(0, 256, 32)])
    ref_dist = 1.0 - oracle.l2_distance_rows(ref_images[:256], ref_fixed_all)
    assert_close(dist, ref_dist, TOL, "1 - torch.dist")
    assert int(is_anom.sum()) == int(np.floor(256 * 0.15))
    assert maxdiff(np.sort(dist)[:10], np.sort(ref_dist)[:10]) <= TOL      # the ten most anomalous distances (their order may swap within TOL)


def test_l2_distance_rows(ctx, oracle):
    from ganrev import synth
    a, b = synth.normal((37, 3, 16, 16), 1), synth.normal((37, 3, 16, 16), 2)
    d = ctx.l2_distance_rows(a, b)
    r = oracle.l2_distance_rows(a, b)
    assert np.max(np.abs(d - r)) <= 1e-12 * np.max(r)
    assert np.allclose(r, np.sqrt(((a.astype(np.float64) - b) ** 2).reshape(37, -1).sum(1)), rtol=1e-6)


def test_train_r_script_learns_to_recover_noise(conv_mode):
    """ganrev.train_r (the train_r.lua mirror): R's MSE against the noise falls over 60 iterations on a fixed random G, in the
    fast (device-resident) loop; and the --compat loop (host tensors, fevalR closure, optim.adam as written in the reference)
    runs the same iteration."""
    from ganrev import train_r
    _, _, losses = train_r.main(["--nbBatches", "60", "--batchSize", "32", "--height", "16", "--width", "16", "--noiseDim", "8",
                                 "--quiet", "--conv-mode", conv_mode, "--save", ""])
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < 0.8 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    _, _, closs = train_r.main(["--nbBatches", "4", "--batchSize", "16", "--height", "16", "--width", "16", "--noiseDim", "8",
                                "--quiet", "--compat", "--conv-mode", conv_mode, "--save", ""])
    assert np.isfinite(closs).all() and len(closs) == 4


@pytest.mark.parametrize("N,d,k,niter", [(10000, 32, 20, 15), (1237, 100, 7, 4), (200000, 100, 20, 3)])
def test_kmeans_and_cluster_assignment(ctx, oracle, N, d, k, niter):
    """apply_r.lua:197-217: unsup.kmeans(attributes, 20, 15) and the (minimum-similarity) nearest-centroid pass.  Labels and
    counts exactly, centroids to fp32 rounding of the fp64 member sums (the device adds them in blocks of 512 rows), cosine
    scores bit for bit (same op order as the search)."""
    from ganrev import synth
    x = synth.normal((N, d), 71)
    x[: N // 3] += 1.5                                # some structure so that clusters differ in size
    c0 = synth.normal((k, d), 72)
    c0 /= np.linalg.norm(c0, axis=1, keepdims=True)
    cent, tot, lab = ctx.kmeans(x, k, niter, c0)
    rcent, rtot, rlab = oracle.kmeans(x, k, niter, c0)
    assert np.array_equal(lab, rlab) and np.array_equal(tot, rtot)
    assert maxdiff(cent, rcent) <= 1e-6
    for take_min in (True, False):
        la, si = ctx.cosine_assign(x, rcent, take_min)
        rla, rsi = oracle.cosine_assign(x, rcent, take_min)
        assert np.array_equal(la, rla) and np.array_equal(si, rsi)


def test_create_cluster_images_mirror(ctx, oracle):
    """ganrev.apply_r.createClusterImages against the same composition on the oracle (apply_r.lua:197-243)."""
    from ganrev import apply_r, synth
    N, d, k = 2000, 32, 20
    attrs = synth.normal((N, d), 81)
    images = synth.uniform((N, 1, 8, 8), 82, 0, 1)
    c0 = apply_r.initialCentroids(k, d, seed=5)
    cent, counts, clusters, faces = apply_r.createClusterImages(k, 15, 50, images, attrs, centroids0=c0)
    rcent, rtot, _ = oracle.kmeans(attrs, k, 15, c0)
    assert maxdiff(cent, rcent) <= 1e-6 and np.array_equal(counts, rtot)
    rla, rsi = oracle.cosine_assign(attrs, rcent, True)
    assert sum(len(c) for c in clusters) <= N and all(len(c) <= 50 for c in clusters)
    for j, members in enumerate(clusters):
        rows = np.nonzero(rla == j)[0]
        ref = rows[np.argsort(-rsi[rows], kind="stable")][:50]
        assert [r for r, _ in members] == list(ref)
        if len(ref):
            assert maxdiff(faces[j], images[ref].mean(0)) <= 1e-6


def test_sharded_search_merge_equals_unsharded(ctx):
    """ganrev.parallel.sharded_cosine_topk with the HIP search as the local search: three unequal shards merged in one process
    (a list-backed communicator) give the unsharded device result bit for bit."""
    from ganrev import synth
    from ganrev.parallel import sharded_cosine_topk
    N, d, k = 30011, 100, 50
    emb = synth.normal((N, d), 88); emb[20000] = emb[150]
    needles = [100, 150, 29999]
    bounds = [(0, 9000), (9000, 21000), (21000, N)]

    shards = [emb[lo:hi] for lo, hi in bounds]
    nd = emb[needles]

    class Comm:
        def __init__(self, r, box): self.rank, self.world, self.box = r, 3, box
        def allreduce_sum(self, arr): return nd.copy()      # what the sum of the three contributions is
        def allgather(self, arr):
            self.box.setdefault(id(self), []).append(arr); return [arr]
    # two phases: collect every rank's candidates, then merge
    from ganrev.parallel import merge_candidates
    cand_i, cand_s = [], []
    for r, (lo, hi) in enumerate(bounds):
        box = {}
        c = Comm(r, box)
        sharded_cosine_topk(ctx.cosine_topk, shards[r], lo, needles, k, c)
        ci, cs = box[id(c)]
        cand_i.append(ci); cand_s.append(cs)
    idx, sc = merge_candidates(cand_i, cand_s, k)
    ridx, rsc = ctx.cosine_topk(emb, needles, k)
    assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)


def test_sharded_search_eight_shards_with_ties_across_three_boundaries(ctx, oracle):
    """cfg4's rank count for the search leg: EIGHT unequal shards of a 160 003 x 100 table searched by the HIP search (every shard large enough
    for the filtered small-needle path and small enough for the plain one: both run), candidates merged as the ranks would; one exact-tie group spread over
    four shards, a duplicate of the needle in the last shard, the cut inside the tie group at k = 4.  Bit-identical to the unsharded device search and to
    the oracle."""
    from helpers import sharded_search_in_process, tied_corpus
    N, d = 160003, 100
    cuts = [0, 9000, 21000, 40000, 40001, 75000, 100000, 131100, N]
    bounds = list(zip(cuts[:-1], cuts[1:]))
    needles = [100, 40000, N - 1]
    emb, group = tied_corpus(N, d, 88, needles[0], bounds)
    for k in (4, 50):
        idx, sc = sharded_search_in_process(ctx.cosine_topk, emb, bounds, needles, k)
        ridx, rsc = ctx.cosine_topk(emb, needles, k)
        assert np.array_equal(idx, ridx) and np.array_equal(sc, rsc)
        oidx, osc = oracle.cosine_topk(emb, needles, k)
        assert np.array_equal(idx, oidx) and np.array_equal(sc, osc)
    assert ridx[0][2:6].tolist() == group


def test_train_r_reads_and_writes_torch7_checkpoints(ctx, tmp_path):
    """train_r.lua:68-75 (G and the geometry come from a Torch7 checkpoint {G=..., opt=...}) and :227-235 (torch.save {R=..., opt=...})
    through ganrev/t7.py: G's images from the loaded model equal the original's, and the saved R reloads to the trained parameters."""
    from ganrev import models, synth, t7, train_r
    G = models.create_G((1, 16, 16), 8); synth.init_params(G, 3)
    gpath, rpath = str(tmp_path / "g.net"), str(tmp_path / "r.net")
    t7.save_checkpoint(gpath, G=G, opt={"noiseDim": 8, "noiseMethod": "normal", "height": 16, "width": 16, "colorSpace": "y"})
    G2, R, losses = train_r.main(["--G", gpath, "--save", rpath, "--nbBatches", "3", "--batchSize", "8", "--quiet", "--height", "99"])
    z = synth.normal((4, 8), 5)
    G.evaluate(); G2.evaluate()
    assert np.array_equal(G.forward(z), G2.forward(z))                    # same weights, same kernels; --height was overridden by opt
    back = t7.load_checkpoint(rpath)
    assert back["opt"]["noiseDim"] == 8 and back["opt"]["height"] == 16
    assert np.array_equal(back["R"]._flat_host(), R._flat_host())
    assert len(losses) == 3 and all(np.isfinite(losses))


def test_train_r_saves_current_running_stats_and_continues(ctx, tmp_path):
    """train_r.lua:227-235 / :101-104.  (1) The default --save is a DIRECTORY: the run writes <save>/r_CxHxW_ndN_<method>.net
    there (every --saveFreq batches and at the end).  (2) The --compat loop (host parameters, optim.adam as written in the
    reference) must save the BatchNorm running statistics the DEVICE holds after its last training forward, not the ones the
    host modules were created with.  (3) --continue reloads that file and goes on from its parameters."""
    from ganrev import t7, train_r
    args = ["--nbBatches", "3", "--batchSize", "8", "--height", "16", "--width", "16", "--noiseDim", "8", "--quiet"]
    for compat in (True, False):
        d = tmp_path / ("compat" if compat else "fast")
        _, R, _ = train_r.main(args + ["--save", str(d), "--saveFreq", "2"] + (["--compat"] if compat else []))
        path = d / "r_1x16x16_nd8_normal.net"
        assert path.exists(), "the reference's file name inside the --save directory (train_r.lua:231)"
        back = t7.load_checkpoint(str(path))["R"]
        dev_bn = [R._net.get_bn_running(i) for i in range(R._net.n_bn())]
        saved_bn = [m for m in back.leaves() if hasattr(m, "running_mean")]
        assert len(dev_bn) == len(saved_bn) == 7
        for (rm, rv), m in zip(dev_bn, saved_bn):
            assert np.array_equal(m.running_mean, rm) and np.array_equal(m.running_var, rv), "stale running statistics were saved"
            assert np.abs(rm).max() > 0 and np.abs(rv - 1).max() > 0           # they did move during training
        assert np.array_equal(back._flat_host(), R._net.get_params())
        _, R2, _ = train_r.main(args[2:] + ["--nbBatches", "0", "--continue", str(path), "--save", ""])
        assert np.array_equal(R2._flat_host(), back._flat_host())


def test_train_r_uniform_noise_targets(ctx):
    """utils/nn_utils.lua:39-51 + models.lua:452-454: with noiseMethod 'uniform' R ends in a Tanh and its targets are drawn
    from uniform(-1, 1) - in the device-resident loop as well as in the --compat loop (N(0,1) targets outside (-1, 1) cannot be
    reached: the loss would stall near the variance of the unreachable part)."""
    from ganrev import train_r
    from ganrev.parallel import DeviceTrainer
    import ganrev._lib as L
    n = 1 << 16
    d = ctx.malloc(4 * n)
    ctx.fill_uniform(d, n, 5)
    u = ctx.download(d, (n,)); ctx.free(d)
    assert u.min() >= -1 and u.max() < 1 and abs(float(u.mean())) < 0.02 and abs(float(u.var()) - 1 / 3) < 0.01
    common = ["--batchSize", "32", "--height", "16", "--width", "16", "--noiseDim", "8", "--quiet", "--noiseMethod", "uniform", "--save", ""]
    _, R, fast = train_r.main(common + ["--nbBatches", "60"])
    assert R.leaves()[-1].typename == "nn.Tanh"
    _, _, compat = train_r.main(common + ["--nbBatches", "6", "--compat"])
    # both loops start near E[(tanh(~0) - u)^2] ~ var(U(-1,1)) = 1/3, far below the ~1 a N(0,1) target would give
    assert 0.2 < fast[0] < 0.6 and 0.2 < compat[0] < 0.6, (fast[0], compat[0])
    assert np.mean(fast[-10:]) < 0.9 * np.mean(fast[:5])


# ---------------------------------------------------------------------------------------------------------------------
# the module types the D network adds (models.lua:272-337) and the network itself
def _d_chain(kind):
    from ganrev import nn
    if kind == "small":       # odd channel counts, planes smaller than a tile
        return (nn.Sequential().add(nn.SpatialConvolution(2, 6, 3, 3, 1, 1, 1, 1)).add(nn.PReLU())
                .add(nn.SpatialConvolution(6, 4, 5, 5, 1, 1, 2, 2)).add(nn.PReLU()).add(nn.SpatialDropout(0.25)).add(nn.SpatialMaxPooling(2, 2))
                .add(nn.View(4 * 4 * 4)).add(nn.Linear(64, 5)).add(nn.PReLU()).add(nn.Linear(5, 1)).add(nn.Sigmoid())), (2, 8, 8)
    if kind == "bn":          # a PReLU behind a BatchNorm opens a stage of its own; 5x5 on a plane wider than one tile
        return (nn.Sequential().add(nn.SpatialConvolution(3, 16, 5, 5, 1, 1, 2, 2)).add(nn.SpatialBatchNormalization(16)).add(nn.PReLU())
                .add(nn.Dropout(0.5)).add(nn.SpatialConvolution(16, 8, 3, 3, 1, 1, 1, 1)).add(nn.PReLU()).add(nn.SpatialMaxPooling(2, 2))
                .add(nn.View(8 * 12 * 10)).add(nn.Linear(8 * 12 * 10, 3))), (3, 24, 20)
    # the D network's own 5x5 layer (models.lua:297) at 32x32 images: createNxN(128, 64, 5, 0.2) on 16x16 planes
    return (nn.Sequential().add(nn.SpatialConvolution(16, 128, 3, 3, 1, 1, 1, 1)).add(nn.PReLU())
            .add(nn.SpatialConvolution(128, 64, 5, 5, 1, 1, 2, 2)).add(nn.PReLU()).add(nn.SpatialDropout(0.25)).add(nn.SpatialMaxPooling(2, 2))
            .add(nn.View(64 * 8 * 8)).add(nn.Linear(64 * 8 * 8, 32)).add(nn.PReLU()).add(nn.Dropout(0.25)).add(nn.Linear(32, 1)).add(nn.Sigmoid())), (16, 16, 16)


@pytest.mark.parametrize("kind,B", [("small", 3), ("bn", 4), ("d2_left", 6)])
def test_5x5_convolution_and_prelu_vs_oracle(oracle, conv_mode, kind, B):
    """nn.SpatialConvolution(.., 5, 5, 1, 1, 2, 2) and nn.PReLU() (one learnable slope, a parameter in the flat vector) inside
    nn.Sequential: forward (training and evaluate), gradInput and every gradient tensor - the 5x5 weights and each PReLU's slope
    included - against the oracle, three arithmetics for the 3x3 / Linear layers around them."""
    from ganrev import synth
    from helpers import adopt_device_argmax, adopt_device_kinks, assert_grads_close
    net, dims = _d_chain(kind)
    synth.init_params(net, 41)
    for k, m in enumerate(m for m in net.leaves() if m.typename == "nn.PReLU"):
        m.weight[0] = np.float32(0.25 + 0.125 * k)
    flat, grads = net.getParameters()
    onet = oracle.from_model(net, dims)
    x = synth.normal((B,) + dims, 43)
    net.training(); onet.set_training(True)
    inject_noise(net, onet, B, 6)
    ref = onet.forward(x)
    out = net.forward(x)
    assert_close(out, ref, TOL * max(1.0, float(np.abs(ref).max())), "forward (training)")
    adopt_device_argmax(net, onet, B, 8)
    ref = onet.forward(x)                        # the oracle again, on the argmax the device took
    assert_close(out, ref, TOL * max(1.0, float(np.abs(ref).max())), "forward vs the argmax-forced oracle")
    adopt_device_kinks(net, onet, B, 8)
    gy = synth.normal(ref.shape, 9) * np.float32(0.5)
    grads[...] = 0; onet.zero_grads()
    gin = net.backward(x, gy)
    ref_gin = onet.backward(x, gy)
    assert_close(gin, ref_gin, TOL * max(1.0, float(np.abs(ref_gin).max())), "gradInput")
    assert_grads_close(net, grads, onet.grads, 1e-4, 1e-3)
    slopes = [(lo, m) for m, nm, lo, hi in __import__("helpers").param_segments(net) if m.typename == "nn.PReLU"]
    assert slopes and max(abs(float(onet.grads[lo])) for lo, _ in slopes) > 1e-3, "no PReLU slope gradient large enough for the bar to see"
    __import__("helpers").release_argmax(net, onet)
    net.pull_params()                            # running statistics: the oracle's saw one more training forward than the device's
    for bi, m in enumerate(m for m in net.leaves() if hasattr(m, "running_mean")):
        rm, rv = onet.bn_running(bi)
        rm[...] = m.running_mean; rv[...] = m.running_var
    net.evaluate(); onet.set_training(False)
    ref_e = onet.forward(x)
    assert_close(net.forward(x), ref_e, TOL * max(1.0, float(np.abs(ref_e).max())), "forward (evaluate)")


@pytest.mark.parametrize("dims,B", [((1, 32, 32), 6), ((3, 32, 32), 4), ((3, 64, 64), 2)])
def test_D2_forward_backward_vs_oracle(oracle, conv_mode, dims, B):
    """MODEL_D = models.create_D2 (models.lua:272-337): nn.Concat(2) of the 5x5 tower and the deeper 3x3 tower, run as four
    compiled parts chained on the host.  Output, gradInput w.r.t. the images (what adversarial.lua:113-118 hands to G) and the
    whole flat gradient in getParameters() order against the oracle composed the same way."""
    from ganrev import models, synth
    from helpers import OracleGraph, adopt_device_argmax, adopt_device_kinks, assert_grads_close
    D = models.create_D2(dims, seed=3); synth.init_params(D, 19)
    flat, grads = D.getParameters()
    assert [type(p).__name__ for p in D.parts()] == ["Sequential", "Concat", "Sequential"]
    og = OracleGraph(oracle, D, dims)
    assert sum(o.params.size for _, o in og.pairs) == flat.size and len(og.pairs) == 4
    x = synth.uniform((B,) + dims, 5, 0, 1)
    D.training(); og.set_training(True)
    for chunk, onet in og.pairs:
        inject_noise(chunk, onet, B, 11)
    ref = og.forward(x)
    out = D.forward(x)
    assert out.shape == (B, 1) and ref.min() > 0 and ref.max() < 1
    assert_close(out, ref, TOL, "D(x) (training)")
    for chunk, onet in og.pairs:
        adopt_device_argmax(chunk, onet, B, 16)
    ref = og.forward(x)                          # the oracle again, on the argmax the device took
    assert_close(out, ref, TOL, "D(x) vs the argmax-forced oracle")
    for chunk, onet in og.pairs:
        adopt_device_kinks(chunk, onet, B, 16)
    gy = synth.normal(ref.shape, 9)
    grads[...] = 0; og.zero_grads()
    gin = D.backward(x, gy)
    ref_gin = og.backward(x, gy)
    assert gin.shape == x.shape
    assert_close(gin, ref_gin, TOL * float(np.abs(ref_gin).max()), "gradInput w.r.t. the images")
    assert_grads_close(D, grads, og.grads, 1e-4, 1e-3)
    for chunk, onet in og.pairs:
        __import__("helpers").release_argmax(chunk, onet)
    D.evaluate(); og.set_training(False)
    assert_close(D.forward(x), og.forward(x), TOL, "D(x) (evaluate)")


def test_adversarial_step_vs_oracle(oracle, conv_mode):
    """adversarial.lua:66-133, the two closures of the GAN game, each from the oracle's state: fevalD (D forward, BCE, D backward,
    L2 + clamp) and fevalG_on_D (G forward in training mode, D forward, BCE against "real", D backward to the images, G backward
    from D's gradInput, clamp) - loss and the whole flat gradient of D resp. G; then optim.adam on the four-part D (stepped
    slice by slice) bit-exact against the oracle's Adam, and one epoch of adversarial.train end to end."""
    from ganrev import adversarial, models, synth
    from helpers import OracleGraph, adopt_device_argmax, adopt_device_kinks, assert_grads_close
    dims, nd, B = (1, 32, 32), 16, 8
    G = models.create_G(dims, nd, seed=1); synth.init_params(G, 2)
    D = models.create_D2(dims, seed=2); synth.init_params(D, 3)
    env = adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd, N_epoch=2)
    oG = oracle.from_model(G, (nd, 1, 1)); oG.set_training(True)
    og = OracleGraph(oracle, D, dims); og.set_training(True)
    theta_d = np.concatenate([o.params for _, o in og.pairs])
    assert np.array_equal(theta_d, env.PARAMETERS_D)

    def inject(seed):
        for chunk, onet in og.pairs:
            inject_noise(chunk, onet, B, seed)

    def adopt(x):
        """the device's pool argmax onto the oracle, the oracle's forward again on it, then the device's side of every kink"""
        for chunk, onet in og.pairs:
            adopt_device_argmax(chunk, onet, B, 16)
        ref = og.forward(x)
        for chunk, onet in og.pairs:
            adopt_device_kinks(chunk, onet, B, 16)
        return ref

    # ---- fevalD on half real, half generated images
    inputs = np.concatenate([synth.uniform((B // 2,) + dims, 7, 0, 1), oG.forward(synth.normal((B // 2, nd), 8))]).astype(np.float32)
    targets = np.concatenate([np.ones(B // 2, np.float32), np.zeros(B // 2, np.float32)])
    inject(21)
    ref_out = og.forward(inputs)
    assert_close(D.forward(inputs), ref_out, TOL, "D(inputs)")
    ref_out = adopt(inputs)
    inject(21)
    f, g = adversarial.make_fevalD(env, inputs, targets)(env.PARAMETERS_D)
    rf, rdf = oracle.bce(ref_out.reshape(-1), targets)
    og.zero_grads(); og.backward(inputs, rdf.reshape(ref_out.shape))
    rg = og.grads + theta_d * np.float32(env.OPT.D_L2)
    np.clip(rg, -env.OPT.D_clamp, env.OPT.D_clamp, out=rg)
    rf += env.OPT.D_L2 * float(np.dot(theta_d.astype(np.float64), theta_d.astype(np.float64))) / 2
    assert abs(f - rf) <= 1e-5 * max(1.0, abs(rf)), f"fevalD loss {f} vs {rf}"
    assert_grads_close(D, g, rg, 1e-4, 1e-3, "fevalD")
    assert env.CONFUSION.sum() == B

    # ---- optim.adam on a model that runs as several gr_nets: slice-by-slice update == one Adam over the flat vector
    theta0, g0 = env.PARAMETERS_D.copy(), g.copy()
    state = {}
    from ganrev import optim
    optim.adam(lambda x: (f, g0), env.PARAMETERS_D, state, model=D)
    m, v = np.zeros_like(theta0), np.zeros_like(theta0)
    rtheta, rgc = theta0.copy(), g0.copy()
    oracle.penalty_clamp_adam(rtheta, rgc, m, v, oracle.GoHyper(l1=0.0, l2=0.0, clamp=0.0), 1)
    assert np.array_equal(env.PARAMETERS_D, rtheta) and np.array_equal(state["m"], m) and np.array_equal(state["v"], v)
    env.PARAMETERS_D[...] = theta0

    # ---- fevalG_on_D
    for chunk, onet in og.pairs:
        __import__("helpers").release_argmax(chunk, onet)
    noise = synth.normal((B, nd), 31)
    ones = np.ones(B, np.float32)
    inject(22)
    rimg = oG.forward(noise)
    img = G.forward(noise).copy()
    assert_close(img, rimg, TOL, "G(noise) in training mode")
    ref_out = og.forward(rimg)
    assert_close(D.forward(img), ref_out, TOL, "D(G(noise))")
    ref_out = adopt(rimg); adopt_device_kinks(G, oG, B, 16)
    inject(22)
    f, g = adversarial.make_fevalG_on_D(env, noise, ones)(env.PARAMETERS_G)
    rf, rdf = oracle.bce(ref_out.reshape(-1), ones)
    og.zero_grads(); rgin = og.backward(rimg, rdf.reshape(ref_out.shape))
    assert_close(D.gradInput, rgin, TOL * float(np.abs(rgin).max()), "D's gradInput w.r.t. the generated images")
    oG.zero_grads(); oG.backward(noise, rgin)
    rg = oG.grads.copy()
    np.clip(rg, -env.OPT.G_clamp, env.OPT.G_clamp, out=rg)
    assert abs(f - rf) <= 1e-5 * max(1.0, abs(rf)), f"fevalG_on_D loss {f} vs {rf}"
    assert_grads_close(G, g, rg, 1e-4, 1e-6, "fevalG_on_D")

    # ---- one epoch end to end (Philox dropout noise, two batches): finite losses, both parameter vectors move, CONFUSION is reset
    pd, pg = env.PARAMETERS_D.copy(), env.PARAMETERS_G.copy()
    tv = adversarial.train(env, synth.uniform((B,) + dims, 40, 0, 1))
    assert 0.0 <= tv <= 1.0 and env.CONFUSION.sum() == 0
    assert len(env.last_losses["D"]) == 2 and len(env.last_losses["G"]) == 2 and np.all(np.isfinite(env.last_losses["D"] + env.last_losses["G"]))
    assert 1e-4 < np.abs(env.PARAMETERS_D - pd).max() <= 2.1e-3 and 1e-4 < np.abs(env.PARAMETERS_G - pg).max() <= 2.1e-3
    with pytest.raises(IndexError):
        adversarial.train(env, synth.uniform((B // 2,) + dims, 41, 0, 1))     # trainData shorter than the epoch needs


@pytest.mark.parametrize("method", ["sgd", "adagrad", "adadelta", "adamax", "rmsprop"])
def test_adversarial_game_with_the_other_optimisers(method):
    """adversarial.lua:156-171 / 183-198 with --D_optmethod / --G_optmethod other than adam: the closures run on the GPU, the update is the
    host mirror of the optim rock on PARAMETERS_D / PARAMETERS_G.  The dispatch itself on fevalD (first step from an empty state: the step
    equals the method's rule applied to the gradient the closure returned), then one whole batch of adversarial.train: both networks move,
    the state lands in OPTSTATE[method], and the next forward runs with the stepped parameters."""
    from ganrev import adversarial, models, synth
    dims, nd, B = (1, 16, 16), 8, 8
    G, D = models.create_G(dims, nd, True, 3), models.create_D2(dims, True, 4)
    env = adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd, N_epoch=1, D_optmethod=method, G_optmethod=method, D_sgd_momentum=0.5)
    first = {"sgd": lambda g: 0.02 * g, "adagrad": lambda g: 1e-3 * g / (np.abs(g) + 1e-10),
             "adadelta": lambda g: np.sqrt(1e-6) / np.sqrt(0.1 * g * g + 1e-6) * g, "adamax": lambda g: 2e-3 * g / (np.abs(g) + 1e-38),
             "rmsprop": lambda g: 1e-2 * g / (np.sqrt(0.01 * g * g) + 1e-8)}[method]
    G.evaluate()
    inputs = np.concatenate([synth.uniform((B // 2,) + dims, 7, 0, 1), G.forward(synth.normal((B // 2, nd), 8))]).astype(np.float32)
    G.training()
    targets = np.concatenate([np.ones(B // 2, np.float32), np.zeros(B // 2, np.float32)])
    before = env.PARAMETERS_D.copy()
    x, fs = adversarial._optimize(env, "D", adversarial.make_fevalD(env, inputs, targets), env.PARAMETERS_D, D)
    g = env.GRAD_PARAMETERS_D.astype(np.float64)                      # what the closure returned (penalty and clamp applied)
    nz = g != 0
    assert x is env.PARAMETERS_D and len(fs) == 1 and np.isfinite(fs[0]) and nz.any() and np.all(np.isfinite(x))
    assert np.allclose((before.astype(np.float64) - x)[nz], first(g[nz]), rtol=1e-3, atol=2e-7), method
    assert np.array_equal(before[~nz], x[~nz])
    assert env.OPTSTATE["adam"]["D"] == {} and len(env.OPTSTATE[method]["D"]) > 0
    pd, pg = env.PARAMETERS_D.copy(), env.PARAMETERS_G.copy()
    adversarial.train(env, synth.uniform((B // 2,) + dims, 40, 0, 1))
    assert np.all(np.isfinite(env.PARAMETERS_D)) and np.all(np.isfinite(env.PARAMETERS_G))
    assert not np.array_equal(env.PARAMETERS_D, pd) and len(env.OPTSTATE[method]["G"]) > 0
    # (rmsprop's first step moves every weight of D by 0.1: D saturates and hands G a gradient of exact zeros - then G rightly stays)
    assert not np.array_equal(env.PARAMETERS_G, pg) or not env.GRAD_PARAMETERS_G.any()
    D.evaluate(); D.forward(synth.uniform((2,) + dims, 41, 0, 1))
    for ch, lo, hi in D._param_chunks():                               # the next forward runs with the stepped parameters
        if hi > lo:
            assert np.array_equal(ch._net.get_params(), env.PARAMETERS_D[lo:hi])


from golden_cases import DCASES as _DCASES, build_dcase as _build_dcase  # noqa: E402

_GOLD_D = np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "golden_v2_dnet.npz"))


@pytest.mark.parametrize("name", sorted(_DCASES))
def test_dnet_vs_golden(conv_mode, name):
    """The D network's module types against the committed vectors (tests/golden/golden_v2_dnet.npz): no oracle in the loop."""
    from ganrev import synth
    case = _DCASES[name]
    model, x = _build_dcase(case)
    model.training()
    flat, grads = model.getParameters()
    # the dropout noise helpers.inject_noise drew for the oracle: seed * 131 + layer index inside the compiled part
    dims = tuple(case["dims"])

    def inject(part, d):
        from ganrev import nn
        if isinstance(part, nn.Concat):
            outs = [inject(b, d) for b in part.modules]
            od = list(outs[0]); od[part.dimension - 2] = sum(o[part.dimension - 2] for o in outs)
            return tuple(od)
        if part._is_graph():
            for p in part.parts():
                d = inject(p, d)
            return d
        index = part._descs(d)[1]
        for m in part.leaves():
            ds, nd_ = m.desc(d)
            if m.typename in ("nn.Dropout", "nn.SpatialDropout"):
                n = case["B"] * (int(np.prod(d)) if m.typename == "nn.Dropout" else d[0])
                part.setNoise(m, synth.bernoulli_keep((n,), case["seed"] * 131 + index[id(m)], m.p))
            d = nd_
        return d
    inject(model, dims)
    out = model.forward(x)
    assert_close(out, _GOLD_D[f"{name}/out"], TOL, f"{name} forward")
    gy = synth.normal(out.shape, case["seed"] + 9)
    grads[...] = 0
    gin = model.backward(x, gy)
    ggin = _GOLD_D[f"{name}/gin"]
    assert_close(gin, ggin, TOL * max(1.0, float(np.abs(ggin).max())), f"{name} gradInput")
    gs = _GOLD_D[f"{name}/grads_sample"]
    assert_close(grads[::_STRIDE], gs, 2 * TOL * max(1.0, float(np.abs(gs).max())), f"{name} gradient sample")
    rel = abs(float(np.abs(grads.astype(np.float64)).sum()) - float(_GOLD_D[f"{name}/grads_abs"])) / float(_GOLD_D[f"{name}/grads_abs"])
    assert rel < 1e-4, f"{name} |grad| checksum off by {rel:.2e}"


def test_device_resident_gan_batch_matches_the_host_mirror(conv_mode):
    """ganrev.adversarial.DeviceGame (images, gradients, parameters and Adam state on the GPU; nn.Concat through gr_copy2d_dev /
    gr_add_dev; penalty + clamp + Adam fused per part) against adversarial.train, the line-by-line mirror whose closures the
    oracle test above pins: one batch from identical state, identical noise and (same seeds, same forward counts) identical
    Philox dropout masks.  Same kernels on both sides, so the losses agree to rounding and the parameters differ only where
    Adam's normalised update amplifies the last-bit difference between g + l2 * theta evaluated on the host and on the device."""
    from ganrev import adversarial, models, nn_utils, synth
    dims, nd, B = (1, 32, 32), 16, 8
    envs = []
    for _ in range(2):
        G = models.create_G(dims, nd, seed=1); synth.init_params(G, 2)
        D = models.create_D2(dims, seed=2); synth.init_params(D, 3)
        envs.append(adversarial.make_env(G, D, dims, batchSize=B, noiseDim=nd, N_epoch=1, seed=5))
    host, dev = envs
    assert np.array_equal(host.PARAMETERS_D, dev.PARAMETERS_D) and np.array_equal(host.PARAMETERS_G, dev.PARAMETERS_G)
    game = adversarial.DeviceGame(dev)                       # compiles with one forward of G and D ...
    host.MODEL_D.forward(host.MODEL_G.forward(nn_utils.createNoiseInputs(2, nd, "normal", seed=1)))     # ... so the mirror does the same
    real = synth.uniform((B // 2,) + dims, 40, 0, 1)
    noise_d = nn_utils.createNoiseInputs(B // 2, nd, "normal", seed=5 * 100003 + 1)     # what adversarial._noise will draw
    noise_g = nn_utils.createNoiseInputs(B, nd, "normal", seed=5 * 100003 + 2)
    pd0, pg0 = host.PARAMETERS_D.copy(), host.PARAMETERS_G.copy()
    import ganrev._lib as L
    c = L.default_context()
    g0 = c.range_guard_stats()
    adversarial.train(host, real)
    g1 = c.range_guard_stats()
    ld, lg = game.batch(real, noise_d, noise_g, want_loss=True)
    game.sync_to_host()
    # both sides must have run the arithmetic under test: a host pass sent to bf16x6 by the f16x3 range guard (or a context left on
    # bf16x6 by an earlier test's tripped trainer guard) would make this a comparison of two different arithmetics
    assert c.conv_mode() == conv_mode and g1[1] == g0[1] == c.range_guard_stats()[1], (c.conv_mode(), g0, g1, c.range_guard_stats())
    # the mirror's f includes the L2 penalty term (adversarial.lua:86-88); the device loss word is the criterion alone
    pen = host.OPT.D_L2 * float(np.dot(pd0.astype(np.float64), pd0.astype(np.float64))) / 2
    assert abs(ld + pen - host.last_losses["D"][0]) <= 1e-5 * max(1.0, abs(ld)), (ld, pen, host.last_losses["D"][0])
    assert abs(lg - host.last_losses["G"][0]) <= 1e-5 * max(1.0, abs(lg)), (lg, host.last_losses["G"][0])
    for name, a, b, p0 in (("D", dev.PARAMETERS_D, host.PARAMETERS_D, pd0), ("G", dev.PARAMETERS_G, host.PARAMETERS_G, pg0)):
        d = np.abs(a.astype(np.float64) - b)
        moved = np.abs(b.astype(np.float64) - p0)
        assert moved.max() > 5e-4, f"{name}: the batch did not move the parameters"
        where = ""
        if not (np.median(d) <= 1e-7 and (d > 1e-5).mean() <= 2e-3 and d.max() <= 2.1e-3):      # say WHICH tensors differ
            from helpers import param_segments
            model = host.MODEL_D if name == "D" else host.MODEL_G
            for chunk, lo, hi in model._param_chunks():
                for mod, nm, l2, h2 in param_segments(chunk):
                    dd = d[lo + l2:lo + h2]
                    if dd.size and (dd > 1e-5).mean() > 2e-3:
                        where += f" | {mod.typename}.{nm}[{lo + l2}:{lo + h2}] {(dd > 1e-5).mean():.1e}"
        assert np.median(d) <= 1e-7 and (d > 1e-5).mean() <= 2e-3 and d.max() <= 2.1e-3, \
            f"{name}: median {np.median(d):.2e}, share above 1e-5 {(d > 1e-5).mean():.2e}, max {d.max():.2e}{where}"


@pytest.mark.parametrize("compat", [False, True])
def test_train_loop_saves_a_loadable_checkpoint(tmp_path, compat):
    """ganrev.train (train.lua:125-257 around adversarial.train): two epochs of two batches on synthetic images in the fast
    (device-resident) and the --compat loop, then the Torch7 checkpoint {D, G, opt, epoch} it wrote reads back into the same
    layers and parameters, the four-part D included, and the loop continues from it (--network)."""
    from ganrev import t7, train
    argv = ["--epochs", "2", "--N_epoch", "2", "--batchSize", "8", "--noiseDim", "16", "--save", str(tmp_path), "--saveFreq", "1", "--quiet"]
    res = train.main(argv + (["--compat"] if compat else []))
    assert res["epoch"] == 2 and np.all(np.isfinite(res["last_losses"]))
    env = res["env"]
    assert _os.path.isfile(res["path"]) and _os.path.isfile(res["path"] + ".old")          # train.lua:247-249 keeps the previous file
    ck = t7.load_checkpoint(res["path"])
    assert "_unconverted" not in ck and ck["epoch"] == 2 and ck["opt"]["batchSize"] == 8
    for key, model in (("D", env.MODEL_D), ("G", env.MODEL_G)):
        assert [m.typename.split(".")[-1] for m in ck[key].leaves()] == [m.typename.split(".")[-1] for m in model.leaves()]
        assert np.array_equal(ck[key]._flat_host(), model.getParameters()[0])
    g_bn = [m for m in ck["G"].leaves() if hasattr(m, "running_mean")][0]
    assert float(np.abs(g_bn.running_mean).max()) > 0                                      # G trained in training mode: its statistics moved and were saved
    res2 = train.main(argv[:1] + ["1"] + argv[2:] + ["--network", res["path"]] + (["--compat"] if compat else []))
    assert res2["epoch"] == 3 and np.all(np.isfinite(res2["last_losses"]))                 # the epoch counter continues behind the file's: EPOCH = tmp.epoch + 1 (train.lua:113)
    assert t7.load_checkpoint(res2["path"])["epoch"] == 3
