"""-m gpu: BASELINE configs[4] as stated - "generated faces -> embeddings -> top-k" - resident on the GPU.

apply_r.lua:145-153 builds the search corpus with forwardBatched(MODEL_G, noise) / forwardBatched(MODEL_R, images)
(utils/nn_utils.lua:5-33).  ganrev.apply_r.embed_dev is that pipeline with nothing visiting the host (gr_embed_dev: per chunk G forward
-> R forward [-> R_fixer forward], every chunk's last kernel writing its rows of the [N x nd] table itself).  Checked here:
  * bit for bit against the host-tensor mirror (apply_r.embed = forwardBatched over gr_net_forward_host) at the same chunk size,
    ragged last chunk included, with and without the images kept;
  * against the oracle at the 1e-4 bar on rows of the first, a middle and the ragged last chunk (evaluate() mode: a row's result
    does not depend on the rest of its batch, so a subset pins every kernel instantiation the full table ran);
  * the search (apply_r.lua:265-318) on the device-resident tables, by attributes and by pixels, bit-exact against the oracle's
    search of the same tables.
"""
import numpy as np
import pytest

from helpers import TOL, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["f32", "f16x3"])
def conv_mode2(request):
    """exact fp32 and the default f16x3 arithmetic (bf16x6 shares f16x3's kernels: the module-level tests cover it)"""
    import ganrev._lib as L
    c = L.default_context()
    c.set_tuning("range_guard", 0); c.set_tuning("range_guard", 1)
    prev = c.conv_mode()
    c.set_conv_mode(request.param)
    yield request.param
    c.set_tuning("range_guard", 0); c.set_tuning("range_guard", 1)
    c.set_conv_mode(prev)


@pytest.mark.parametrize("dims,nd,N,batch", [((1, 32, 32), 32, 4096 + 40, 512), ((3, 64, 64), 100, 4096 + 24, 512)])
def test_embed_dev_equals_forwardBatched_and_the_oracle(ctx, oracle, conv_mode2, dims, nd, N, batch):
    from ganrev import apply_r, models, nn_utils, synth
    G = models.create_G(dims, nd); synth.init_params(G, 3)
    R = models.create_R(dims, nd); synth.init_params(R, 4)
    Rf = models.create_R(dims, nd, "normal", True); synth.init_params(Rf, 5)
    noise = nn_utils.createNoiseInputsDev(ctx, N, nd, "normal", seed=11)
    noise_h = noise.numpy()

    # host-tensor mirror of apply_r.lua:145-153 (every chunk crosses PCIe twice), same chunk size
    Rf.manualSeed(77)                                   # restarts the Philox counter of the fixer's always-on Dropout
    images_h, attr_h, fix_h = apply_r.embed(G, R, noise_h, batch, Rf)
    # the same, resident on the GPU
    Rf.manualSeed(77)
    images_d, attr_d, fix_d = apply_r.embed_dev(G, R, noise, batch, Rf, keep_images=True)
    assert images_d.shape == images_h.shape and attr_d.shape == attr_h.shape == (N, nd)
    a_d, f_d, i_d = attr_d.numpy(), fix_d.numpy(), images_d.numpy()
    assert np.array_equal(i_d, images_h), "G images: device-resident pipeline != forwardBatched"
    assert np.array_equal(a_d, attr_h), "attributes: device-resident pipeline != forwardBatched"
    assert np.array_equal(f_d, fix_h), "attributesFixer: device-resident pipeline != forwardBatched"
    # without keeping the images (a chunk's images live in G's output buffer until R has read them): same tables
    Rf.manualSeed(77)
    none, attr_d2, fix_d2 = apply_r.embed_dev(G, R, noise, batch, Rf)
    assert none is None
    assert np.array_equal(attr_d2.numpy(), a_d) and np.array_equal(fix_d2.numpy(), f_d)
    # forwardBatchedDev alone (utils/nn_utils.lua:5-33 on device rows) reproduces the R table from the kept images
    attr_d3 = nn_utils.forwardBatchedDev(R, images_d, batch)
    assert np.array_equal(attr_d3.numpy(), a_d)
    assert R._net.lib.gr_net_output_dev(R._net.h) == attr_d3.ptr + 4 * nd * (((N - 1) // batch) * batch), "m.output = the last chunk's rows"

    # oracle: rows of the first chunk, of a middle chunk and the whole ragged tail
    oG, oR, oRf = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims), oracle.from_model(Rf, dims)
    for o in (oG, oR, oRf):
        o.set_training(False)
    tail = N % batch
    sel = np.concatenate([np.arange(0, 24), np.arange(5 * batch + 100, 5 * batch + 116), np.arange(N - tail, N)])
    ref_img = oG.forward(noise_h[sel])
    assert_close(i_d[sel], ref_img, TOL, "G images vs oracle")
    assert_close(a_d[sel], oR.forward(ref_img), TOL, "attributes vs oracle")
    # the fixer: the noise of the LAST forward (the ragged tail chunk) is still readable; inject it into the oracle
    drop = Rf.modules[1]
    keep = Rf.getNoise(drop, tail)
    oRf.set_mask(oRf.layer_index[id(drop)], keep)
    assert_close(f_d[N - tail:], oRf.forward(ref_img[-tail:]), TOL, "attributesFixer (tail chunk) vs oracle")
    assert 0.3 < keep.mean() < 0.7

    # the search where the tables were written: apply_r.lua:265-318, five needles (rows 100..500), top-100
    by_attr, by_pix = apply_r.createSimilaritySearchDev(5, 100, attr_d, images_d)
    needles = np.array([99, 199, 299, 399, 499])
    ra, _ = oracle.cosine_topk(a_d, needles, 100)
    rp, _ = oracle.cosine_topk(i_d.reshape(N, -1), needles, 100)
    assert np.array_equal(by_attr, ra), "top-100 by recovered noise"
    assert np.array_equal(by_pix, rp), "top-100 by pixels"
    for t in (noise, images_d, attr_d, fix_d, attr_d2, fix_d2, attr_d3):
        t.free()


def test_forward_dev_destination_in_training_mode_is_a_copy(ctx):
    """In training mode the stage buffers are state the backward reads: a caller's destination receives a copy and m.output stays
    the net's own buffer; in evaluate() mode the last kernel writes the destination itself and m.output points there."""
    from ganrev import models, nn_utils, synth
    dims, nd, B = (1, 16, 16), 8, 12
    R = models.create_R(dims, nd); synth.init_params(R, 4)
    x = nn_utils.DeviceTensor(ctx, (B,) + dims)
    ctx.upload(synth.uniform((B,) + dims, 3, 0, 1), x.ptr)
    out = nn_utils.DeviceTensor(ctx, (B, nd))
    R.training()
    net = R.device_net(dims)
    net.set_seed(5)
    net.forward_dev(x.ptr, B, out.ptr)
    own = net.lib.gr_net_output_dev(net.h)
    assert own != out.ptr
    assert np.array_equal(out.numpy(), ctx.download(own, (B, nd)))
    net.backward_dev(x.ptr, out.ptr, B, None)            # the forward state is intact: backward runs
    R.evaluate()
    net = R.device_net(dims)
    net.forward_dev(x.ptr, B, out.ptr)
    assert net.lib.gr_net_output_dev(net.h) == out.ptr
    ref = R.forward(x.numpy())
    assert np.array_equal(out.numpy(), ref)
    # an unaligned destination (row count x nd x 4 not a multiple of 16) falls back to the copy
    odd = nn_utils.DeviceTensor(ctx, (B + 1, nd))
    net.forward_dev(x.ptr, B, odd.ptr + 4)
    assert np.array_equal(ctx.download(odd.ptr + 4, (B, nd)), ref)
    for t in (x, out, odd):
        t.free()


def test_apply_r_main_device_resident_and_host_loops_agree(tmp_path):
    """python -m ganrev.apply_r (apply_r.lua:25-193 without the image writing): the device-resident run - embed_dev, the search on the device
    tables - and the --host run - forwardBatched per chunk as apply_r.lua spells it - leave the same arrays, bit for bit."""
    import json
    from ganrev import apply_r
    outs = {}
    for mode in ("device", "host"):
        d = tmp_path / mode
        args = ["--synthetic", "1x16x16x8", "--nbImages", "700", "--batchSize", "64", "--writeTo", str(d), "--quiet"] + (["--host"] if mode == "host" else [])
        summary = apply_r.main(args)
        assert summary["path"] == mode and summary["anomalies"] == int(np.floor(700 * 0.15)) and sum(summary["cluster_sizes"]) > 0
        outs[mode] = {f: np.load(d / f) for f in ("attributes.npy", "attributes_fixer.npy", "similar_by_attributes.npy", "similar_by_pixels.npy",
                                                  "fixed_faces.npy", "anomaly_distances.npy", "cluster_centroids.npy", "variations.npy")}
        assert json.load(open(d / "summary.json"))["nbImages"] == 700
    for f, a in outs["device"].items():
        assert np.array_equal(a, outs["host"][f]), f"{f}: device-resident and host loops differ"
    assert outs["device"]["similar_by_attributes.npy"].shape == (5, 100) and outs["device"]["variations.npy"].shape == (8, 16, 1, 16, 16)


@pytest.mark.parametrize("dims,nd,B", [((1, 32, 32), 32, 12), ((3, 64, 64), 100, 6), ((1, 32, 32), 32, 130)])
def test_evaluate_mode_operand_ready_chain_vs_oracle(ctx, oracle, dims, nd, B):
    """evaluate()-mode R (apply_r.lua:120-153) in f16x3: every convolution after the first takes its input operand-ready - written by the
    epilogue of the convolution before it (scaled by the weight-norm bound of launch_eval_bound, the true maximum tracked beside it) or by the
    pooling stage's pipeline kernel.  Checked: the kernels that ran; the result against the oracle and against the same forward with the
    hand-over switched off (fp32 tensors between the stages: round 3's path); a batch whose images carry a few pixels 300 x larger than the
    rest (the bound then overshoots the next tensors' maxima by many bits: what is left of the 22-bit split must still do)."""
    from ganrev import models, synth
    prev = ctx.conv_mode(); ctx.set_conv_mode("f16x3")
    ctx.set_tuning("p16_min_tiles", 1)
    try:
        R = models.create_R(dims, nd); synth.init_params(R, 14)
        oR = oracle.from_model(R, dims); oR.set_training(False)
        R.evaluate()
        x = synth.uniform((B,) + dims, 15, 0, 1)
        spiky = x.copy()
        spiky[:, :, 5, 7] *= 300.0; spiky[0, 0, 20, 20] = -900.0
        for name, inp in (("plain", x), ("spiky", spiky)):
            ctx.set_tuning("eval_p16", 1)
            ctx.set_timing(2)
            got = R.forward(inp).copy()
            kt = ctx.kernel_times(); ctx.set_timing(0)
            count = lambda prefix: sum(k["launches"] for k in kt if k["kernel"].startswith(prefix))
            names = sorted((k["kernel"], k["launches"]) for k in kt if k["kernel"].startswith("conv3x3") or k["kernel"].startswith("post_forward"))
            assert count("conv3x3_fewin_p16o_kernel") == 1, names
            assert count("conv3x3_p16_quad_po_kernel") == 3, names        # conv2, conv4, conv5 hand over from their epilogue
            assert count("conv3x3_p16_quad_kernel") + count("conv3x3_p16_k32_kernel") == 2, names      # conv3, conv6: raw output for the pooling stage (16x16 planes: the 32-channel-chunk kernel)
            assert count("post_forward_g8_kernel") == 1 and count("conv3x3_split") == 0, names
            ctx.set_tuning("eval_p16", 0)
            ctrl = R.forward(inp).copy()
            ref = oR.forward(inp)
            scale = max(1.0, float(np.abs(ref).max()))
            assert_close(got, ref, TOL * scale, f"{name}: operand-ready chain vs oracle")
            assert_close(ctrl, ref, TOL * scale, f"{name}: fp32 hand-over vs oracle")
            assert_close(got, ctrl, 2e-5 * scale, f"{name}: the two hand-overs against each other")
    finally:
        ctx.set_tuning("eval_p16", 1)
        ctx.set_tuning("p16_min_tiles", 128)
        ctx.set_conv_mode(prev)
