"""not-gpu: pin the CPU oracle.  The reference holds no golden vectors for this path (parity
unpinned vs Torch7), so the oracle is cross-checked operator by operator and net by net against an independent
float64 PyTorch-CPU evaluation of the same definitions, and by the known-answer checks below."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ganrev import models, synth
from helpers import assert_close, dropout_modules, maxdiff
from torch_twin import Twin

T = lambda a: torch.tensor(np.asarray(a, np.float64))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 3, 5, 6, 7), (3, 16, 8, 8, 8), (1, 1, 64, 32, 32), (2, 64, 3, 16, 16)])
def test_conv3_ops(oracle, B, Cin, Cout, H, W):
    x, w, b = synth.normal((B, Cin, H, W), 1), synth.uniform((Cout, Cin, 3, 3), 2, -0.3, 0.3), synth.uniform((Cout,), 3)
    gy = synth.normal((B, Cout, H, W), 4)
    xt, wt, bt = T(x).requires_grad_(True), T(w).requires_grad_(True), T(b).requires_grad_(True)
    y = F.conv2d(xt, wt, bt, padding=1)
    gx, gw, gb = torch.autograd.grad(y, (xt, wt, bt), T(gy))
    assert_close(oracle.conv3_forward(x, w, b), y.detach().numpy(), 2e-5, "conv forward")
    assert_close(oracle.conv3_backward_data(gy, w), gx.numpy(), 2e-5, "conv backward-data")
    ogw, ogb = oracle.conv3_backward_weight(x, gy)
    assert_close(ogw, gw.numpy(), 1e-4 * max(1, np.abs(gw.numpy()).max()), "conv backward-weight")
    assert_close(ogb, gb.numpy(), 1e-4 * max(1, np.abs(gb.numpy()).max()), "conv backward-bias")


@pytest.mark.parametrize("B,Cin,Cout,H,W,K", [(2, 3, 5, 6, 7, 5), (2, 16, 8, 16, 16, 5), (1, 4, 3, 9, 8, 7), (2, 3, 4, 5, 5, 3)])
def test_convk_ops(oracle, B, Cin, Cout, H, W, K):
    """nn.SpatialConvolution(.., K, K, 1, 1, (K-1)/2, (K-1)/2): the D network's 5x5 layer (models.lua:297)."""
    x, w, b = synth.normal((B, Cin, H, W), 1), synth.uniform((Cout, Cin, K, K), 2, -0.3, 0.3), synth.uniform((Cout,), 3)
    gy = synth.normal((B, Cout, H, W), 4)
    xt, wt, bt = T(x).requires_grad_(True), T(w).requires_grad_(True), T(b).requires_grad_(True)
    y = F.conv2d(xt, wt, bt, padding=(K - 1) // 2)
    gx, gw, gb = torch.autograd.grad(y, (xt, wt, bt), T(gy))
    assert_close(oracle.convk_forward(x, w, b), y.detach().numpy(), 5e-5, "KxK conv forward")
    assert_close(oracle.convk_backward_data(gy, w), gx.numpy(), 5e-5, "KxK conv backward-data")
    ogw, ogb = oracle.convk_backward_weight(x, gy, K)
    assert_close(ogw, gw.numpy(), 1e-4 * max(1, np.abs(gw.numpy()).max()), "KxK conv backward-weight")
    assert_close(ogb, gb.numpy(), 1e-4 * max(1, np.abs(gb.numpy()).max()), "KxK conv backward-bias")
    if K == 3:      # the window-size-generic loops agree with the 3x3 functions the golden fixtures pin
        assert np.array_equal(oracle.convk_forward(x, w, b), oracle.conv3_forward(x, w, b))


def test_conv3_known_answer(oracle):
    # identity kernel (centre tap = 1) reproduces the input; a shifted delta reproduces a zero-padded shift
    x = synth.normal((1, 1, 5, 5), 7)
    w = np.zeros((1, 1, 3, 3), np.float32); w[0, 0, 1, 1] = 1
    assert np.array_equal(oracle.conv3_forward(x, w, np.zeros(1, np.float32)), x)
    w[...] = 0; w[0, 0, 0, 2] = 1           # out[y,x] = in[y-1, x+1]  (cross-correlation, no flip)
    y = oracle.conv3_forward(x, w, np.zeros(1, np.float32))
    ref = np.zeros_like(x); ref[0, 0, 1:, :-1] = x[0, 0, :-1, 1:]
    assert np.array_equal(y, ref)


def test_linear_ops(oracle):
    x, w, b, gy = synth.normal((5, 37), 1), synth.uniform((11, 37), 2), synth.uniform((11,), 3), synth.normal((5, 11), 4)
    xt, wt, bt = T(x).requires_grad_(True), T(w).requires_grad_(True), T(b).requires_grad_(True)
    y = F.linear(xt, wt, bt)
    gx, gw, gb = torch.autograd.grad(y, (xt, wt, bt), T(gy))
    assert_close(oracle.linear_forward(x, w, b), y.detach().numpy(), 1e-5)
    ogx, ogw, ogb = oracle.linear_backward(x, gy, w)
    assert_close(ogx, gx.numpy(), 1e-5); assert_close(ogw, gw.numpy(), 1e-5); assert_close(ogb, gb.numpy(), 1e-5)


@pytest.mark.parametrize("shape", [(6, 5, 4, 4), (8, 12)])
def test_batchnorm_ops(oracle, shape):
    x = synth.normal(shape, 1) * np.float32(1.7) + np.float32(0.3)
    C = shape[1]
    gamma, beta, gy = synth.uniform((C,), 2, 0.5, 1.5), synth.uniform((C,), 3), synth.normal(shape, 4)
    rm, rv = synth.uniform((C,), 5, -0.2, 0.2), synth.uniform((C,), 6, 0.5, 1.5)
    rm_t, rv_t = T(rm).clone(), T(rv).clone()
    xt, gt, bt = T(x).requires_grad_(True), T(gamma).requires_grad_(True), T(beta).requires_grad_(True)
    y = F.batch_norm(xt, rm_t, rv_t, gt, bt, True, 0.1, 1e-5)
    gx, gg, gb = torch.autograd.grad(y, (xt, gt, bt), T(gy))
    orm, orv = rm.copy(), rv.copy()
    oy, sm, si = oracle.bn_forward_train(x, gamma, beta, orm, orv)
    assert_close(oy, y.detach().numpy(), 1e-5, "BN train forward")
    assert_close(orm, rm_t.numpy(), 1e-6, "running_mean"); assert_close(orv, rv_t.numpy(), 1e-6, "running_var (unbiased)")
    ogx, ogg, ogb = oracle.bn_backward_train(x, gy, gamma, sm, si)
    assert_close(ogx, gx.numpy(), 1e-5); assert_close(ogg, gg.numpy(), 1e-4); assert_close(ogb, gb.numpy(), 1e-4)
    ye = F.batch_norm(T(x), T(rm), T(rv), T(gamma), T(beta), False, 0.1, 1e-5)
    assert_close(oracle.bn_forward_eval(x, gamma, beta, rm, rv), ye.numpy(), 1e-5, "BN eval forward")


def _twin_and_oracle(oracle, model, in_dims, B, training, seed):
    descs, index = model._descs(tuple(in_dims))
    onet = oracle.from_model(model, in_dims)
    onet.set_training(training)
    masks = {}
    for m in dropout_modules(model):
        if training or getattr(m, "always_on", False):
            li = index[id(m)]
            keep = synth.bernoulli_keep((onet.mask_size(li, B),), seed * 131 + li, m.p)
            onet.set_mask(li, keep); masks[li] = keep
    running = [(m.running_mean.copy(), m.running_var.copy()) for m in model.leaves() if hasattr(m, "running_mean")]
    twin = Twin(descs, in_dims, model._flat_host(), running, training, masks)
    return onet, twin


@pytest.mark.parametrize("dims,nd,B,method,fixer", [((1, 8, 8), 6, 6, "normal", False), ((3, 16, 16), 10, 4, "uniform", True),
                                                    ((1, 32, 32), 32, 3, "normal", False)])
def test_R_net_forward_backward(oracle, dims, nd, B, method, fixer):
    R = models.create_R(dims, nd, method, fixer); synth.init_params(R, 3)
    onet, twin = _twin_and_oracle(oracle, R, dims, B, True, 7)
    x = synth.uniform((B,) + dims, 5, 0, 1)
    out = onet.forward(x)
    assert_close(out, twin.forward(x), 2e-5, "R forward (training)")
    gy = synth.normal(out.shape, 9) * np.float32(0.1)
    onet.zero_grads(); onet.backward(x, gy)
    ref = twin.backward(gy)
    assert_close(onet.grads, ref, 2e-4 * max(1.0, np.abs(ref).max()), "R flat gradient vs float64 autograd")
    # evaluate mode (apply_r.lua path)
    onet2, twin2 = _twin_and_oracle(oracle, R, dims, B, False, 7)
    assert_close(onet2.forward(x), twin2.forward(x), 2e-5, "R forward (evaluate)")


@pytest.mark.parametrize("dims,nd,B", [((1, 32, 32), 32, 3), ((3, 16, 16), 10, 2)])
def test_G_net_forward(oracle, dims, nd, B):
    G = models.create_G(dims, nd); synth.init_params(G, 2)
    onet, twin = _twin_and_oracle(oracle, G, (nd, 1, 1), B, False, 1)
    z = synth.normal((B, nd), 8)
    img = onet.forward(z)
    assert img.shape == (B,) + dims and img.min() >= 0 and img.max() <= 1
    assert_close(img, twin.forward(z), 1e-5, "G images")


def test_fullconv_and_leakyrelu_extras(oracle):
    from ganrev import nn
    m = nn.Sequential().add(nn.SpatialFullConvolution(4, 6)).add(nn.LeakyReLU(0.333))
    synth.init_params(m, 4)
    onet, twin = _twin_and_oracle(oracle, m, (4, 8, 8), 2, True, 1)
    x = synth.normal((2, 4, 8, 8), 3)
    out = onet.forward(x)
    assert_close(out, twin.forward(x), 1e-5, "SpatialFullConvolution + LeakyReLU forward")
    gy = synth.normal(out.shape, 5)
    onet.zero_grads(); onet.backward(x, gy)
    assert_close(onet.grads, twin.backward(gy), 1e-4, "SpatialFullConvolution gradients")


def test_prelu_and_5x5_net(oracle):
    """The D network's module types in one chain (models.lua:275-276,290): conv3 + PReLU + 5x5 conv + PReLU + SpatialDropout +
    MaxPool + Linear + PReLU + Sigmoid, against float64 autograd - including the gradient of each PReLU's one shared slope."""
    from ganrev import nn
    m = (nn.Sequential().add(nn.SpatialConvolution(2, 6, 3, 3, 1, 1, 1, 1)).add(nn.PReLU())
         .add(nn.SpatialConvolution(6, 4, 5, 5, 1, 1, 2, 2)).add(nn.PReLU()).add(nn.SpatialDropout(0.25)).add(nn.SpatialMaxPooling(2, 2))
         .add(nn.View(4 * 4 * 4)).add(nn.Linear(64, 5)).add(nn.PReLU()).add(nn.Linear(5, 1)).add(nn.Sigmoid()))
    synth.init_params(m, 6)
    slopes = [mod for mod in m.leaves() if mod.typename == "nn.PReLU"]
    assert all(float(mod.weight[0]) == 0.25 for mod in slopes)           # nn.PReLU's initial slope survives the init
    for k, mod in enumerate(slopes):
        mod.weight[0] = np.float32(0.25 + 0.1 * k)
    onet, twin = _twin_and_oracle(oracle, m, (2, 8, 8), 3, True, 2)
    x = synth.normal((3, 2, 8, 8), 3)
    out = onet.forward(x)
    assert_close(out, twin.forward(x), 1e-5, "conv + PReLU + 5x5 conv forward")
    gy = synth.normal(out.shape, 5)
    onet.zero_grads(); gin = onet.backward(x, gy)
    ref = twin.backward(gy)
    assert_close(onet.grads, ref, 1e-4 * max(1.0, np.abs(ref).max()), "gradients incl. the PReLU slopes")
    off = 2 * 6 * 9 + 6
    assert abs(ref[off]) > 1e-3 and abs(onet.grads[off] - ref[off]) <= 1e-4 * max(1.0, abs(ref[off]))     # first PReLU's slope


def test_mse_and_adam(oracle):
    x, t = synth.normal((16, 32), 1), synth.normal((16, 32), 2)
    loss, g = oracle.mse(x, t)
    xt = T(x).requires_grad_(True)
    l = F.mse_loss(xt, T(t)); l.backward()
    assert abs(loss - l.item()) < 1e-7 and maxdiff(g, xt.grad.numpy()) < 1e-7
    # Adam: three steps against torch.optim.Adam (eps outside the bias correction in both 2016 optim.adam and PyTorch<=1.x
    # differs by the sqrt(bc2) placement of eps: compare against a float64 transcription of optim/adam.lua instead)
    n = 1000
    th = synth.normal((n,), 3).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    th64, m64, v64 = th.astype(np.float64), np.zeros(n), np.zeros(n)
    h = oracle.GoHyper()
    for t_ in (1, 2, 3):
        g = synth.normal((n,), 10 + t_) * np.float32(3.0)
        g64 = g.astype(np.float64) + 1e-4 * th64                     # L2 (train_r.lua:158-159), L1 = 0
        g64 = np.clip(g64, -1, 1)                                    # train_r.lua:163-165
        m64 = 0.9 * m64 + 0.1 * g64; v64 = 0.999 * v64 + 0.001 * g64 * g64
        step = 1e-3 * np.sqrt(1 - 0.999 ** t_) / (1 - 0.9 ** t_)
        th64 = th64 - step * m64 / (np.sqrt(v64) + 1e-8)
        gg = g.copy()
        oracle.penalty_clamp_adam(th, gg, m, v, h, t_)
        assert maxdiff(gg, g64) < 1e-6 and maxdiff(th, th64) < 1e-6 and maxdiff(m, m64) < 1e-6 and maxdiff(v, v64) < 1e-6


def test_cosine_known_answers(oracle):
    a = np.array([1, 2, 3, 4], np.float32)
    assert abs(oracle.cosine_similarity(a, a) - 1) < 1e-6 and abs(oracle.cosine_similarity(a, -a) + 1) < 1e-6
    assert oracle.cosine_similarity(a, np.zeros(4, np.float32)) == 0.0          # eps keeps the zero vector finite
    e = np.eye(4, dtype=np.float32)
    assert oracle.cosine_similarity(e[0], e[1]) == 0.0
    emb = synth.normal((500, 16), 1); emb[77] = emb[10]
    idx, sc = oracle.cosine_topk(emb, [10], 5)
    assert list(idx[0][:2]) == [10, 77] and sc[0][0] == sc[0][1]                # tie -> index ascending
    full = np.array([oracle.cosine_similarity(emb[10], emb[j]) for j in range(500)])
    order = np.lexsort((np.arange(500), -full.astype(np.float64)))
    assert np.array_equal(idx[0], order[:5])
    ref = (emb @ emb[10]) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(emb[10]))
    assert maxdiff(full, ref) < 1e-6


def test_bn_groups_model_per_rank_statistics(oracle):
    """BN evaluated in P groups == P independent nets each fed one shard (what P data-parallel ranks compute)."""
    dims, nd, B, P = (1, 8, 8), 4, 8, 2
    R = models.create_R(dims, nd); synth.init_params(R, 5)
    x = synth.uniform((B,) + dims, 1, 0, 1)
    big = oracle.from_model(R, dims); big.set_bn_groups(P)
    keeps = {}
    for m in dropout_modules(R):
        li = big.layer_index[id(m)]
        keeps[li] = synth.bernoulli_keep((big.mask_size(li, B),), li, m.p); big.set_mask(li, keeps[li])
    out = big.forward(x)
    for r in range(P):
        small = oracle.from_model(R, dims)
        for li, k in keeps.items():
            per = k.size // B
            small.set_mask(li, k[r * (B // P) * per:(r + 1) * (B // P) * per])
        assert np.array_equal(small.forward(x[r * (B // P):(r + 1) * (B // P)]), out[r * (B // P):(r + 1) * (B // P)])


def test_kmeans_and_cosine_assign_vs_float64():
    """unsup.kmeans (apply_r.lua:198, restated from memory) and the nearest-centroid loop (apply_r.lua:205-217) of the oracle
    against an independent float64 numpy evaluation of the same algorithm; an empty cluster keeps its centroid."""
    from oracle import oracle
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3000, 32)).astype(np.float32)
    k, niter = 20, 15
    c0 = rng.standard_normal((k, 32)).astype(np.float32)
    c0 /= np.linalg.norm(c0, axis=1, keepdims=True)
    c0[7] = 100.0                                     # far away: never wins a row
    cent, tot, lab = oracle.kmeans(x, k, niter, c0)
    c = c0.astype(np.float64).copy(); totn = np.zeros(k)
    xd = x.astype(np.float64)
    for _ in range(niter):
        lbl = np.argmax(xd @ c.T - 0.5 * (c ** 2).sum(1), axis=1)
        for j in range(k):
            if (lbl == j).any():
                c[j] = xd[lbl == j].mean(0)
        totn += np.bincount(lbl, minlength=k)
    assert np.array_equal(lab, lbl)
    assert np.abs(cent - c).max() < 1e-6 and np.array_equal(tot, totn.astype(np.float32))
    assert np.array_equal(cent[7], c0[7]) and tot[7] == 0
    for take_min in (True, False):
        la, si = oracle.cosine_assign(x, cent, take_min)
        S = (xd @ cent.astype(np.float64).T) / np.sqrt((xd ** 2).sum(1, keepdims=True) * (cent.astype(np.float64) ** 2).sum(1))
        ref = S.argmin(1) if take_min else S.argmax(1)
        assert np.array_equal(la, ref)
        assert np.abs(si - (S.min(1) if take_min else S.max(1))).max() < 1e-6


def test_philox_known_answers(oracle):
    """The counter-based generator behind gr_fill_normal_dev / gr_fill_uniform_dev / the Dropout masks is Philox4x32-10 (Salmon et al.,
    SC'11).  The oracle's numpy restatement must reproduce the known-answer vectors published with the Random123 library
    (kat_vectors: counter words, key words -> output words); the device is then compared with the oracle value by value
    (tests/test_gpu_abi_behaviour.py::test_device_noise_equals_the_oracle_stream).  Also: distribution sanity of the two fills."""
    kat = [((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = oracle.philox4x32_10(*[np.array([v]) for v in ctr], *key)
        assert tuple(int(v[0]) for v in got) == want, (ctr, key)
    # vectorised call = element-wise calls
    i = np.arange(1000, dtype=np.uint64)
    many = oracle.philox4x32_10(i, 7, 0x6e6f6973, 0, 9, 0)
    one = oracle.philox4x32_10(np.array([123]), 7, 0x6e6f6973, 0, 9, 0)
    assert all(int(m[123]) == int(o[0]) for m, o in zip(many, one))
    z = oracle.fill_normal(1 << 18, 9)
    assert z.dtype == np.float32 and abs(z.mean()) < 1e-2 and abs(z.std() - 1) < 1e-2 and abs(((z ** 3).mean())) < 3e-2 and abs((z ** 4).mean() - 3) < 0.1
    u = oracle.fill_uniform(1 << 18, 9)
    assert -1 <= u.min() and u.max() < 1 and abs(u.mean()) < 1e-2 and abs(u.var() - 1 / 3) < 1e-2
    assert abs(oracle.dropout_keep(1 << 18, 0.5, 5, 1, 3).mean() - 0.5) < 5e-3 and abs(oracle.dropout_keep(1 << 18, 0.25, 5, 1, 3).mean() - 0.75) < 5e-3
