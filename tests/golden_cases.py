"""Definition of the golden cases (shared by tests/golden/make_golden.py, the oracle pin test and the GPU test)."""
import numpy as np

from ganrev import models, synth

STRIDE = 97   # gradients are stored sub-sampled (every 97th entry) plus their sum and abs-sum

CASES = {
    "R_gray8_train": dict(kind="R", dims=(1, 8, 8), nd=6, B=4, method="normal", fixer=False, training=True, seed=11),
    "R_gray32_train": dict(kind="R", dims=(1, 32, 32), nd=32, B=4, method="normal", fixer=False, training=True, seed=12),
    "R_rgb16_uniform_fixer_train": dict(kind="R", dims=(3, 16, 16), nd=10, B=4, method="uniform", fixer=True, training=True, seed=13),
    "R_rgb64_eval": dict(kind="R", dims=(3, 64, 64), nd=100, B=2, method="normal", fixer=False, training=False, seed=14),
    "G_gray32": dict(kind="G", dims=(1, 32, 32), nd=32, B=4, training=False, seed=21),
    "G_rgb64": dict(kind="G", dims=(3, 64, 64), nd=100, B=2, training=False, seed=22),
    "step_gray32": dict(kind="step", dims=(1, 32, 32), nd=32, B=8, seed=31, steps=3),
    "search_10k_32": dict(kind="search", N=10000, d=32, k=100, needles=[100, 200, 300, 400, 500], seed=41),   # apply_r.lua:145,170-172,267
    "search_pixel": dict(kind="search", N=400, d=3 * 16 * 16, k=50, needles=[100, 200], seed=42),             # apply_r.lua:308-314
}


def build_case(case):
    """-> (model, in_dims, x, masks{layer_index: keep})"""
    dims, nd, B, seed = case["dims"], case["nd"], case["B"], case["seed"]
    if case["kind"] == "G":
        model = models.create_G(dims, nd); in_dims = (nd, 1, 1)
        x = synth.normal((B, nd), seed + 1)
    else:
        model = models.create_R(dims, nd, case["method"], case["fixer"]); in_dims = dims
        x = synth.uniform((B,) + tuple(dims), seed + 1, 0, 1)
    synth.init_params(model, seed)
    descs, index = model._descs(tuple(in_dims))
    masks = {}
    d = tuple(in_dims)
    for m in model.leaves():
        ds, nd_ = m.desc(d)
        if m.typename in ("nn.Dropout", "nn.SpatialDropout") and (case["training"] or getattr(m, "always_on", False)):
            n = B * (int(np.prod(d)) if m.typename == "nn.Dropout" else d[0])
            masks[index[id(m)]] = synth.bernoulli_keep((n,), seed * 131 + index[id(m)], m.p)
        d = nd_
    return model, in_dims, x, masks


def summarize_grads(g):
    g = np.asarray(g)
    return dict(grads_sample=g[::STRIDE].copy(), grads_sum=np.float64(g.astype(np.float64).sum()),
                grads_abs=np.float64(np.abs(g.astype(np.float64)).sum()))


def step_inputs(case, t):
    B, nd = case["B"], case["nd"]
    return synth.normal((B, nd), case["seed"] * 10 + t)


def run_oracle_case(oracle, case):
    kind = case["kind"]
    if kind in ("R", "G"):
        model, in_dims, x, masks = build_case(case)
        net = oracle.from_model(model, in_dims)
        net.set_training(case["training"])
        for li, k in masks.items():
            net.set_mask(li, k)
        out = net.forward(x)
        res = dict(out=out)
        if kind == "R" and case["training"]:
            gy = synth.normal(out.shape, case["seed"] + 9) * np.float32(0.1)
            net.zero_grads()
            res["gin"] = net.backward(x, gy)
            res.update(summarize_grads(net.grads)); res["grads_full"] = net.grads.copy()
            rm = [net.bn_running(i) for i in range(net.n_bn())]
            res["running_mean0"], res["running_var0"] = rm[0][0].copy(), rm[0][1].copy()
        return res
    if kind == "step":
        dims, nd, B = case["dims"], case["nd"], case["B"]
        G = models.create_G(dims, nd); synth.init_params(G, case["seed"])
        R = models.create_R(dims, nd); synth.init_params(R, case["seed"] + 1)
        oG, oR = oracle.from_model(G, (nd, 1, 1)), oracle.from_model(R, dims)
        m = np.zeros(oR.n_params, np.float32); v = np.zeros_like(m)
        losses, psum, pabs = [], [], []
        from helpers import inject_noise
        for t in range(1, case["steps"] + 1):
            inject_noise(R, oR, B, case["seed"] + t)
            loss, img = oracle.train_r_step(oG, oR, step_inputs(case, t), oracle.GoHyper(), m, v, t, want_images=(t == 1))
            if t == 1:
                first_img, first_grads = img, oR.grads.copy()
            losses.append(loss); psum.append(oR.params.astype(np.float64).sum()); pabs.append(np.abs(oR.params.astype(np.float64)).sum())
        return dict(losses=np.array(losses), params_sum=np.array(psum), params_abs=np.array(pabs), images1=first_img,
                    **{k + "1": v_ for k, v_ in summarize_grads(first_grads).items()})
    if kind == "search":
        emb = synth.normal((case["N"], case["d"]), case["seed"])
        emb[case["needles"][0] + 7] = emb[case["needles"][0]]       # an exact duplicate: tie broken by index
        idx, sc = oracle.cosine_topk(emb, case["needles"], case["k"])
        return dict(idx=idx, scores=sc)
    raise ValueError(kind)


# ---------------------------------------------------------------------------------------------------------------------
# D-network cases (tests/golden/golden_v2_dnet.npz): the module types models.lua:272-337 adds
DCASES = {
    "Dchain_small": dict(kind="chain", dims=(2, 8, 8), B=3, seed=51),          # conv3 + PReLU + 5x5 conv + PReLU + SpatialDropout + MaxPool + Linear + PReLU + Linear + Sigmoid
    "D2_gray16": dict(kind="D2", dims=(1, 16, 16), B=2, seed=52),              # models.create_D2: nn.Concat(2) of the 5x5 and the 3x3 tower
}


def build_dcase(case):
    from ganrev import nn
    dims, seed = case["dims"], case["seed"]
    if case["kind"] == "chain":
        model = (nn.Sequential().add(nn.SpatialConvolution(2, 6, 3, 3, 1, 1, 1, 1)).add(nn.PReLU())
                 .add(nn.SpatialConvolution(6, 4, 5, 5, 1, 1, 2, 2)).add(nn.PReLU()).add(nn.SpatialDropout(0.25)).add(nn.SpatialMaxPooling(2, 2))
                 .add(nn.View(4 * 4 * 4)).add(nn.Linear(64, 5)).add(nn.PReLU()).add(nn.Linear(5, 1)).add(nn.Sigmoid()))
    else:
        model = models.create_D2(dims, seed=seed)
    synth.init_params(model, seed)
    for k, m in enumerate(m for m in model.leaves() if m.typename == "nn.PReLU"):
        m.weight[0] = np.float32(0.25 + 0.03125 * k)
    x = synth.uniform((case["B"],) + tuple(dims), seed + 1, 0, 1)
    return model, x


def run_oracle_dcase(oracle, case, model=None):
    """Training-mode forward + backward of the oracle twin (one oracle net per compiled part, helpers.OracleGraph) with seeded
    dropout noise.  -> (results, twin, model)"""
    from helpers import OracleGraph, inject_noise
    if model is None:
        model, x = build_dcase(case)
    else:
        _, x = build_dcase(case)
    B = case["B"]
    model.training()
    og = OracleGraph(oracle, model, case["dims"]) if model._is_graph() else None
    pairs = og.pairs if og else [(model, oracle.from_model(model, case["dims"]))]
    for chunk, onet in pairs:
        onet.set_training(True)
        inject_noise(chunk, onet, B, case["seed"])
    twin = og if og else pairs[0][1]
    out = twin.forward(x)
    gy = synth.normal(out.shape, case["seed"] + 9)
    twin.zero_grads()
    gin = twin.backward(x, gy)
    grads = np.concatenate([o.grads for _, o in pairs])
    res = dict(out=np.array(out), gin=np.array(gin), **summarize_grads(grads))
    if grads.size < 4096:
        res["grads"] = grads.copy()
    return res, twin, model, pairs
