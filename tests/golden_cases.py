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
