"""Shared helpers for the parity tests: build a ganrev model and its oracle twin with identical parameters,
identical BN running stats and identical dropout noise."""
import numpy as np

from ganrev import synth

TOL = 1e-4   # north_star: recovered noise vectors and G images within 1e-4 fp32


def dropout_modules(model):
    return [m for m in model.leaves() if m.typename in ("nn.Dropout", "nn.SpatialDropout")]


def inject_noise(model, onet, B, seed, training=True):
    """Same keep flags into the HIP net (for its next forward) and the oracle net."""
    for m in dropout_modules(model):
        if not (training or getattr(m, "always_on", False)):
            continue
        li = onet.layer_index[id(m)]
        n = onet.mask_size(li, B)
        keep = synth.bernoulli_keep((n,), seed * 131 + li, m.p)
        onet.set_mask(li, keep)
        model.setNoise(m, keep)


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0


def assert_close(a, b, tol=TOL, what=""):
    d = maxdiff(a, b)
    assert np.isfinite(d) and d <= tol, f"{what}: max |diff| = {d:.3e} > {tol:g}"


def rel_close(a, b, rtol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1e-30, float(np.max(np.abs(b))))
    d = float(np.max(np.abs(a - b))) / scale
    assert d <= rtol, f"{what}: max rel diff = {d:.3e} > {rtol:g}"


GAPS = (1e-5, 2e-6)     # tried in this order by the tests that search for a well-conditioned input


def pools_well_conditioned(model, onet, B, gap=2e-6):
    """True when no 2x2 max-pool window of the oracle's last forward has its two largest values closer than `gap`
    without being exactly equal.  A near-tie lets fp32 rounding pick a different argmax on the two sides, which re-routes
    a gradient: a legitimate difference that an element-wise tolerance cannot express.  The forward error of a correct fp32
    implementation at the pooling inputs is 2-3e-6 (all arithmetic modes), so two values closer than about 1e-5 can swap
    order; small cases can afford that gap (see pick_well_conditioned), a million-window tensor always holds a few such
    pairs and only the 2e-6 default is satisfiable there."""
    leaves = model.leaves()
    for i, m in enumerate(leaves):
        if m.typename != "nn.SpatialMaxPooling":
            continue
        li = onet.layer_index[id(leaves[i - 1])]
        x = onet.layer_output(li)
        pi = onet.layer_index[id(m)]
        n_out = onet.layer_output(pi).size
        # input is [B, C, H, W] with H*W = 4 * (out H*W)
        assert x.size // n_out == 4
        dims = onet.pool_in_dims[pi]
        xr = x.reshape(B, dims[0], dims[1] // 2, 2, dims[2] // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(-1, 4)
        srt = np.sort(xr, axis=1)
        g = srt[:, 3] - srt[:, 2]
        if np.any((g > 0) & (g < gap)):
            return False
    return True
