"""Shared by the data-parallel parity tests of the HIP path (tests/test_gpu_dp.py) and their rank worker
(tests/dp_rank_worker.py): the cases, the seeded inputs, and the oracle side (ONE process, the GLOBAL batch, BatchNorm
evaluated in `world` groups = per-rank batch statistics).

Order under test (train_r.lua:147-165): MSE normalised by the GLOBAL element count -> SUM of the ranks'
flat gradients -> L2 penalty -> clamp -> Adam, identically on every replica."""
import numpy as np

WORLD = 2
# (name, dims, noise dim, per-rank batch): cfg2 geometry at B = 2 x 8, cfg3 geometry at B = 2 x 4
CASES = [("cfg2-geometry", (1, 32, 32), 32, 8), ("cfg3-geometry", (3, 64, 64), 100, 4)]
# cfg4's rank count (BASELINE configs[3]: 8 x MI355X): (world, name, dims, noise dim, per-rank batch) - eight shards of 4 images at cfg3's geometry (global batch 32;
# cfg4's own 8 x 512 needs the eight devices), and eight of 4 at cfg2's.  Four rows per rank, not two: nn.BatchNormalization over two rows normalises them to
# exactly +-1 and its backward amplifies rounding noise ~50x.  VERDICT round 5, item 1.
WORLD_CASES = [(WORLD,) + c for c in CASES] + [(8, "cfg3-geometry-P8", (3, 64, 64), 100, 4), (8, "cfg2-geometry-P8", (1, 32, 32), 32, 4)]
MODES = ("f32", "bf16x6", "f16x3")
T_STEP = 1


def make_models(dims, nd, seed=31):
    from ganrev import models, synth
    G = models.create_G(dims, nd); synth.init_params(G, seed)
    R = models.create_R(dims, nd); synth.init_params(R, seed + 1)
    return G, R


def global_inputs(R, layer_of, mask_size_of, dims, nd, per_rank, seed=400, world=WORLD):
    """Noise of the global batch and the dropout keep flags of the global batch per dropout layer ({layer: uint8[GB * per]})."""
    from ganrev import synth
    GB = per_rank * world
    noise = synth.normal((GB, nd), seed)
    masks = {}
    for m in R.leaves():
        if m.typename in ("nn.Dropout", "nn.SpatialDropout"):
            li = layer_of(m)
            masks[li] = synth.bernoulli_keep((mask_size_of(li, GB),), seed * 131 + li, m.p)
    return noise, masks


def shard(arr, rank, world=WORLD):
    """Rows of `rank` of a tensor whose leading extent is the global batch (flattened masks included: per-sample blocks are
    contiguous in NCHW)."""
    per = arr.shape[0] // world
    return arr[rank * per:(rank + 1) * per]


def oracle_grouped_step(oracle, oG, oR, noise, masks, theta0, hyper, dev_index=None, dev_y=None, R=None, max_flips=64, report=None, mode=None, groups=WORLD):
    """The reference side: one train_r.lua:138-170 iteration on the global batch with BatchNorm in WORLD groups.
    Returns dict(images, preds, loss, raw = the un-penalised flat gradient (the SUM the ranks must reproduce),
    grads = penalised + clamped, theta, m, v).  dev_index / dev_y: the device's pool argmax and raw conv outputs of the pooled
    stages, concatenated over the shards - adopted as in every other gradient parity test (helpers.adopt_device_argmax)."""
    from helpers import adopt_device_argmax, release_argmax
    GB = noise.shape[0]
    oG.set_training(False)
    images = oG.forward(noise)                                    # train_r.lua:139
    oR.set_bn_groups(groups)          # WORLD: per-rank batch statistics (the default DP semantics); 1: synchronised BatchNorm
    oR.set_training(True)
    oR.params[...] = theta0
    release_argmax(R, oR)
    for li, k in masks.items():
        oR.set_mask(li, k)
    oR.zero_grads()
    preds = np.array(oR.forward(images), copy=True)               # :146
    if dev_index is not None:
        adopt_device_argmax(R, oR, GB, max_flips, dev_index=dev_index, dev_y=dev_y, groups=groups, report=report, mode=mode)
        for li, k in masks.items():
            oR.set_mask(li, k)
        oR.zero_grads()
        preds = np.array(oR.forward(images), copy=True)
    loss, dfdo = oracle.mse(preds, noise)                         # :147,150  (global normaliser)
    oR.backward(images, dfdo, want_gin=False)                     # :151
    release_argmax(R, oR)
    raw = oR.grads.copy()
    theta, g = theta0.copy(), raw.copy()
    m, v = np.zeros_like(raw), np.zeros_like(raw)
    oracle.penalty_clamp_adam(theta, g, m, v, hyper, T_STEP)      # :153-170
    return dict(images=images, preds=preds, loss=loss, raw=raw, grads=g, theta=theta, m=m, v=v)
