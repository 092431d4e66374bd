"""Mirror of the analysis steps of the reference's apply_r.lua that sit on (or right next to) the hot path.

  embed                        apply_r.lua:145-153   images = G(noise); attributes = R(images)   (forwardBatched, batch 32)
  createSimilaritySearch       apply_r.lua:265-318   needle rows i*100, cosine top-n on recovered noise / on raw pixels
  cosineSimilarity             apply_r.lua:396-400
  fixFaces                     apply_r.lua:324-352   noise -> G -> image -> R_fixer -> noise -> G -> image
  detectAnomalies              apply_r.lua:355-390   1 - torch.dist(image, fixed image), lowest `threshold` share = anomalies
  createClusterImages          apply_r.lua:197-231   unsup.kmeans on the recovered noise, nearest-centroid pass, per-cluster lists

Everything image-writing (image.toDisplayTensor / image.save / colour conversion) is out of scope: these functions return
the tensors / index lists the reference would have rendered.
"""
import math

import numpy as np

from . import _lib as L
from .nn_utils import DeviceTensor, forwardBatched


def embed(model_g, model_r, noise, batchSize=32, model_r_fixer=None):
    """apply_r.lua:145-153.  Returns (images, attributes[, attributesFixer])."""
    model_g.evaluate()
    images = forwardBatched(model_g, noise, batchSize)                      # :146
    model_r.evaluate()
    attributes = forwardBatched(model_r, images, batchSize)                 # :152
    if model_r_fixer is None:
        return images, attributes
    model_r_fixer.evaluate()                                                # the fixer's first Dropout stays on (models.lua:402-405)
    return images, attributes, forwardBatched(model_r_fixer, images, batchSize)   # :153


def embed_dev(model_g, model_r, noise, batchSize=512, model_r_fixer=None, keep_images=False, dims=None):
    """apply_r.lua:145-153 resident on the GPU (gr_embed_dev): `noise` is a DeviceTensor [N x noiseDim] (nn_utils.createNoiseInputsDev);
    per chunk of batchSize rows  G:forward -> R:forward [-> R_fixer:forward], the recovered noise written straight into [N x nd] device
    tables.  -> (images or None, attributes[, attributesFixer]) as DeviceTensors.  The images are a chunk-sized intermediate unless
    keep_images (the pixel-wise search and fix-faces want them: N x C x H x W floats).  dims = (C, H, W) of the images; taken from
    model_g's compiled net when it has run before.  Same kernels, same chunking as forwardBatched with this batchSize: same bits."""
    ctx = noise.ctx
    model_g.evaluate(); model_r.evaluate()
    gnet = model_g.device_net(noise.shape[1:])
    dims = tuple(dims) if dims is not None else gnet.out_dims
    rs = [model_r] + ([model_r_fixer] if model_r_fixer is not None else [])
    if model_r_fixer is not None:
        model_r_fixer.evaluate()                                            # its first Dropout stays on (models.lua:402-405)
    rnets = [m.device_net(dims) for m in rs]
    N = noise.shape[0]
    if ctx.conv_mode() == "f16x3":
        for n in [gnet] + rnets:          # the *_dev calls are not range-guarded (include/ganrev.h): one synchronous scan per net
            n.range_guard_scan()
    images = DeviceTensor(ctx, (N,) + tuple(dims)) if keep_images else None
    attrs = [DeviceTensor(ctx, (N,) + L.Net._shape(n.out_dims)) for n in rnets]
    L.embed_dev(gnet, rnets, noise.ptr, N, batchSize, [a.ptr for a in attrs], images.ptr if images is not None else None)
    return (images,) + tuple(attrs)


def createSimilaritySearchDev(nbSimilarNeedles, nbShowMax, attributes, images=None):
    """apply_r.lua:265-318 on device-resident tables (DeviceTensors from embed_dev): the needles' rows are searched where the embeddings
    were written.  -> (idx_by_attributes, idx_by_pixels or None)."""
    N = attributes.shape[0]
    needles = np.array([i * 100 - 1 for i in range(1, nbSimilarNeedles + 1)], dtype=np.int64)
    if needles.max() >= N:
        raise IndexError(f"needle row {needles.max() + 1} out of range for {N} rows")
    n = min(nbShowMax, N)
    ctx = attributes.ctx
    by_attr, _ = ctx.cosine_topk(None, needles, n, emb_dev=attributes.ptr, n=N, d=attributes.size // N)
    by_pix = None
    if images is not None:
        by_pix, _ = ctx.cosine_topk(None, needles, n, emb_dev=images.ptr, n=N, d=images.size // N)
    return by_attr, by_pix


def cosineSimilarity(v1, v2):
    """apply_r.lua:396-400."""
    return L.default_context().cosine_similarity(v1, v2)


def createSimilaritySearch(nbSimilarNeedles, nbShowMax, images, attributes):
    """apply_r.lua:265-318.  -> (idx_by_attributes, idx_by_pixels): for needle i (row i*100, 1-based in the reference) the
    row indices (0-based here) of the min(nbShowMax, N) most similar rows, best first, ties by ascending index."""
    N = len(attributes)
    needles = np.array([i * 100 - 1 for i in range(1, nbSimilarNeedles + 1)], dtype=np.int64)   # face_i_idx = i*100 (1-based)
    if needles.max() >= N:
        raise IndexError(f"needle row {needles.max() + 1} out of range for {N} rows")           # the reference would index nil
    n = min(nbShowMax, N)
    ctx = L.default_context()
    by_attr, _ = ctx.cosine_topk(attributes, needles, n)                                        # similarityMeasureAttributes
    by_pix, _ = ctx.cosine_topk(np.asarray(images, np.float32).reshape(N, -1), needles, n)      # similarityMeasurePixelwise
    return by_attr, by_pix


def fixFaces(nbFixedImages, model_g, attributesFixer, batchSize=32):
    """apply_r.lua:344-351: the G(R_fixer(G(z))) images of the first nbFixedImages rows."""
    model_g.evaluate()
    return forwardBatched(model_g, attributesFixer[:nbFixedImages], batchSize)


def detectAnomalies(nbImagesCalculations, threshold, images, model_g, attributesFixer, batchSize=32):
    """apply_r.lua:355-390.  -> (distances, anomalyBelow, isAnomaly) with distances[i] = 1 - torch.dist(images[i], fixed[i]).
    The reference forwards each row in a batch of two (:361-363, BatchNorm needs a batch) — G is in evaluate() mode, so the
    result does not depend on the batch composition and the rows are forwarded in normal batches here."""
    n = nbImagesCalculations
    model_g.evaluate()
    fixed = forwardBatched(model_g, attributesFixer[:n], batchSize)
    dist = 1.0 - L.default_context().l2_distance_rows(images[:n], fixed)
    srt = np.sort(dist)                                                     # table.sort(distancesForSort)
    anomalyBelow = srt[max(int(math.floor(n * threshold)) - 1, 0)]          # distancesForSort[floor(#*threshold)]  (1-based)
    return dist, anomalyBelow, dist <= anomalyBelow


def initialCentroids(nbClusters, nDims, seed=1):
    """unsup.kmeans draws its initial centroids as `x.new(k, ndims):normal()` and divides each row by its norm.  Torch's
    Mersenne-twister stream is not reproducible here; the same distribution from the package's counter RNG."""
    from . import synth
    c = synth.normal((nbClusters, nDims), seed).astype(np.float32)
    return c / np.linalg.norm(c.astype(np.float64), axis=1, keepdims=True).astype(np.float32)


def createClusterImages(nbClusters, nbIterations, nbMaxPerCluster, images, attributes, centroids0=None, seed=1, closest=False):
    """apply_r.lua:197-231 without the image writing.  -> (centroids, counts, clusters, averageFaces): clusters[j] is the list
    of (row index, similarity) kept for cluster j, sorted like the reference (similarity descending, first nbMaxPerCluster);
    averageFaces[j] the mean image of those rows (zeros / NaN-free for an empty cluster: the reference divides by zero there
    and skips the cluster when saving, :243,:251).

    Reference behaviour preserved: a row joins the centroid with the MINIMUM cosine similarity (:207-214 keep `dist <
    minDist` of a similarity).  closest=True assigns to the most similar centroid instead."""
    attributes = np.asarray(attributes, np.float32)
    N, d = attributes.shape
    if centroids0 is None:
        centroids0 = initialCentroids(nbClusters, d, seed)
    ctx = L.default_context()
    centroids, counts, _ = ctx.kmeans(attributes, nbClusters, nbIterations, centroids0)            # :198
    label, sim = ctx.cosine_assign(attributes, centroids, take_min=not closest)                     # :205-217
    clusters, faces = [], []
    images = np.asarray(images, np.float32)
    for j in range(nbClusters):
        rows = np.nonzero(label == j)[0]
        order = np.argsort(-sim[rows], kind="stable")                                               # :221 (a[2] > b[2]); ties by row
        keep = rows[order][:nbMaxPerCluster]                                                        # :223-227
        clusters.append([(int(r), float(sim[r])) for r in keep])
        faces.append(images[keep].mean(axis=0, dtype=np.float64).astype(np.float32) if len(keep)
                     else np.zeros(images.shape[1:], np.float32))                                   # :233-243
    return centroids, counts, clusters, faces


# ---------------------------------------------------------------------------------------------------------------------
# apply_r.lua:25-193 main(): the whole analysis as one run.  Image writing (image.toDisplayTensor / image.save, apply_r.lua:136,
# 233-262, 283-300, 336-352, 374-390) is out of scope: what the reference would have rendered is written as arrays (.npy) and one
# summary.json under --writeTo.  G / R / R_fixer come from Torch7 checkpoints (ganrev.t7: train.lua:256, train_r.lua:234) or, with
# --synthetic, from random-initialised nets of the requested shape (a smoke run: no trained checkpoint exists in this repository).
def parse(argv=None):
    import argparse
    p = argparse.ArgumentParser(description="apply_r.lua options (apply_r.lua:13-23)")
    p.add_argument("--batchSize", type=int, default=32)                       # apply_r.lua:14 (the device-resident pipeline likes 512)
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--G", default="logs/adversarial.net")
    p.add_argument("--R", default="logs/r_3x32x32_nd32_normal.net")
    p.add_argument("--R_fixer", default="logs/r_3x32x32_nd32_normal_fixer.net")
    p.add_argument("--writeTo", default="r_results")
    p.add_argument("--nbImages", type=int, default=10000)                     # apply_r.lua:145
    p.add_argument("--synthetic", default="", help="CxHxWxND, e.g. 1x32x32x32: random-initialised G / R / R_fixer instead of checkpoints")
    p.add_argument("--host", action="store_true", help="the host-tensor loop (forwardBatched per chunk, as apply_r.lua spells it) instead of the device-resident pipeline")
    p.add_argument("--conv-mode", default="f16x3", choices=["f32", "bf16x6", "f16x3"])
    p.add_argument("--quiet", action="store_true")
    return p.parse_args(argv)


def main(argv=None):
    import json
    import os
    import time
    from . import models, nn_utils, synth
    OPT = parse(argv)
    ctx = L.Context(OPT.gpu) if OPT.gpu != int(os.environ.get("LOCAL_RANK", "0")) else L.default_context()
    ctx.set_conv_mode(OPT.conv_mode)
    say = (lambda *a: None) if OPT.quiet else print
    if OPT.synthetic:
        c, h, w, nd = (int(v) for v in OPT.synthetic.split("x"))
        dims, method = (c, h, w), "normal"
        MODEL_G = models.create_G(dims, nd, seed=OPT.seed); synth.init_params(MODEL_G, OPT.seed)
        MODEL_R = models.create_R(dims, nd, method, False, seed=OPT.seed + 1); synth.init_params(MODEL_R, OPT.seed + 1)
        MODEL_R_FIXER = models.create_R(dims, nd, method, True, seed=OPT.seed + 2); synth.init_params(MODEL_R_FIXER, OPT.seed + 2)
    else:
        from . import t7
        ck = t7.load_checkpoint(OPT.G)                                        # apply_r.lua:62-69
        MODEL_G, o = ck["G"], ck.get("opt", {})
        nd, method = int(o.get("noiseDim", 32)), o.get("noiseMethod", "normal")
        dims = (1 if o.get("colorSpace", "rgb") == "y" else 3, int(o.get("height", 32)), int(o.get("width", 32)))
        MODEL_R = t7.load_checkpoint(OPT.R)["R"]                              # :92-94
        MODEL_R_FIXER = t7.load_checkpoint(OPT.R_fixer)["R"]                  # :101-103
    for m in (MODEL_G, MODEL_R, MODEL_R_FIXER):
        m._ctx = ctx
        m.evaluate()
    MODEL_R_FIXER.manualSeed(OPT.seed)
    os.makedirs(OPT.writeTo, exist_ok=True)
    out = lambda name: os.path.join(OPT.writeTo, name)
    summary = dict(dims=list(dims), noiseDim=nd, noiseMethod=method, nbImages=OPT.nbImages, batchSize=OPT.batchSize, path="host" if OPT.host else "device")

    say("Varying components...")                                              # apply_r.lua:110-136
    nbSteps = 16
    steps = np.linspace(-1, 1, nbSteps) if method == "uniform" else np.linspace(-3, 3, nbSteps)
    noise1 = nn_utils.createNoiseInputs(1, nd, method, seed=OPT.seed)
    var_noise = np.repeat(noise1, nd * nbSteps, axis=0)
    for i in range(nd):
        var_noise[i * nbSteps:(i + 1) * nbSteps, i] = steps
    np.save(out("variations.npy"), forwardBatched(MODEL_G, var_noise, OPT.batchSize).reshape((nd, nbSteps) + tuple(dims)))

    say("Generating images, converting images to attributes...")             # :141-153
    N = OPT.nbImages
    t0 = time.perf_counter()
    dn = nn_utils.createNoiseInputsDev(ctx, N, nd, method, seed=OPT.seed + 1)      # utils/nn_utils.lua:39-51, drawn on the device (both paths: same noise)
    if OPT.host:
        noise = dn.numpy(); dn.free()
        images, attributes, attributesFixer = embed(MODEL_G, MODEL_R, noise, OPT.batchSize, MODEL_R_FIXER)
        by_attr, by_pix = createSimilaritySearch(5, 100, images, attributes) if N >= 500 else (None, None)
    else:
        di, da, df = embed_dev(MODEL_G, MODEL_R, dn, OPT.batchSize, MODEL_R_FIXER, keep_images=True, dims=dims)
        by_attr, by_pix = createSimilaritySearchDev(5, 100, da, di) if N >= 500 else (None, None)     # :170-172 on the tables where they were written
        noise, images, attributes, attributesFixer = dn.numpy(), di.numpy(), da.numpy(), df.numpy()
        for t in (dn, di, da, df):
            t.free()
    ctx.synchronize()
    summary["embed_and_search_seconds"] = round(time.perf_counter() - t0, 4)
    np.save(out("attributes.npy"), attributes); np.save(out("attributes_fixer.npy"), attributesFixer)
    if by_attr is not None:
        np.save(out("similar_by_attributes.npy"), by_attr); np.save(out("similar_by_pixels.npy"), by_pix)

    say("Clustering...")                                                      # :158-162
    centroids, counts, clusters, faces = createClusterImages(20, 15, 64 + 7, images, attributes, seed=OPT.seed)
    np.save(out("cluster_centroids.npy"), centroids); np.save(out("cluster_average_faces.npy"), np.stack(faces))
    summary["cluster_sizes"] = [len(c) for c in clusters]; summary["cluster_total_counts"] = [float(v) for v in counts]

    say("Fixing faces...")                                                    # :178-181
    nbFixed = min(512 + 16, N)
    np.save(out("fixed_faces.npy"), fixFaces(nbFixed, MODEL_G, attributesFixer, OPT.batchSize))

    say("Detecting anomalies...")                                             # :186-191
    nbCalc = min(1024, N)
    dist, below, is_anom = detectAnomalies(nbCalc, 0.15, images, MODEL_G, attributesFixer, OPT.batchSize)
    np.save(out("anomaly_distances.npy"), dist)
    summary.update(anomaly_below=float(below), anomalies=int(is_anom.sum()))
    json.dump(summary, open(out("summary.json"), "w"), indent=1)
    say("<apply_r> results in", OPT.writeTo, summary)
    return summary


if __name__ == "__main__":
    main()
