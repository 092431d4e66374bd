"""Mirror of the reference's train.lua main loop (train.lua:125-257): build or load G and D, then per epoch load N_epoch *
batchSize / 2 * D_iterations training images (train.lua:214-216), play adversarial.train on them and save
{D, G, opt, epoch} as a Torch7 checkpoint (train.lua:241-257) - the file train_r.lua:68 and apply_r.lua:62 read G from.

    python -m ganrev.train --epochs 5 --N_epoch 30 --batchSize 32 --save logs [--data images.npy] [--compat]

Same option names and defaults as train.lua:12-60 for what is mirrored.  The dataset loader, normalisation, plots / `display`
and image grids are out of scope: training images come from --data (an [N x C x H x W] float32 .npy
in [0, 1]) or, without it, from a synthetic generator, which is what makes the loop runnable here.

Two loops, as in ganrev.train_r:
  fast (default)  - adversarial.DeviceGame: every batch device-resident, parameters pulled to the host only before a save;
  --compat        - adversarial.train, the closures exactly as adversarial.lua:66-133 spells them.
"""
import argparse
import os
import time

import numpy as np

from . import _lib as L
from . import adversarial, models, synth, t7


def parse(argv=None):
    p = argparse.ArgumentParser(description="train.lua options (train.lua:12-60)")
    p.add_argument("--save", default="logs")                       # train.lua:13
    p.add_argument("--saveFreq", type=int, default=30)             # train.lua:14
    p.add_argument("--network", default="")                        # train.lua:15: continue from this checkpoint
    p.add_argument("--batchSize", type=int, default=32)
    p.add_argument("--N_epoch", type=int, default=30)
    p.add_argument("--epochs", type=int, default=1, help="epochs to play (train.lua runs until interrupted)")
    p.add_argument("--G_L1", type=float, default=0.0)
    p.add_argument("--G_L2", type=float, default=0.0)
    p.add_argument("--D_L1", type=float, default=0.0)
    p.add_argument("--D_L2", type=float, default=1e-4)
    p.add_argument("--D_iterations", type=int, default=1)
    p.add_argument("--G_iterations", type=int, default=1)
    p.add_argument("--D_clamp", type=float, default=1.0)
    p.add_argument("--G_clamp", type=float, default=5.0)
    p.add_argument("--D_optmethod", default="adam", help="sgd|adagrad|adadelta|adamax|adam|rmsprop (anything but adam plays the --compat game)")
    p.add_argument("--G_optmethod", default="adam")
    p.add_argument("--D_sgd_lr", type=float, default=0.02)
    p.add_argument("--G_sgd_lr", type=float, default=0.02)
    p.add_argument("--D_sgd_momentum", type=float, default=0.0)
    p.add_argument("--G_sgd_momentum", type=float, default=0.0)
    p.add_argument("--noiseDim", type=int, default=100)
    p.add_argument("--noiseMethod", default="normal", choices=["normal", "uniform"])
    p.add_argument("--height", type=int, default=32)
    p.add_argument("--width", type=int, default=32)
    p.add_argument("--colorSpace", default="gray", choices=["gray", "rgb"])
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--data", default="", help="[N x C x H x W] float32 .npy of training images; default: synthetic")
    p.add_argument("--compat", action="store_true")
    p.add_argument("--conv-mode", default="f16x3", choices=["f32", "bf16x6", "f16x3"])
    p.add_argument("--quiet", action="store_true")
    return p.parse_args(argv)


def synthetic_images(n, dims, seed):
    """Stand-in for DATASET.loadRandomImages (dataset.lua, out of scope): smooth blobs in [0, 1], different per call."""
    c, h, w = dims
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cy, cx = synth.uniform((n, 1, 1, 1), seed, 0.25 * h, 0.75 * h), synth.uniform((n, 1, 1, 1), seed + 1, 0.25 * w, 0.75 * w)
    r = synth.uniform((n, 1, 1, 1), seed + 2, 0.1 * h, 0.3 * h)
    img = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r)).astype(np.float32)
    return np.ascontiguousarray(np.broadcast_to(img, (n, c, h, w)), dtype=np.float32)


def save(OPT, env, epoch, quiet=True):
    """train.lua:236-257: logs/adversarial.net (the previous file moved to .old), {D, G, opt, epoch}."""
    filename = os.path.join(OPT.save, "adversarial.net")
    env.MODEL_G.pull_params(); env.MODEL_D.pull_params()             # current BatchNorm running statistics (and parameters) into the modules
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    if os.path.isfile(filename):
        os.replace(filename, filename + ".old")
    if not quiet:
        print("<trainer> saving network to %s" % filename)
    t7.save_checkpoint(filename, D=env.MODEL_D, G=env.MODEL_G, opt={k: v for k, v in vars(OPT).items() if isinstance(v, (int, float, str, bool))},
                       epoch=epoch)
    return filename


def main(argv=None):
    OPT = parse(argv)
    dims = (3 if OPT.colorSpace == "rgb" else 1, OPT.height, OPT.width)
    ctx = L.default_context()
    ctx.set_conv_mode(OPT.conv_mode)
    epoch0 = 1
    if OPT.network:                                                   # train.lua:125-140
        ck = t7.load_checkpoint(OPT.network)
        if "_unconverted" in ck:
            raise L.GanrevError(f"{OPT.network}: {ck['_unconverted']}")
        MODEL_D, MODEL_G, epoch0 = ck["D"], ck["G"], int(ck.get("epoch", 0)) + 1      # train.lua:113  EPOCH = tmp.epoch + 1
    else:                                                             # train.lua:143,160
        MODEL_D = models.create_D(dims, True, OPT.seed)
        MODEL_G = models.create_G(dims, OPT.noiseDim, True, OPT.seed + 1)
    env = adversarial.make_env(MODEL_G, MODEL_D, dims, **{k: getattr(OPT, k) for k in
                               ("batchSize", "N_epoch", "noiseDim", "noiseMethod", "G_L1", "G_L2", "D_L1", "D_L2", "D_iterations", "G_iterations",
                                "D_clamp", "G_clamp", "D_optmethod", "G_optmethod", "seed", "D_sgd_lr", "G_sgd_lr", "D_sgd_momentum", "G_sgd_momentum")})
    env.EPOCH = epoch0
    data = np.load(OPT.data).astype(np.float32) if OPT.data else None
    only_adam = OPT.D_optmethod == "adam" and OPT.G_optmethod == "adam"      # the fused device update is Adam's; the rest are host mirrors
    game = None if (OPT.compat or not only_adam) else adversarial.DeviceGame(env)
    N_epoch = OPT.N_epoch if OPT.N_epoch > 0 else 100                 # adversarial.lua:42-45: N_epoch <= 0 means 100 batches
    D_it, G_it = max(0, OPT.D_iterations), max(0, OPT.G_iterations)   # 0 iterations freeze that net (adversarial.lua:127,168 loop zero times)
    # a continued run must not replay the first epochs' noise: the counters start where epoch0 - 1 finished epochs left them
    # (the reference's Torch RNG is not restored from a checkpoint either; Adam's state restarts empty, as train.lua does)
    done = (epoch0 - 1) * N_epoch * (D_it + G_it)
    env.noise_counter = getattr(env, "noise_counter", 0) + done
    if game is not None:
        game.noise_counter += done
    # train.lua:214 multiplies OPT.N_epoch itself; with N_epoch <= 0 that loads nothing while adversarial.lua:42-45 still runs 100
    # batches and indexes past the loaded examples - here the 100 batches get their images
    nbLoad = (N_epoch * OPT.batchSize // 2) * D_it
    cursor, last, t0, images = (epoch0 - 1) * nbLoad, None, time.perf_counter(), 0
    for _ in range(OPT.epochs):
        if data is not None:
            idx = (cursor + np.arange(nbLoad)) % len(data); cursor += nbLoad
            TRAIN_DATA = data[idx]
        else:
            TRAIN_DATA = synthetic_images(nbLoad, dims, OPT.seed * 7919 + env.EPOCH * 3)
        if game is None:
            adversarial.train(env, TRAIN_DATA, quiet=OPT.quiet)       # train.lua:229
            last = (env.last_losses["D"][-1], env.last_losses["G"][-1])
        else:
            per = (OPT.batchSize // 2) * D_it                         # real images one batch consumes: batchSize/2 per D iteration (adversarial.lua:139-149)
            for b in range(N_epoch):
                want = b == N_epoch - 1
                res = game.batch(TRAIN_DATA[b * per:(b + 1) * per], want_loss=want)
                if want:
                    last = res
        images += N_epoch * OPT.batchSize * G_it
        if not OPT.quiet:
            print("<trainer> epoch %d: loss D=%.4f G=%.4f" % (env.EPOCH, last[0], last[1]))
        if env.EPOCH % OPT.saveFreq == 0:                             # train.lua:232-234
            if game is not None:
                game.sync_to_host()
            save(OPT, env, env.EPOCH, OPT.quiet)
        env.EPOCH += 1
    if game is not None:
        game.sync_to_host()
    path = save(OPT, env, env.EPOCH - 1, OPT.quiet)                   # train.lua:209-211 "Last epoch reached."
    if not OPT.quiet:
        print("<trainer> %.1f generated images/s" % (images / (time.perf_counter() - t0)))
    return dict(path=path, last_losses=last, epoch=env.EPOCH - 1, env=env)


if __name__ == "__main__":
    main()
