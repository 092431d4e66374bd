"""Mirror of the reference's train_r.lua: train R to recover the noise vectors G was fed (train_r.lua:131-225).

    python -m ganrev.train_r --G g.npz --batchSize 32 --nbBatches 2000 --R_L2 1e-4 --R_clamp 1 --seed 1

Same options and defaults as train_r.lua:12-29 (plots / `display` / `--noplot` are out of scope).  The G "checkpoint" is an
.npz written by save_model() below (the reference's .t7 format is Torch7's own serialisation, out of scope); without --G a
random-initialised create_G3 of the requested shape is used, which is what bench.py measures.

Two loops:
  fast (default)  — everything resident on the GPU, one gr_train_r_step per iteration (ganrev.parallel.DeviceTrainer);
  --compat        — the loop exactly as train_r.lua:138-170 spells it: host tensors, fevalR closure, optim.adam.
"""
import argparse
import time

import numpy as np

from . import _lib as L
from . import models, nn, nn_utils, optim, synth
from .parallel import DeviceTrainer


def save_model(path, model, opt):
    flat = model._flat[0] if model._flat is not None else model._flat_host()
    bn = [m for m in model.leaves() if hasattr(m, "running_mean")]
    np.savez(path, params=flat, opt=np.array(list(opt.items()), dtype=object),
             **{f"rm{i}": m.running_mean for i, m in enumerate(bn)}, **{f"rv{i}": m.running_var for i, m in enumerate(bn)})


def load_params(path, model):
    z = np.load(path, allow_pickle=True)
    flat, _ = model.getParameters()
    flat[...] = z["params"]
    for i, m in enumerate([m for m in model.leaves() if hasattr(m, "running_mean")]):
        m.running_mean[...] = z[f"rm{i}"]; m.running_var[...] = z[f"rv{i}"]
    return dict(z["opt"].tolist())


def parse(argv=None):
    p = argparse.ArgumentParser(description="train_r.lua options (train_r.lua:12-29)")
    p.add_argument("--batchSize", type=int, default=32)
    p.add_argument("--nbBatches", type=int, default=2000)          # README.md:103 (train_r.lua:15 default -1: "<0 is infinite")
    p.add_argument("--save", default="logs")                       # a directory (train_r.lua:13,231) - or one explicit .net/.t7/.npz file
    p.add_argument("--saveFreq", type=int, default=2000)           # train_r.lua:19
    p.add_argument("--continue", dest="continue_", default="")     # train_r.lua:26,101-104
    p.add_argument("--G", default="")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--threads", type=int, default=8)
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--R_L1", type=float, default=0.0)
    p.add_argument("--R_L2", type=float, default=1e-4)
    p.add_argument("--R_clamp", type=float, default=1.0)
    p.add_argument("--fixer", action="store_true")
    p.add_argument("--noiseDim", type=int, default=32)
    p.add_argument("--noiseMethod", default="normal", choices=["normal", "uniform"])
    p.add_argument("--height", type=int, default=32)
    p.add_argument("--width", type=int, default=32)
    p.add_argument("--channels", type=int, default=1)
    p.add_argument("--compat", action="store_true")
    p.add_argument("--conv-mode", default="f16x3", choices=["f32", "bf16x6", "f16x3"])
    p.add_argument("--quiet", action="store_true")
    return p.parse_args(argv)


def main(argv=None):
    OPT = parse(argv)
    dims = (OPT.channels, OPT.height, OPT.width)
    ctx = L.Context(OPT.gpu) if OPT.gpu != int(__import__("os").environ.get("LOCAL_RANK", "0")) else L.default_context()
    ctx.set_conv_mode(OPT.conv_mode)
    if OPT.G and OPT.G.endswith((".net", ".t7")):
        # a checkpoint in Torch7's own format (train.lua:256 {G=..., opt=...}): train_r.lua:68-75 takes G and the image
        # geometry / noise settings from it
        from . import t7
        ck = t7.load_checkpoint(OPT.G)
        MODEL_G = ck["G"]
        o = ck.get("opt", {})
        OPT.noiseDim, OPT.noiseMethod = int(o.get("noiseDim", OPT.noiseDim)), o.get("noiseMethod", OPT.noiseMethod)
        OPT.height, OPT.width = int(o.get("height", OPT.height)), int(o.get("width", OPT.width))
        OPT.channels = 1 if o.get("colorSpace", "rgb") == "y" else 3
        dims = (OPT.channels, OPT.height, OPT.width)
    else:
        MODEL_G = models.create_G(dims, OPT.noiseDim, seed=OPT.seed)
        if OPT.G:
            load_params(OPT.G, MODEL_G)                                  # train_r.lua:68-75 (this package's npz form)
        else:
            synth.init_params(MODEL_G, OPT.seed)
    MODEL_G.evaluate()                                                   # train_r.lua:70
    if OPT.continue_:                                                    # train_r.lua:101-104  MODEL_R = torch.load(OPT.continue).R
        if OPT.continue_.endswith((".net", ".t7")):
            from . import t7
            MODEL_R = t7.load_checkpoint(OPT.continue_)["R"]
        else:
            MODEL_R = models.create_R(dims, OPT.noiseDim, OPT.noiseMethod, OPT.fixer, seed=OPT.seed)
            load_params(OPT.continue_, MODEL_R)
            if MODEL_R._flat is not None:                                # module arrays are views of the loaded flat storage
                MODEL_R._flat = None
    else:
        MODEL_R = models.create_R(dims, OPT.noiseDim, OPT.noiseMethod, OPT.fixer, seed=OPT.seed)   # train_r.lua:106
    MODEL_G._ctx = MODEL_R._ctx = ctx
    losses = []
    opt_table = {k.rstrip("_"): v for k, v in vars(OPT).items() if isinstance(v, (int, float, str, bool))}

    def save():
        """train_r.lua:227-235: torch.save(<OPT.save>/r_CxHxW_ndN_<method>[_fixer].net, {R=MODEL_R, opt=OPT}).  An --save that names a
        .net/.t7/.npz file is written as that file instead."""
        if not OPT.save:
            return None
        MODEL_R.pull_params()                   # parameters AND BatchNorm running statistics live on the device
        target = str(OPT.save)
        if not target.endswith((".net", ".t7", ".npz")):
            import os
            os.makedirs(target, exist_ok=True)
            target = os.path.join(target, "r_%dx%dx%d_nd%d_%s%s.net" % (dims[0], dims[1], dims[2], OPT.noiseDim, OPT.noiseMethod,
                                                                        "_fixer" if OPT.fixer else ""))
        if target.endswith(".npz"):
            save_model(target, MODEL_R, opt_table)
        else:
            from . import t7
            t7.save_checkpoint(target, R=MODEL_R, opt=opt_table)
        if not OPT.quiet:
            print("Saving networks...", target)
        return target

    def batches():
        i = 1
        while OPT.nbBatches < 0 or i <= OPT.nbBatches:                   # train_r.lua:131-135: "<0 is infinite"
            yield i
            i += 1
    if OPT.compat:
        CRITERION_R = nn.MSECriterion()                                  # :119
        PARAMETERS_R, GRAD_PARAMETERS_R = MODEL_R.getParameters()        # :122
        state = {}                                                       # OPTSTATE = {adam={R={}}}  :125
        MODEL_R.manualSeed(OPT.seed)
        for batchIdx in batches():
            noise = nn_utils.createNoiseInputs(OPT.batchSize, OPT.noiseDim, OPT.noiseMethod, seed=OPT.seed * 100003 + batchIdx)
            images = MODEL_G.forward(noise).copy()                       # :139

            def fevalR(x):
                GRAD_PARAMETERS_R[...] = 0                               # :143
                predsByR = MODEL_R.forward(images.copy()).copy()         # :146
                f = CRITERION_R.forward(predsByR, noise)                 # :147
                df_do = CRITERION_R.backward(predsByR, noise)            # :150
                MODEL_R.backward(images, df_do)                          # :151
                if OPT.R_L1 != 0 or OPT.R_L2 != 0:                       # :154-160
                    f += OPT.R_L1 * np.abs(PARAMETERS_R).sum() + OPT.R_L2 * float(np.dot(PARAMETERS_R, PARAMETERS_R)) / 2
                    GRAD_PARAMETERS_R[...] += np.sign(PARAMETERS_R) * np.float32(OPT.R_L1) + PARAMETERS_R * np.float32(OPT.R_L2)
                if OPT.R_clamp != 0:                                     # :163-165
                    np.clip(GRAD_PARAMETERS_R, -OPT.R_clamp, OPT.R_clamp, out=GRAD_PARAMETERS_R)
                return f, GRAD_PARAMETERS_R
            MODEL_R.training()
            optim.adam(fevalR, PARAMETERS_R, state, model=MODEL_R)       # :170
            losses.append(CRITERION_R.output)
            if not OPT.quiet:
                print("[batch %d of %d (%.2f%%)] loss R=%.4f" % (batchIdx, OPT.nbBatches, 100 * batchIdx / OPT.nbBatches, CRITERION_R.output))
            if batchIdx % OPT.saveFreq == 0:                             # :185-187
                save()
    else:
        MODEL_G.forward(synth.normal((2, OPT.noiseDim), 1))              # compile both nets
        MODEL_R.training(); MODEL_R.forward(synth.uniform((2,) + dims, 2, 0, 1)); MODEL_R.push_params()
        MODEL_R._net.set_seed(OPT.seed); MODEL_R._net.adam_reset()
        tr = DeviceTrainer(ctx, MODEL_G._net, MODEL_R._net, L.Hyper(l1=OPT.R_L1, l2=OPT.R_L2, clamp=OPT.R_clamp), OPT.batchSize,
                           noise_method=OPT.noiseMethod)
        t0 = time.perf_counter()
        nb = 0
        for batchIdx in batches():
            nb = batchIdx
            tr.new_noise(OPT.seed * 100003 + batchIdx)
            loss = tr.step(want_loss=True)
            losses.append(loss)
            if not OPT.quiet:
                print("[batch %d of %d (%.2f%%)] loss R=%.4f" % (batchIdx, OPT.nbBatches, 100 * batchIdx / OPT.nbBatches, loss))
            if batchIdx % OPT.saveFreq == 0:                             # :185-187
                save()
        if not OPT.quiet:
            print("<trainer> Last batch reached. %.1f images/s" % (OPT.batchSize * nb / (time.perf_counter() - t0)))
    save()                                                               # :136 "Last batch reached" -> save()
    return MODEL_G, MODEL_R, losses


if __name__ == "__main__":
    main()
