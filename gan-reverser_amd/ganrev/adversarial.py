"""Mirror of the reference's GAN game, adversarial.lua:1-205 (driven by train.lua:125-330) - SURVEY.md 8f rank 4.

adversarial.train(env, trainData) plays one epoch: per batch, D is updated on half a batch of real and half a batch of
generated images (fevalD, adversarial.lua:66-100), then G is updated through D (fevalG_on_D, adversarial.lua:104-133: G forward,
D forward, BCE against "real", D backward to the images, G backward from D's gradInput).  The reference keeps its state in Lua
globals (OPT, MODEL_D, MODEL_G, CRITERION, PARAMETERS_*, GRAD_PARAMETERS_*, OPTSTATE, CONFUSION, EPOCH ...); here the same
names live on one `env` object (make_env builds it the way train.lua:125-200 does).

Everything numeric runs on the GPU through libganrev.so: G and the compiled parts of D (ganrev.nn), nn.BCECriterion
(gr_bce_host), optim.adam (gr_adam_step).  The penalty / clamp lines are host numpy on the flat vectors, as they are Torch
tensor ops on the host-visible flat vectors in the reference.
"""
import types

import numpy as np

from . import _lib as L
from . import nn, nn_utils, optim

Y_GENERATOR = 0          # train.lua:67-68
Y_NOT_GENERATOR = 1


def clamp(gradParameters, clampValue):
    """adversarial.lua:8-12"""
    if clampValue != 0:
        np.clip(gradParameters, -clampValue, clampValue, out=gradParameters)


def l1(parameters, gradParameters, lossValue, l1weight):
    """adversarial.lua:14-20.  The reference misspells its own argument in the gradient line (`l1Weight`, a nil global), so a
    non-zero weight raises there; every shipped configuration has L1 = 0 (train.lua:29,31).  Implemented as evidently meant."""
    if l1weight != 0:
        lossValue = lossValue + l1weight * float(np.abs(parameters).sum(dtype=np.float64))
        gradParameters += np.sign(parameters) * np.float32(l1weight)
    return lossValue


def l2(parameters, gradParameters, lossValue, l2weight):
    """adversarial.lua:22-28"""
    if l2weight != 0:
        lossValue = lossValue + l2weight * float(np.dot(parameters.astype(np.float64), parameters.astype(np.float64))) / 2
        gradParameters += parameters * np.float32(l2weight)
    return lossValue


def make_env(MODEL_G, MODEL_D, IMG_DIMENSIONS, **opt):
    """The globals train.lua:125-200 sets up, with its option defaults (train.lua:27-38)."""
    OPT = types.SimpleNamespace(batchSize=32, N_epoch=30, noiseDim=100, noiseMethod="normal", G_L1=0.0, G_L2=0.0, D_L1=0.0, D_L2=1e-4,
                                D_iterations=1, G_iterations=1, D_clamp=1.0, G_clamp=5.0, D_optmethod="adam", G_optmethod="adam", seed=1,
                                D_sgd_lr=0.02, G_sgd_lr=0.02, D_sgd_momentum=0.0, G_sgd_momentum=0.0)
    for k, v in opt.items():
        if not hasattr(OPT, k):
            raise L.GanrevError(f"unknown option '{k}'")
        setattr(OPT, k, v)
    env = types.SimpleNamespace(OPT=OPT, MODEL_G=MODEL_G, MODEL_D=MODEL_D, IMG_DIMENSIONS=tuple(IMG_DIMENSIONS), EPOCH=1,
                                Y_GENERATOR=Y_GENERATOR, Y_NOT_GENERATOR=Y_NOT_GENERATOR)
    env.CRITERION = nn.BCECriterion()                                           # train.lua:173
    env.PARAMETERS_D, env.GRAD_PARAMETERS_D = MODEL_D.getParameters()           # train.lua:176-177
    env.PARAMETERS_G, env.GRAD_PARAMETERS_G = MODEL_G.getParameters()
    env.CONFUSION = np.zeros((2, 2), np.int64)                                  # train.lua:180 optim.ConfusionMatrix: [predicted][target]
    env.OPTSTATE = {k: {"D": {}, "G": {}} for k in ("adagrad", "adadelta", "adamax", "adam", "rmsprop")}      # train.lua:183-193
    env.OPTSTATE["sgd"] = {"D": {"learningRate": OPT.D_sgd_lr, "momentum": OPT.D_sgd_momentum},
                           "G": {"learningRate": OPT.G_sgd_lr, "momentum": OPT.G_sgd_momentum}}
    env.noise_counter = 0
    MODEL_D.training(); MODEL_G.training()                                      # train.lua:133-134
    return env


def _noise(env, N):
    env.noise_counter += 1
    return nn_utils.createNoiseInputs(N, env.OPT.noiseDim, env.OPT.noiseMethod, seed=env.OPT.seed * 100003 + env.noise_counter)


def createImages(env, N):
    """NN_UTILS.createImages(N, false) (utils/nn_utils.lua:57-89): MODEL_G:forward on fresh noise, OPT.batchSize rows at a time."""
    noise = _noise(env, N)
    out = None
    for lo in range(0, N, env.OPT.batchSize):
        gen = env.MODEL_G.forward(noise[lo:lo + env.OPT.batchSize]).copy()
        if out is None:
            out = np.empty((N,) + gen.shape[1:], np.float32)
        out[lo:lo + gen.shape[0]] = gen
    return out


def make_fevalD(env, inputs, targets):
    """adversarial.lua:66-100: f(X) and df/dX of the discriminator on the batch (inputs, targets)."""
    def fevalD(x):
        if x is not env.PARAMETERS_D:
            env.PARAMETERS_D[...] = x
        env.GRAD_PARAMETERS_D[...] = 0                                          # :74
        outputs = env.MODEL_D.forward(inputs)                                   # :79
        f = env.CRITERION.forward(outputs, targets.reshape(outputs.shape))      # :80
        df_do = env.CRITERION.backward(outputs, targets.reshape(outputs.shape)) # :83
        env.MODEL_D.backward(inputs, df_do)                                     # :84
        f = l1(env.PARAMETERS_D, env.GRAD_PARAMETERS_D, f, env.OPT.D_L1)        # :86-88
        f = l2(env.PARAMETERS_D, env.GRAD_PARAMETERS_D, f, env.OPT.D_L2)
        clamp(env.GRAD_PARAMETERS_D, env.OPT.D_clamp)
        for i in range(outputs.shape[0]):                                       # :91-96
            c = 1 if outputs[i][0] > 0.5 else 0
            env.CONFUSION[c, int(targets[i])] += 1
        return f, env.GRAD_PARAMETERS_D
    return fevalD


def make_fevalG_on_D(env, noiseInputs, targets):
    """adversarial.lua:104-133: f(X) and df/dX of the generator, rated by D."""
    def fevalG_on_D(x):
        if x is not env.PARAMETERS_G:
            env.PARAMETERS_G[...] = x
        env.GRAD_PARAMETERS_G[...] = 0                                          # :110
        samples = env.MODEL_G.forward(noiseInputs).copy()                       # :113
        outputs = env.MODEL_D.forward(samples)                                  # :114
        f = env.CRITERION.forward(outputs, targets.reshape(outputs.shape))      # :115
        df_samples = env.CRITERION.backward(outputs, targets.reshape(outputs.shape))   # :119
        env.MODEL_D.backward(samples, df_samples)                               # :120
        df_do = env.MODEL_D.gradInput                                           # :121  MODEL_D.modules[1].gradInput
        env.MODEL_G.backward(noiseInputs, df_do)                                # :124
        f = l1(env.PARAMETERS_G, env.GRAD_PARAMETERS_G, f, env.OPT.G_L1)        # :126-128
        f = l2(env.PARAMETERS_G, env.GRAD_PARAMETERS_G, f, env.OPT.G_L2)
        clamp(env.GRAD_PARAMETERS_G, env.OPT.G_clamp)
        return f, env.GRAD_PARAMETERS_G
    return fevalG_on_D


def _optimize(env, which, feval, params, model):
    """adversarial.lua:156-171 / 183-198: the optimiser chosen with --D_optmethod / --G_optmethod on the host flat vectors.  adam (the
    default) updates through the fused device kernel; the other five are host mirrors of the optim rock (ganrev/optim.py)."""
    method = getattr(env.OPT, which + "_optmethod")
    if method == "adam":
        return optim.adam(feval, params, env.OPTSTATE["adam"][which], model=model)
    if method not in optim.METHODS:
        raise L.GanrevError(f"Unknown optimizer method '{method}' chosen for {which}.")        # adversarial.lua:170, 197
    return optim.METHODS[method](feval, params, env.OPTSTATE[method][which])      # the host vector is authoritative: the next forward uploads it


def train(env, trainData, quiet=True):
    """adversarial.lua:37-205.  trainData: [N x C x H x W] real images; N >= N_epoch * batchSize/2 * D_iterations (train.lua:214)."""
    OPT = env.OPT
    batchSize, batchesPerEpoch = OPT.batchSize, OPT.N_epoch
    if batchesPerEpoch <= 0:
        batchesPerEpoch = 100                                                   # :42
    exampleForDIdx = 0
    nbExamplesD = nbExamplesG = 0
    losses = {"D": [], "G": []}
    if not quiet:
        print("<trainer> Epoch #%d [batchSize = %d]" % (env.EPOCH, batchSize))
    for batchIdx in range(1, batchesPerEpoch + 1):
        inputs = np.empty((batchSize,) + env.IMG_DIMENSIONS, np.float32)        # :54-56
        targets = np.empty(batchSize, np.float32)
        for _ in range(OPT.D_iterations):                                       # (1) update D  :139-174
            half = batchSize // 2
            if exampleForDIdx + half > len(trainData):
                raise IndexError("trainData exhausted (adversarial.lua:146 indexes past the loaded examples)")
            inputs[:half] = trainData[exampleForDIdx:exampleForDIdx + half]     # :141-149
            targets[:half] = env.Y_NOT_GENERATOR
            exampleForDIdx += half
            inputs[half:2 * half] = createImages(env, half)                     # :152-157
            targets[half:2 * half] = env.Y_GENERATOR
            _, fs = _optimize(env, "D", make_fevalD(env, inputs, targets), env.PARAMETERS_D, env.MODEL_D)
            losses["D"].append(fs[0])
            nbExamplesD += inputs.shape[0]
        for _ in range(OPT.G_iterations):                                       # (2) update G  :178-201
            noiseInputs = _noise(env, batchSize)
            targets[...] = env.Y_NOT_GENERATOR
            _, fs = _optimize(env, "G", make_fevalG_on_D(env, noiseInputs, targets), env.PARAMETERS_G, env.MODEL_G)
            losses["G"].append(fs[0])
            nbExamplesG += noiseInputs.shape[0]
    if not quiet:
        print("Trained G on: %d | Trained D on: %d" % (nbExamplesG, nbExamplesD))
        print(env.CONFUSION)
    tV = float(np.trace(env.CONFUSION)) / max(1, int(env.CONFUSION.sum()))     # :201 CONFUSION.totalValid
    env.CONFUSION[...] = 0
    env.last_losses = losses
    return tV


# ---------------------------------------------------------------------------------------------------------------------
# Fast mode: the same game with images, gradients, parameters and the Adam state resident on the GPU (what ganrev.train_r's
# DeviceTrainer / gr_train_r_step is to train_r.lua:138-170).  The containers stay host code that enqueues work: every compiled
# part runs through gr_net_forward_dev / gr_net_backward_dev, nn.Concat joins / slices / sums with gr_copy2d_dev / gr_add_dev,
# nn.BCECriterion is gr_bce_dev, and adversarial.l1 / l2 / clamp + optim.adam are the fused gr_adam_step of each part.  Nothing
# is copied to the host inside a batch except, on request, the two loss values.
class _DevGraph:
    """Device-resident executor of a ganrev model that runs as compiled parts (nn.Sequential.parts(), nn.Concat)."""

    def __init__(self, ctx, model):
        self.ctx, self.model, self.bufs = ctx, model, {}
        self.nets = [ch._net for ch, _, _ in model._param_chunks()]
        self.plan = self._plan(model)

    def _plan(self, node):
        if isinstance(node, nn.Concat):
            if node.dimension != 2:
                raise L.GanrevError("device-resident nn.Concat: only nn.Concat(2) of [batch x features] outputs (models.lua:293)")
            return ("concat", [self._plan(b) for b in node.modules])
        if node._is_graph():
            return ("seq", [self._plan(p) for p in node.parts()])
        if node._net is None:
            raise L.GanrevError("compile the model first (one forward)")
        return ("net", node._net)

    def _buf(self, key, n_floats):
        cur = self.bufs.get(key)
        if cur is None or cur[1] < n_floats:
            if cur is not None:
                self.ctx.free(cur[0])
            cur = (self.ctx.malloc(4 * int(n_floats)), int(n_floats))
            self.bufs[key] = cur
        return cur[0]

    @staticmethod
    def _vol(d):
        return int(d[0]) * int(d[1]) * int(d[2])

    def out_features(self, plan=None):
        plan = plan or self.plan
        if plan[0] == "net":
            return self._vol(plan[1].out_dims)
        if plan[0] == "seq":
            return self.out_features(plan[1][-1])
        return sum(self.out_features(p) for p in plan[1])

    def in_features(self, plan=None):
        plan = plan or self.plan
        if plan[0] == "net":
            return self._vol(plan[1].in_dims)
        return self.in_features(plan[1][0])

    def forward(self, x_dev, B, plan=None):
        plan = plan or self.plan
        self.bufs[(id(plan), "x")] = x_dev
        if plan[0] == "net":
            return plan[1].forward_dev(x_dev, B)
        if plan[0] == "seq":
            for p in plan[1]:
                x_dev = self.forward(x_dev, B, p)
            return x_dev
        total = self.out_features(plan)
        cat = self._buf((id(plan), "cat"), B * total)
        lo = 0
        for p in plan[1]:
            o, k = self.forward(x_dev, B, p), self.out_features(p)
            self.ctx.copy2d(cat + 4 * lo, total, o, k, B, k)
            lo += k
        return cat

    def backward(self, g_dev, B, want_gin, plan=None):
        """gradOutput (device) -> gradInput (device pointer, or None when not wanted); accumulates every part's parameter gradient"""
        plan = plan or self.plan
        x_dev = self.bufs[(id(plan), "x")]
        if plan[0] == "net":
            gin = self._buf((id(plan), "gin"), B * self._vol(plan[1].in_dims)) if want_gin else None
            plan[1].backward_dev(x_dev, g_dev, B, gin)
            return gin
        if plan[0] == "seq":
            for i in range(len(plan[1]) - 1, -1, -1):
                g_dev = self.backward(g_dev, B, want_gin or i > 0, plan[1][i])
            return g_dev
        total, lo, acc = self.out_features(plan), 0, None
        nin = B * self.in_features(plan)
        for j, p in enumerate(plan[1]):
            k = self.out_features(p)
            gs = self._buf((id(plan), "gslice", j), B * k)
            self.ctx.copy2d(gs, k, g_dev + 4 * lo, total, B, k)
            gi = self.backward(gs, B, want_gin, p)
            if want_gin:
                if acc is None:
                    acc = gi                      # the first branch's own gradInput buffer holds the sum
                else:
                    self.ctx.add(acc, gi, nin)
            lo += k
        return acc

    def zero_grads(self):
        for n in self.nets:
            n.zero_grads()

    def close(self):
        for key, v in list(self.bufs.items()):
            if isinstance(v, tuple):           # (pointer, size): a buffer this executor allocated ("x" entries are borrowed pointers)
                self.ctx.free(v[0])
            del self.bufs[key]

    def adam_step(self, hyper, t):
        for n in self.nets:
            n.adam_step(hyper, t)


class DeviceGame:
    """adversarial.lua:139-201 per batch, device-resident.  game = DeviceGame(env); game.batch(real_half_batch) per batch;
    game.sync_to_host() before anything reads env.PARAMETERS_* or the host-side modules again (save, evaluate, compat mode)."""

    def __init__(self, env):
        self.env = env
        OPT, G, D = env.OPT, env.MODEL_G, env.MODEL_D
        if OPT.D_optmethod != "adam" or OPT.G_optmethod != "adam":
            raise L.GanrevError("DeviceGame: only the default optimizer 'adam' (train.lua:37-38); the other methods run in adversarial.train")
        self.ctx = G._context()
        B = OPT.batchSize
        # Compile every net with one small forward (parameters uploaded) WITHOUT side effects on the models: the
        # reference has no such step, so the BatchNorm running statistics a training-mode forward of two samples would write - and
        # which evaluate()-mode users of G (train_r, apply_r) and the saved checkpoint would then carry - are put back afterwards.
        img = G.forward(nn_utils.createNoiseInputs(2, OPT.noiseDim, OPT.noiseMethod, seed=1))
        D.forward(img)
        self.gnet = G._net
        self.dg = _DevGraph(self.ctx, D)
        for model in (G, D):
            for chunk, _, _ in model._param_chunks():
                bi = 0
                for mod in chunk.leaves():
                    if hasattr(mod, "running_mean"):
                        chunk._net.set_bn_running(bi, mod.running_mean, mod.running_var)     # the modules still hold the pre-compile values
                        bi += 1
        for n in [self.gnet] + self.dg.nets:
            n.set_training(True)
            n.adam_reset()
        self.t = {"D": 0, "G": 0}
        self.npix = int(np.prod(env.IMG_DIMENSIONS))
        m = self.ctx.malloc
        self.inputs, self.targets, self.ones = m(4 * B * self.npix), m(4 * B), m(4 * B)
        self.noise, self.df, self.loss = m(4 * B * OPT.noiseDim), m(4 * B), m(16)
        half = B // 2
        self.ctx.upload(np.concatenate([np.full(half, Y_NOT_GENERATOR, np.float32), np.full(B - half, Y_GENERATOR, np.float32)]), self.targets)
        self.ctx.upload(np.full(B, Y_NOT_GENERATOR, np.float32), self.ones)
        self.hyper_d = L.Hyper(l1=OPT.D_L1, l2=OPT.D_L2, clamp=OPT.D_clamp)
        self.hyper_g = L.Hyper(l1=OPT.G_L1, l2=OPT.G_L2, clamp=OPT.G_clamp)
        self.noise_counter = 0

    def _fill_noise(self, N, host=None):
        nd = self.env.OPT.noiseDim
        if host is not None:
            self.ctx.upload(np.ascontiguousarray(host, np.float32), self.noise)
            return
        self.noise_counter += 1
        seed = self.env.OPT.seed * 100003 + self.noise_counter
        if self.env.OPT.noiseMethod == "uniform":
            self.ctx.fill_uniform(self.noise, N * nd, seed)
        else:
            self.ctx.fill_normal(self.noise, N * nd, seed)

    def _read_loss(self):
        return float(self.ctx.download(self.loss, (1,), np.float64)[0])

    GUARD_PERIOD = 64       # batches between two f16x3 range-guard scans of the parameters (gr_range_guard_scan_params)

    def batch(self, real, noise_d=None, noise_g=None, want_loss=False):
        """One batch of adversarial.lua:139-201: OPT.D_iterations updates of D, each on batchSize/2 fresh real images + batchSize/2
        generated ones, then OPT.G_iterations updates of G through D.  real: [D_iterations * batchSize/2 x C x H x W] host array (the
        one unavoidable upload).  noise_*: host noise to use instead of device-generated noise (parity tests): arrays of
        [D_iterations * batchSize/2 x noiseDim] / [G_iterations * batchSize x noiseDim].  Returns the LAST (loss_D, loss_G)."""
        OPT, B = self.env.OPT, self.env.OPT.batchSize
        half = B // 2
        ctx = self.ctx
        # `for k=1,OPT.D_iterations` (adversarial.lua:127,168): 0 iterations run zero times - that phase's net stays frozen
        nD, nG = max(0, int(OPT.D_iterations)), max(0, int(OPT.G_iterations))
        real = np.ascontiguousarray(real, np.float32).reshape(-1, self.npix)
        if real.shape[0] < nD * half:
            raise IndexError("trainData exhausted (adversarial.lua:146 indexes past the loaded examples): "
                             f"{real.shape[0]} images for {nD} D iterations of {half}")
        self.batches = getattr(self, "batches", 0) + 1
        if self.ctx.conv_mode() == "f16x3" and (self.batches - 1) % self.GUARD_PERIOD == 0:
            for n in [self.gnet] + self.dg.nets:      # the *_dev calls below are unguarded: sampled parameter scan, as gr_train_r_step does
                n.range_guard_scan()
        loss_d = loss_g = None
        for k in range(nD):
            # (1) D on half real, half generated (adversarial.lua:141-157, fevalD :66-100)
            ctx.upload(real[k * half:(k + 1) * half], self.inputs)
            self._fill_noise(half, None if noise_d is None else np.asarray(noise_d).reshape(nD, half, -1)[k])
            fake = self.gnet.forward_dev(self.noise, half)
            ctx.copy2d(self.inputs + 4 * half * self.npix, self.npix, fake, self.npix, half, self.npix)
            self.dg.zero_grads()
            out = self.dg.forward(self.inputs, B)
            ctx.bce_dev(out, self.targets, B, self.loss, self.df)
            self.dg.backward(self.df, B, False)
            self.t["D"] += 1
            self.dg.adam_step(self.hyper_d, self.t["D"])
            if want_loss and k == nD - 1:
                loss_d = self._read_loss()
        for k in range(nG):
            # (2) G through D (adversarial.lua:178-201, fevalG_on_D :104-133)
            self._fill_noise(B, None if noise_g is None else np.asarray(noise_g).reshape(nG, B, -1)[k])
            self.gnet.zero_grads()
            samples = self.gnet.forward_dev(self.noise, B)
            out = self.dg.forward(samples, B)
            ctx.bce_dev(out, self.ones, B, self.loss, self.df)
            df_do = self.dg.backward(self.df, B, True)
            self.gnet.backward_dev(self.noise, df_do, B, None)
            self.t["G"] += 1
            self.gnet.adam_step(self.hyper_g, self.t["G"])
            if want_loss and k == nG - 1:
                loss_g = self._read_loss()
        return loss_d, loss_g

    def sync_to_host(self):
        self.env.MODEL_D.pull_params()
        self.env.MODEL_G.pull_params()

    def close(self):
        """Release the device buffers of the game (the nets stay with their models)."""
        self.dg.close()
        for p in (self.inputs, self.targets, self.ones, self.noise, self.df, self.loss):
            self.ctx.free(p)
        self.inputs = self.targets = self.ones = self.noise = self.df = self.loss = None
