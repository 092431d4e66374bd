"""optim.adam as the reference calls it (train_r.lua:125,170): adam(opfunc, x, config) with an initially empty state.

Two forms:
  adam(opfunc, x, state)        — drop-in: opfunc(x) -> f, dfdx on host arrays; the update runs on the GPU through
                                  gr_adam_step when x is a model's flat parameter storage.
  fused step inside gr_train_r_step — the production path (ganrev.train_r.Trainer).
"""
import ctypes as C

import numpy as np

from . import _lib as L


def adam(opfunc, x, config=None, state=None, model=None):
    config = config if config is not None else {}
    state = state if state is not None else config
    fx, dfdx = opfunc(x)
    chunks = model._param_chunks() if model is not None else []
    if not chunks or any(ch._net is None for ch, _, _ in chunks):
        raise L.GanrevError("optim.adam needs the model whose getParameters() produced x (model=...), after a forward")
    state["t"] = state.get("t", 0) + 1
    if "m" not in state:
        state["m"] = np.zeros_like(x)
        state["v"] = np.zeros_like(x)
    h = L.Hyper(lr=config.get("learningRate", 1e-3), beta1=config.get("beta1", 0.9), beta2=config.get("beta2", 0.999),
                eps=config.get("epsilon", 1e-8), l1=0.0, l2=0.0, clamp=0.0)   # penalties are fevalR's job here
    # the update is element-wise: a model that runs as several gr_nets (one with an nn.Concat, the D network) is stepped
    # slice by slice, each net on the part of the flat vector it owns
    for ch, lo, hi in chunks:
        if hi == lo:
            continue
        net = ch._net
        net.set_params(x[lo:hi])
        net.set_grads(dfdx[lo:hi])
        net.set_adam_state(state["m"][lo:hi], state["v"][lo:hi])
        net.adam_step(h, state["t"])
        x[lo:hi] = net.get_params()
        state["m"][lo:hi], state["v"][lo:hi] = net.adam_state()
    return x, [fx]


# ---------------------------------------------------------------------------------------------------------------------
# The other five methods adversarial.lua:156-171 / 183-198 can select (train.lua:37-38: sgd|adagrad|adadelta|adamax|adam|rmsprop).
# They belong to the un-vendored, un-pinned `optim` luarock, like adam: restated here FROM MEMORY of that package (each function
# says which defaults and which update it takes), on the host flat vectors exactly as the reference runs them - `optim.*` there
# are Torch tensor expressions on PARAMETERS_D / PARAMETERS_G, element-wise, with the state kept in the table the caller passes
# (train.lua:183-193: empty tables, and {learningRate = OPT.X_sgd_lr, momentum = OPT.X_sgd_momentum} for sgd).  adam is the
# default and the only one with a fused device kernel; these run in float32 numpy (TH float tensors) and serve the --compat game.
def _state(config, state):
    config = config if config is not None else {}
    return config, (state if state is not None else config)


def sgd(opfunc, x, config=None, state=None):
    """optim.sgd [upstream, from memory]: weightDecay 0, momentum 0, dampening = momentum, nesterov false, learningRateDecay 0;
    clr = lr / (1 + nevals * lrd); with momentum: dfdx_buf = mom * buf + (1 - damp) * dfdx (first call: buf = dfdx); x -= clr * dfdx."""
    config, state = _state(config, state)
    lr, lrd = config.get("learningRate", 1e-3), config.get("learningRateDecay", 0.0)
    wd, mom = config.get("weightDecay", 0.0), config.get("momentum", 0.0)
    damp, nesterov = config.get("dampening", mom), config.get("nesterov", False)
    if nesterov and not (mom > 0 and damp == 0):
        raise ValueError("Nesterov momentum requires a momentum and zero dampening")
    state["evalCounter"] = state.get("evalCounter", 0)
    fx, dfdx = opfunc(x)
    dfdx = np.array(dfdx, np.float32, copy=True)
    if wd != 0:
        dfdx += np.float32(wd) * x
    if mom != 0:
        if "dfdx" not in state:
            state["dfdx"] = dfdx.copy()
        else:
            state["dfdx"] = np.float32(mom) * state["dfdx"] + np.float32(1 - damp) * dfdx
        dfdx = dfdx + np.float32(mom) * state["dfdx"] if nesterov else state["dfdx"]
    clr = lr / (1 + state["evalCounter"] * lrd)
    x -= np.float32(clr) * dfdx
    state["evalCounter"] += 1
    return x, [fx]


def adagrad(opfunc, x, config=None, state=None):
    """optim.adagrad [upstream, from memory]: learningRate 1e-3, learningRateDecay 0; paramVariance += dfdx^2;
    x -= clr * dfdx / (sqrt(paramVariance) + 1e-10)."""
    config, state = _state(config, state)
    lr, lrd = config.get("learningRate", 1e-3), config.get("learningRateDecay", 0.0)
    state["evalCounter"] = state.get("evalCounter", 0)
    fx, dfdx = opfunc(x)
    dfdx = np.asarray(dfdx, np.float32)
    clr = lr / (1 + state["evalCounter"] * lrd)
    if "paramVariance" not in state:
        state["paramVariance"] = np.zeros_like(x)
    state["paramVariance"] += dfdx * dfdx
    x -= np.float32(clr) * dfdx / (np.sqrt(state["paramVariance"]) + np.float32(1e-10))
    state["evalCounter"] += 1
    return x, [fx]


def adadelta(opfunc, x, config=None, state=None):
    """optim.adadelta [upstream, from memory]: rho 0.9, eps 1e-6; paramVariance = rho * pv + (1 - rho) * g^2; std = sqrt(pv + eps);
    delta = sqrt(accDelta + eps) / std * g; x -= delta; accDelta = rho * accDelta + (1 - rho) * delta^2."""
    config, state = _state(config, state)
    rho, eps = np.float32(config.get("rho", 0.9)), np.float32(config.get("eps", 1e-6))
    state["evalCounter"] = state.get("evalCounter", 0)
    fx, dfdx = opfunc(x)
    g = np.asarray(dfdx, np.float32)
    if "paramVariance" not in state:
        state["paramVariance"] = np.zeros_like(x)
        state["accDelta"] = np.zeros_like(x)
    state["paramVariance"] = rho * state["paramVariance"] + (np.float32(1) - rho) * g * g
    std = np.sqrt(state["paramVariance"] + eps)
    delta = np.sqrt(state["accDelta"] + eps) / std * g
    x -= delta
    state["accDelta"] = rho * state["accDelta"] + (np.float32(1) - rho) * delta * delta
    state["evalCounter"] += 1
    return x, [fx]


def adamax(opfunc, x, config=None, state=None):
    """optim.adamax [upstream, from memory]: learningRate 2e-3, beta1 0.9, beta2 0.999, epsilon 1e-38; m = b1 m + (1 - b1) g;
    u = max(b2 u, |g| + eps); x -= lr / (1 - b1^t) * m / u."""
    config, state = _state(config, state)
    lr, b1, b2 = config.get("learningRate", 2e-3), config.get("beta1", 0.9), config.get("beta2", 0.999)
    eps = np.float32(config.get("epsilon", 1e-38))
    fx, dfdx = opfunc(x)
    g = np.asarray(dfdx, np.float32)
    state["t"] = state.get("t", 0) + 1
    if "m" not in state:
        state["m"] = np.zeros_like(x)
        state["u"] = np.zeros_like(x)
    state["m"] = np.float32(b1) * state["m"] + np.float32(1 - b1) * g
    state["u"] = np.maximum(np.float32(b2) * state["u"], np.abs(g) + eps)
    step = lr / (1 - b1 ** state["t"])
    x -= np.float32(step) * state["m"] / state["u"]
    return x, [fx]


def rmsprop(opfunc, x, config=None, state=None):
    """optim.rmsprop [upstream, from memory]: learningRate 1e-2, alpha 0.99, epsilon 1e-8; m = alpha m + (1 - alpha) g^2 (m starts
    at zero); x -= lr * g / (sqrt(m) + eps)."""
    config, state = _state(config, state)
    lr, alpha, eps = config.get("learningRate", 1e-2), np.float32(config.get("alpha", 0.99)), np.float32(config.get("epsilon", 1e-8))
    fx, dfdx = opfunc(x)
    g = np.asarray(dfdx, np.float32)
    if "m" not in state:
        state["m"] = np.zeros_like(x)
    state["m"] = alpha * state["m"] + (np.float32(1) - alpha) * g * g
    x -= np.float32(lr) * g / (np.sqrt(state["m"]) + eps)
    return x, [fx]


METHODS = {"sgd": sgd, "adagrad": adagrad, "adadelta": adadelta, "adamax": adamax, "rmsprop": rmsprop}      # + adam (device kernel)
