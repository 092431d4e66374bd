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
    if model is None or model._net is None:
        raise L.GanrevError("optim.adam needs the model whose getParameters() produced x (model=...)")
    net = model._net
    state["t"] = state.get("t", 0) + 1
    if "m" not in state:
        state["m"] = np.zeros_like(x)
        state["v"] = np.zeros_like(x)
    h = L.Hyper(lr=config.get("learningRate", 1e-3), beta1=config.get("beta1", 0.9), beta2=config.get("beta2", 0.999),
                eps=config.get("epsilon", 1e-8), l1=0.0, l2=0.0, clamp=0.0)   # penalties are fevalR's job here
    net.set_params(x)
    net.set_grads(dfdx)
    net.set_adam_state(state["m"], state["v"])
    net.adam_step(h, state["t"])
    x[...] = net.get_params()
    state["m"], state["v"] = net.adam_state()
    return x, [fx]
