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
