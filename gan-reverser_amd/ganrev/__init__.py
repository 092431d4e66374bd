"""ganrev — MI355X (gfx950) implementation of gan-reverser's hot path behind the Torch7 nn.Module surface.

    from ganrev import nn, models, optim, nn_utils

All compute goes through libganrev.so (hand-written HIP, include/ganrev.h); importing this package never touches
the GPU, calling it without the library or without a gfx950 device raises GanrevError (no CPU fallback).
"""
from . import _lib  # noqa: F401
from ._lib import GanrevError, Hyper  # noqa: F401
from . import nn, models, optim, nn_utils, weight_init, synth, apply_r, parallel  # noqa: F401

__all__ = ["nn", "models", "optim", "nn_utils", "weight_init", "synth", "apply_r", "parallel", "GanrevError", "Hyper"]
