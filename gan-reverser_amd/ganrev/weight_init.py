"""Mirror of the reference's weight-init.lua (heuristic = sqrt(1/(3*fan_in)), top-level modules only, every bias zeroed)."""
import math

import numpy as np


def w_init_heuristic(fan_in, fan_out):
    return math.sqrt(1.0 / (3 * fan_in))          # weight-init.lua:14-16


def w_init_xavier(fan_in, fan_out):
    return math.sqrt(2.0 / (fan_in + fan_out))     # weight-init.lua:21-23


def w_init_xavier_caffe(fan_in, fan_out):
    return math.sqrt(1.0 / fan_in)                 # weight-init.lua:28-30


def w_init_kaiming(fan_in, fan_out):
    return math.sqrt(4.0 / (fan_in + fan_out))     # weight-init.lua:35-37


_METHODS = {"heuristic": w_init_heuristic, "xavier": w_init_xavier, "xavier_caffe": w_init_xavier_caffe, "kaiming": w_init_kaiming}


def w_init(net, arg, seed=0):
    """weight-init.lua:40-75.  Only typenames 'nn.SpatialConvolution' and 'nn.Linear' are re-initialised — the
    reference's G is built from cudnn.SpatialConvolution, which the typename test does not match (weight-init.lua:54-67),
    so G's convolutions keep their constructor init and only get their bias zeroed (weight-init.lua:70-72)."""
    from . import nn
    method = _METHODS[arg]
    rng = nn._rng()              # the same process-wide stream the constructors drew from (seeded by models.create_*)
    for m in getattr(net, "modules", []):
        tn = m.typename
        if tn == "nn.SpatialConvolution":
            m.reset(method(m.nInputPlane * m.kH * m.kW, m.nOutputPlane * m.kH * m.kW), rng)
        elif tn == "nn.Linear":
            m.reset(method(m.weight.shape[1], m.weight.shape[0]), rng)
        if getattr(m, "bias", None) is not None:
            m.bias[...] = 0
    return net
