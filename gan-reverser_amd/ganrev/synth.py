"""Deterministic synthetic inputs (counter-based, numpy only) shared by tests, fixtures and bench: the reference's
createNoiseInputs draws from Torch's MT19937 (utils/nn_utils.lua:39-51), which is an input generator, not part of the
path; benches and parity tests use these seeded streams instead."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(shape, seed):
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        r = _splitmix64(_splitmix64(idx))
    u = ((r >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)
    return u.reshape(shape)


def uniform(shape, seed, lo=-1.0, hi=1.0):
    return (lo + (hi - lo) * uniform01(shape, seed)).astype(np.float32)


def normal(shape, seed):
    u1 = uniform01(shape, seed * 2 + 1)
    u2 = uniform01(shape, seed * 2 + 2)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def bernoulli_keep(shape, seed, p_drop):
    return (uniform01(shape, seed) >= p_drop).astype(np.uint8)


def init_params(model, seed):
    """Realistic weights for parity tests: conv/linear U(-sqrt(1/fan_in), +) (= weight-init heuristic stdv*sqrt(3)),
    biases small non-zero (so bias paths are exercised), BN gamma U(0.5,1.5), beta small, running stats non-trivial."""
    from . import nn
    k = seed * 1000
    for m in model.leaves():
        k += 1
        if isinstance(m, nn.BatchNormalization):
            m.weight[...] = uniform(m.weight.shape, k, 0.5, 1.5)
            m.bias[...] = uniform(m.bias.shape, k + 500, -0.2, 0.2)
            m.running_mean[...] = uniform(m.running_mean.shape, k + 600, -0.3, 0.3)
            m.running_var[...] = uniform(m.running_var.shape, k + 700, 0.5, 1.5)
        elif isinstance(m, (nn.SpatialConvolution, nn.Linear)):
            fan_in = int(np.prod(m.weight.shape[1:])) if not isinstance(m, nn.SpatialFullConvolution) else m.weight.shape[0] * 9
            s = 1.0 / np.sqrt(fan_in)
            m.weight[...] = uniform(m.weight.shape, k, -s, s)
            m.bias[...] = uniform(m.bias.shape, k + 500, -0.05, 0.05)
    return model
