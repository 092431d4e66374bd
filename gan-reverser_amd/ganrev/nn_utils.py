"""Mirror of the two hot-path helpers of utils/nn_utils.lua."""
import numpy as np

from . import synth


def forwardBatched(model, input, batchSize):
    """utils/nn_utils.lua:5-33 — chunked forward; the reference copies row by row in Lua (:25-28), here each chunk's
    result lands in the output with one strided copy."""
    N = len(input)
    output = None
    nBatches = -(-N // batchSize)
    for i in range(nBatches):
        s, e = i * batchSize, min((i + 1) * batchSize, N)
        forwarded = model.forward(input[s:e])
        if output is None:
            output = np.empty((N,) + forwarded.shape[1:], dtype=np.float32)
        output[s:e] = forwarded
    return output


def createNoiseInputs(N, noiseDim, method="normal", seed=1):
    """utils/nn_utils.lua:39-51 — N x noiseDim, normal(0,1) or uniform(-1,1)."""
    if method == "uniform":
        return synth.uniform((N, noiseDim), seed)
    if method == "normal":
        return synth.normal((N, noiseDim), seed)
    raise ValueError(f"Unknown noise method '{method}'")   # utils/nn_utils.lua:48
