"""Mirror of the two hot-path helpers of utils/nn_utils.lua."""
import numpy as np

from . import _lib as L
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


class DeviceTensor:
    """A row-major fp32 tensor in GPU memory (what a torch.CudaTensor is to the reference's scripts): pointer + shape, nothing else.
    .numpy() copies it to the host; .free() releases it."""

    def __init__(self, ctx, shape, ptr=None):
        self.ctx, self.shape = ctx, tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape))
        self.owned = ptr is None
        self.ptr = ctx.malloc(4 * max(self.size, 1)) if ptr is None else ptr

    def rows(self, lo, hi=None):
        """view of rows [lo, hi)"""
        hi = self.shape[0] if hi is None else hi
        per = self.size // max(self.shape[0], 1)
        return DeviceTensor(self.ctx, (hi - lo,) + self.shape[1:], self.ptr + 4 * per * lo)

    def numpy(self):
        return self.ctx.download(self.ptr, self.shape)

    def free(self):
        if self.owned and self.ptr is not None:
            self.ctx.free(self.ptr)
        self.ptr = None


def forwardBatchedDev(model, input, batchSize, out=None):
    """utils/nn_utils.lua:5-33 with input and result resident on the GPU: `input` is a DeviceTensor [N x ...], the result a
    DeviceTensor [N x model output]; each chunk's last kernel writes its rows of the result itself (gr_net_forward_batched_dev) -
    the reference's per-row copy loop (:25-28) has no counterpart and nothing visits the host."""
    N = input.shape[0]
    net = model.device_net(input.shape[1:])
    shape = (N,) + L.Net._shape(net.out_dims)
    if out is None:
        out = DeviceTensor(input.ctx, shape)
    assert out.shape == shape, (out.shape, shape)
    net.forward_batched_dev(input.ptr, N, batchSize, out.ptr)
    return out


def createNoiseInputsDev(ctx, N, noiseDim, method="normal", seed=1):
    """utils/nn_utils.lua:39-51 drawn on the GPU (Philox4x32-10; the stream tests/test_gpu_abi_behaviour.py pins)."""
    t = DeviceTensor(ctx, (N, noiseDim))
    if method == "uniform":
        ctx.fill_uniform(t.ptr, N * noiseDim, seed)
    elif method == "normal":
        ctx.fill_normal(t.ptr, N * noiseDim, seed)
    else:
        raise ValueError(f"Unknown noise method '{method}'")   # utils/nn_utils.lua:48
    return t


def createNoiseInputs(N, noiseDim, method="normal", seed=1):
    """utils/nn_utils.lua:39-51 — N x noiseDim, normal(0,1) or uniform(-1,1)."""
    if method == "uniform":
        return synth.uniform((N, noiseDim), seed)
    if method == "normal":
        return synth.normal((N, noiseDim), seed)
    raise ValueError(f"Unknown noise method '{method}'")   # utils/nn_utils.lua:48
