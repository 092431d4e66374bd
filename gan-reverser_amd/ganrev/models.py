"""Mirror of the reference model zoo's two live constructors (models.lua): create_G -> create_G3 (models.lua:201-203,
104-143) and create_R -> create_R_default (models.lua:385-387, 389-464).  Same layer lists, same argument meaning."""
from . import nn
from .weight_init import w_init


class _CudnnSpatialConvolution(nn.SpatialConvolution):
    TYPENAME = "cudnn.SpatialConvolution"      # not matched by weight-init.lua:54 (only the bias is zeroed)


class _CudnnReLU(nn.ReLU):
    TYPENAME = "cudnn.ReLU"


def create_G3(dimensions, noiseDim, cuda=True, seed=0):
    """models.lua:104-143.  dimensions = (channels, height, width)."""
    nn.manualSeed(seed)          # the cudnn.* convolutions and the BatchNorm gammas keep their constructor draw (weight-init.lua:54-67)
    model = nn.Sequential()
    if cuda:
        model.add(nn.Copy("torch.FloatTensor", "torch.CudaTensor", True, True))
    startHeight = dimensions[1] // 2 // 2
    startWidth = dimensions[2] // 2 // 2
    model.add(nn.Linear(noiseDim, 512 * startHeight * startWidth))
    model.add(nn.BatchNormalization(512 * startHeight * startWidth))
    model.add(_CudnnReLU(True))
    model.add(nn.View(512, startHeight, startWidth))
    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(_CudnnSpatialConvolution(512, 256, 3, 3, 1, 1, 1, 1))
    model.add(nn.SpatialBatchNormalization(256))
    model.add(_CudnnReLU(True))
    model.add(nn.SpatialUpSamplingNearest(2))
    model.add(_CudnnSpatialConvolution(256, 128, 3, 3, 1, 1, 1, 1))
    model.add(nn.SpatialBatchNormalization(128))
    model.add(_CudnnReLU(True))
    model.add(_CudnnSpatialConvolution(128, dimensions[0], 3, 3, 1, 1, 1, 1))
    model.add(nn.Sigmoid())
    if cuda:
        model.add(nn.Copy("torch.CudaTensor", "torch.FloatTensor", True, True))
        model.cuda()
    return w_init(model, "heuristic", seed)


def create_G(dimensions, noiseDim, cuda=True, seed=0):
    return create_G3(dimensions, noiseDim, cuda, seed)        # models.lua:201-203


def create_R_default(dimensions, noiseDim, noiseMethod="normal", fixer=False, cuda=True, seed=0):
    """models.lua:389-464."""
    assert noiseMethod in ("normal", "uniform")                # models.lua:390
    nn.manualSeed(seed)
    conv = nn.Sequential()
    if cuda:
        conv.add(nn.Copy("torch.FloatTensor", "torch.CudaTensor", True, True))
    if fixer:
        conv.add(nn.Dropout(0.5, True).keepAlwaysOn())         # models.lua:399-406
    c = dimensions[0]
    for block, (cin, cout) in enumerate([(c, 64), (64, 64), (64, 64), (64, 128), (128, 128), (128, 128)]):
        conv.add(nn.SpatialConvolution(cin, cout, 3, 3, 1, 1, 1, 1))
        conv.add(nn.SpatialBatchNormalization(cout))
        conv.add(nn.ELU())
        if block == 2:                                          # models.lua:419-423
            conv.add(nn.SpatialMaxPooling(2, 2))
            conv.add(nn.Dropout())
        elif block == 5:                                        # models.lua:436-440
            conv.add(nn.SpatialDropout(0.25))
            conv.add(nn.SpatialMaxPooling(2, 2))
        else:
            conv.add(nn.Dropout())
    height = dimensions[1] // 2 // 2
    width = dimensions[2] // 2 // 2
    conv.add(nn.View(128 * height * width))
    conv.add(nn.Linear(128 * height * width, 512))
    conv.add(nn.BatchNormalization(512))
    conv.add(nn.ELU())
    conv.add(nn.Dropout(0.5))
    conv.add(nn.Linear(512, noiseDim))
    if noiseMethod != "normal":
        conv.add(nn.Tanh())
    if cuda:
        conv.add(nn.Copy("torch.CudaTensor", "torch.FloatTensor", True, True))
        conv.cuda()
    return w_init(conv, "heuristic", seed)


def create_R(dimensions, noiseDim, noiseMethod="normal", fixer=False, cuda=True, seed=0):
    return create_R_default(dimensions, noiseDim, noiseMethod, fixer, cuda, seed)   # models.lua:385-387
