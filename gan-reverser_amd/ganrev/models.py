"""Mirror of the reference model zoo's live constructors (models.lua): create_G -> create_G3 (models.lua:201-203,
104-143), create_R -> create_R_default (models.lua:385-387, 389-464) and create_D -> create_D2 (models.lua:209-211, 272-337;
the discriminator adversarial.lua trains G against - SURVEY.md 8f rank 4).  Same layer lists, same argument meaning."""
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


def create_D2(dimensions, cuda=True, seed=0):
    """models.lua:272-337: two 3x3 convolutions, then nn.Concat(2) of a 5x5 tower and a deeper 3x3 tower, joined by two Linear
    layers into one sigmoid unit.  Every activation is an nn.PReLU() with its own learnable slope."""
    nn.manualSeed(seed)

    def createNxN(nbKernelsIn, nbKernelsOut, kernelSize, dropout):      # models.lua:273-281
        model = nn.Sequential()
        pad = (kernelSize - 1) // 2
        model.add(nn.SpatialConvolution(nbKernelsIn, nbKernelsOut, kernelSize, kernelSize, 1, 1, pad, pad))
        model.add(nn.PReLU())
        if dropout > 0:
            model.add(nn.SpatialDropout(0.25))
        return model

    conv = nn.Sequential()
    if cuda:
        conv.add(nn.Copy("torch.FloatTensor", "torch.CudaTensor", True, True))
    conv.add(createNxN(dimensions[0], 128, 3, 0))
    conv.add(createNxN(128, 128, 3, 0.2))
    conv.add(nn.SpatialMaxPooling(2, 2))
    concat = nn.Concat(2)
    left, right = nn.Sequential(), nn.Sequential()
    h4, w4 = dimensions[1] // 2 // 2, dimensions[2] // 2 // 2
    left.add(createNxN(128, 64, 5, 0.2))
    left.add(nn.SpatialMaxPooling(2, 2))
    left.add(nn.View(64 * h4 * w4))
    left.add(nn.Linear(64 * h4 * w4, 512))
    left.add(nn.PReLU())
    left.add(nn.Dropout(0.25))
    right.add(createNxN(128, 128, 3, 0.2))
    right.add(nn.SpatialMaxPooling(2, 2))
    right.add(createNxN(128, 256, 3, 0.2))
    right.add(createNxN(256, 256, 3, 0.2))
    right.add(nn.SpatialMaxPooling(2, 2))
    height, width = dimensions[1] // 2 // 2 // 2, dimensions[2] // 2 // 2 // 2
    right.add(nn.View(256 * height * width))
    right.add(nn.Linear(256 * height * width, 512))
    right.add(nn.PReLU())
    concat.add(left)
    concat.add(right)
    conv.add(concat)
    conv.add(nn.Linear(512 + 512, 256))
    conv.add(nn.PReLU())
    conv.add(nn.Dropout(0.25))
    conv.add(nn.Linear(256, 1))
    conv.add(nn.Sigmoid())
    if cuda:
        conv.add(nn.Copy("torch.CudaTensor", "torch.FloatTensor", True, True))
        conv.cuda()
    return w_init(conv, "heuristic", seed)


def create_D(dimensions, cuda=True, seed=0):
    return create_D2(dimensions, cuda, seed)                           # models.lua:209-211
