"""Host-side mirror of the Torch7 `nn` surface the reference scripts use, backed by libganrev.so.

Same names, argument meaning and error behaviour as the modules the reference instantiates
(reference models.lua:104-143, 389-464) and the protocol it calls on them:
  m:forward(input) -> m.output           train_r.lua:139,146 ; utils/nn_utils.lua:18
  m:backward(input, gradOutput)          train_r.lua:151
  m:training() / m:evaluate()            train_r.lua:70,189,222 ; apply_r.lua:64,94,103
  m:getParameters() -> flat, flatGrads   train_r.lua:122
  m:listModules(), m.modules[i]          utils/nn_utils.lua:395-426 ; weight-init.lua:52-72
Tensors are host numpy float32 arrays, contiguous NCHW — the reference's FloatTensors (train_r.lua:63);
the nn.Copy modules that bracket every reference model (models.lua:108,136,394,457) are accepted and ignored:
the host<->device crossing happens inside forward()/backward() exactly where those modules sat.

A Sequential is compiled into ONE gr_net on first use; leaf modules called on their own are one-layer nets.
"""
import math

import numpy as np

from . import _lib as L

# Torch7 draws every module's initial parameters from ONE process-wide generator (torch.manualSeed, train_r.lua:38-39), so two
# modules of equal shape never start out identical.  Constructors and reset() without an explicit rng draw from this one.
_RNG = np.random.default_rng(0)


def manualSeed(seed):
    """torch.manualSeed for parameter initialisation: restart the generator the constructors draw from."""
    global _RNG
    _RNG = np.random.default_rng(int(seed))


def _rng(rng=None):
    return rng if rng is not None else _RNG


class Module:
    __typename = "nn.Module"

    def __init__(self):
        self.train = True
        self.output = None
        self.gradInput = None
        self._net = None
        self._net_key = None
        self._flat = None          # (flatParams, flatGrads) after getParameters()
        self._pending_masks = {}
        self._ctx = None

    # ---- tree protocol
    @property
    def typename(self):
        return self.__class__.TYPENAME

    def listModules(self):
        return [self]

    def leaves(self):
        return [self]

    def desc(self, dims):
        """-> (list of (kind,a,b,c,p,flags), new dims).  dims = (C,H,W) per sample."""
        raise NotImplementedError

    def param_arrays(self):
        return []

    def set_param_arrays(self, arrays):
        pass

    def training(self):
        for m in self.listModules():
            m.train = True
        return self

    def evaluate(self):
        for m in self.listModules():
            if not getattr(m, "always_on", False):     # models.lua:404  drop.evaluate = function() end
                m.train = False
        return self

    def float(self):   # train_r.lua:78,104 — tensors are already float
        return self

    def cuda(self):    # models.lua:137,458 — device placement happens at compile time
        return self

    def zeroGradParameters(self):
        if self._flat is not None:
            self._flat[1][...] = 0
        if self._net is not None:
            self._net.zero_grads()

    def __repr__(self):
        return self.typename

    # ---- compilation
    def _context(self):
        if self._ctx is None:
            self._ctx = L.default_context()
        return self._ctx

    def _in_dims(self, x):
        if x.ndim == 4:
            return tuple(x.shape[1:])
        if x.ndim == 2:
            return (x.shape[1], 1, 1)
        raise ValueError(f"expected a batched 2-D or 4-D tensor, got shape {x.shape} "
                         "(BatchNormalization forces a batch dimension: apply_r.lua:330-331)")

    def _descs(self, dims):
        descs, index = [], {}
        d = dims
        for m in self.leaves():
            ds, d = m.desc(d)
            index[id(m)] = len(descs)
            descs.extend(ds)
        return descs, index

    def _compile(self, x):
        dims = self._in_dims(x)
        if self._net is not None and self._net_key == dims:
            return self._net
        if self._net is not None:
            self.pull_params()
            self._net.close()
        descs, self._layer_index = self._descs(dims)
        if not descs:
            raise L.GanrevError("empty module")
        self._net = L.Net(self._context(), descs, dims)
        self._net_key = dims
        self.push_params()
        return self._net

    def _flat_host(self):
        arrs = [a for m in self.leaves() for a in m.param_arrays()]
        if not arrs:
            return np.zeros(0, np.float32)
        return np.concatenate([a.ravel() for a in arrs]).astype(np.float32)

    def push_params(self):
        """host parameter arrays (+ BN running stats) -> device."""
        if self._net is None:
            return
        flat = self._flat[0] if self._flat is not None else self._flat_host()
        if flat.size:
            self._net.set_params(flat)
        bi = 0
        for m in self.leaves():
            if isinstance(m, BatchNormalization):
                self._net.set_bn_running(bi, m.running_mean, m.running_var)
                bi += 1

    def pull_params(self):
        """device -> host parameter arrays (+ BN running stats)."""
        if self._net is None:
            return
        flat = self._net.get_params()
        if self._flat is not None:
            self._flat[0][...] = flat
        else:
            off = 0
            for m in self.leaves():
                new = []
                for a in m.param_arrays():
                    new.append(flat[off:off + a.size].reshape(a.shape).copy())
                    off += a.size
                m.set_param_arrays(new)
        bi = 0
        for m in self.leaves():
            if isinstance(m, BatchNormalization):
                m.running_mean, m.running_var = self._net.get_bn_running(bi)
                bi += 1

    def getParameters(self):
        """Flattens every parameter into one storage; module weight/bias become views into it (train_r.lua:122)."""
        if self._flat is None:
            n = self._param_count()
            self._bind_flat(np.zeros(n, np.float32), np.zeros(n, np.float32))
        return self._flat

    def _param_count(self):
        return int(sum(a.size for m in self.leaves() for a in m.param_arrays()))

    def _bind_flat(self, flat, grads):
        """Make `flat` / `grads` (given storage: a container hands each compiled part its slice) this module's flat vectors."""
        if self._net is not None:
            self.pull_params()
        flat[...] = self._flat_host()
        off = 0
        for m in self.leaves():
            views, gviews = [], []
            for a in m.param_arrays():
                views.append(flat[off:off + a.size].reshape(a.shape))
                gviews.append(grads[off:off + a.size].reshape(a.shape))
                off += a.size
            m.set_param_arrays(views)
            m.set_grad_arrays(gviews)
        self._flat = (flat, grads)

    def _param_chunks(self, lo=0):
        """[(module compiled into one gr_net, lo, hi)]: the slices of the flat vector each device net owns."""
        return [(self, lo, lo + self._param_count())]

    def set_grad_arrays(self, arrays):
        pass

    # ---- dropout noise injection (tests) / read-back
    def _leaf_layer(self, module):
        return self._layer_index[id(module)] + module.noise_layer_offset()

    def noise_layer_offset(self):
        return 0

    def setNoise(self, module, keep):
        """Use `keep` (0/1 per element of module's noise tensor) for the NEXT forward instead of Philox noise."""
        self._pending_masks[id(module)] = (module, np.ascontiguousarray(keep, dtype=np.uint8))

    def getNoise(self, module, batch):
        n = self._net.mask_size(self._leaf_layer(module), batch)
        return self._net.get_mask(self._leaf_layer(module), n)

    def manualSeed(self, seed):
        self._seed = int(seed)
        if self._net is not None:
            self._net.set_seed(self._seed)

    # ---- the nn.Module protocol
    def _prepare(self, x):
        """Everything :forward does before the data moves: compile for x's dims, seed, parameters, mode, injected noise."""
        net = self._compile(x)
        if getattr(self, "_seed", None) is not None and not getattr(self, "_seed_applied", False):
            net.set_seed(self._seed)
            self._seed_applied = True
        if self._flat is not None:
            net.set_params(self._flat[0])      # host storage is authoritative after getParameters()
        mods = [m for m in self.leaves() if not getattr(m, "always_on", False)] or self.leaves()
        net.set_training(any(m.train for m in mods))     # the fixer's always-on Dropout does not make the net "training"
        self._sync_modes(net)
        for module, keep in self._pending_masks.values():
            net.set_mask(self._leaf_layer(module), keep)
        self._pending_masks = {}
        return net

    def device_net(self, dims):
        """The compiled gr_net for per-sample input dims (C, H, W) / (n,), ready for the *_dev calls (no data moved): what the
        device-resident loops (apply_r.embed_dev, DeviceTrainer) drive instead of :forward."""
        dims = tuple(int(d) for d in (dims if len(dims) == 3 else (dims[0], 1, 1)))
        probe = np.empty((0,) + (dims if dims[1:] != (1, 1) else dims[:1]), np.float32)
        return self._prepare(probe)

    def forward(self, input):
        x = L.f32(input)
        net = self._prepare(x)
        b = x.shape[0]
        shape = (b,) + L.Net._shape(net.out_dims)
        if self.output is None or self.output.shape != shape:
            self.output = np.empty(shape, dtype=np.float32)
        net.forward(x, self.output)
        return self.output

    def _sync_modes(self, net):
        modes = {m.train for m in self.leaves() if not getattr(m, "always_on", False)}
        if len(modes) > 1:
            raise L.GanrevError("mixed training()/evaluate() modes inside one Sequential are not supported")

    updateOutput = forward

    def backward(self, input, gradOutput, scale=1):
        if scale != 1:
            raise L.GanrevError("only scale=1 is supported (nn.Sequential:backward default)")
        if self._net is None:
            raise L.GanrevError("backward called before forward")
        x, g = L.f32(input), L.f32(gradOutput)
        net = self._net
        if self._flat is not None:
            net.zero_grads()
        self.gradInput = net.backward(x, g, want_gin=True)
        if self._flat is not None:
            self._flat[1][...] += net.get_grads()  # accGradParameters accumulates into the flat gradient
        return self.gradInput


class Sequential(Module):
    TYPENAME = "nn.Sequential"

    def __init__(self):
        super().__init__()
        self.modules = []

    def add(self, m):
        self.modules.append(m)
        self._parts = None             # the execution plan of a model with an nn.Concat is rebuilt on the next use
        return self

    def get(self, i):
        return self.modules[i - 1]     # Lua is 1-based

    def size(self):
        return len(self.modules)

    def listModules(self):
        out = [self]
        for m in self.modules:
            out.extend(m.listModules())
        return out

    def leaves(self):
        out = []
        for m in self.modules:
            out.extend(m.leaves())
        return out

    # ---- a Sequential that holds an nn.Concat (models.lua:293-321, the D network) cannot be one gr_net: it runs as a chain
    # of PARTS - every run of plain modules is compiled into one gr_net (a chunk), a branching container runs its branches.
    # Host arrays travel between the parts, as Torch7 tensors travel between the modules of the reference's containers.
    def _is_graph(self):
        return any(isinstance(m, Concat) or (isinstance(m, Sequential) and m._is_graph()) for m in self.modules)

    def parts(self):
        if getattr(self, "_parts", None) is None:
            parts, run = [], None
            for m in self.modules:
                if isinstance(m, Concat) or (isinstance(m, Sequential) and m._is_graph()):
                    if run is not None:
                        parts.append(run)
                        run = None
                    parts.append(m)
                elif m.leaves():
                    if run is None:
                        run = Sequential()
                    run.add(m)
            if run is not None:
                parts.append(run)
            self._parts = parts
        return self._parts

    def forward(self, input):
        if not self._is_graph():
            return Module.forward(self, input)
        x = L.f32(input)
        self._part_inputs = []
        for p in self.parts():
            self._part_inputs.append(x)
            x = p.forward(x)
        self.output = x
        return x

    updateOutput = forward

    def backward(self, input, gradOutput, scale=1):
        if not self._is_graph():
            return Module.backward(self, input, gradOutput, scale)
        if getattr(self, "_part_inputs", None) is None:
            raise L.GanrevError("backward called before forward")
        g = L.f32(gradOutput)
        for p, x in zip(reversed(self.parts()), reversed(self._part_inputs)):
            g = p.backward(x, g, scale)
        self.gradInput = g
        return g

    def _bind_flat(self, flat, grads):
        if not self._is_graph():
            return Module._bind_flat(self, flat, grads)
        off = 0
        for p in self.parts():
            k = p._param_count()
            p._bind_flat(flat[off:off + k], grads[off:off + k])
            off += k
        self._flat = (flat, grads)

    def _param_chunks(self, lo=0):
        if not self._is_graph():
            return Module._param_chunks(self, lo)
        out = []
        for p in self.parts():
            out.extend(p._param_chunks(lo))
            lo += p._param_count()
        return out

    def push_params(self):
        if not self._is_graph():
            return Module.push_params(self)
        for p in self.parts():
            p.push_params()

    def pull_params(self):
        if not self._is_graph():
            return Module.pull_params(self)
        for p in self.parts():
            p.pull_params()

    def zeroGradParameters(self):
        if not self._is_graph():
            return Module.zeroGradParameters(self)
        if self._flat is not None:
            self._flat[1][...] = 0
        for p in self.parts():
            p.zeroGradParameters()

    def manualSeed(self, seed):
        if not self._is_graph():
            return Module.manualSeed(self, seed)
        for i, (chunk, _, _) in enumerate(self._param_chunks()):
            chunk.manualSeed(int(seed) * 1009 + i)

    def _owner(self, module):
        """The compiled chunk a leaf module sits in (dropout-noise injection / read-back of a graph model)."""
        for p in self.parts():
            if isinstance(p, Concat) or p._is_graph():
                o = p._owner(module)
                if o is not None:
                    return o
            elif any(m is module for m in p.leaves()):
                return p
        return None

    def setNoise(self, module, keep):
        if not self._is_graph():
            return Module.setNoise(self, module, keep)
        self._owner(module).setNoise(module, keep)

    def getNoise(self, module, batch):
        if not self._is_graph():
            return Module.getNoise(self, module, batch)
        return self._owner(module).getNoise(module, batch)

    def __repr__(self):
        lines = ["nn.Sequential {"]
        lines += [f"  ({i + 1}): {m!r}" for i, m in enumerate(self.modules)]
        return "\n".join(lines + ["}"])


class Concat(Module):
    """nn.Concat(dimension): every branch gets the same input, the outputs are joined along `dimension` (1-based, the batch
    is dimension 1): the D network's two convolution towers (models.lua:293-321, `nn.Concat(2)` of two [B x 512] feature
    vectors).  backward hands each branch its slice of gradOutput and sums the branches' gradInputs."""
    TYPENAME = "nn.Concat"

    def __init__(self, dimension):
        super().__init__()
        self.dimension = int(dimension)
        self.modules = []

    def add(self, m):
        if not isinstance(m, Sequential):
            m = Sequential().add(m)
        self.modules.append(m)
        return self

    def get(self, i):
        return self.modules[i - 1]

    def size(self):
        return len(self.modules)

    def listModules(self):
        out = [self]
        for m in self.modules:
            out.extend(m.listModules())
        return out

    def leaves(self):
        out = []
        for m in self.modules:
            out.extend(m.leaves())
        return out

    def _is_graph(self):
        return True

    def forward(self, input):
        x = L.f32(input)
        outs = [b.forward(x) for b in self.modules]
        ax = self.dimension - 1
        if any(o.ndim <= ax or o.shape[:ax] != outs[0].shape[:ax] or o.shape[ax + 1:] != outs[0].shape[ax + 1:] for o in outs):
            raise L.GanrevError(f"nn.Concat({self.dimension}): branch outputs {[o.shape for o in outs]} do not line up")
        self._sizes = [o.shape[ax] for o in outs]
        self.output = np.concatenate(outs, axis=ax)
        return self.output

    updateOutput = forward

    def backward(self, input, gradOutput, scale=1):
        if getattr(self, "_sizes", None) is None:
            raise L.GanrevError("backward called before forward")
        x, g = L.f32(input), L.f32(gradOutput)
        ax, lo, gin = self.dimension - 1, 0, None
        for b, k in zip(self.modules, self._sizes):
            sl = [slice(None)] * g.ndim
            sl[ax] = slice(lo, lo + k)
            gi = b.backward(x, np.ascontiguousarray(g[tuple(sl)]), scale)
            gin = gi.copy() if gin is None else gin + gi
            lo += k
        self.gradInput = gin
        return gin

    def _bind_flat(self, flat, grads):
        off = 0
        for b in self.modules:
            k = b._param_count()
            b._bind_flat(flat[off:off + k], grads[off:off + k])
            off += k
        self._flat = (flat, grads)

    def _param_chunks(self, lo=0):
        out = []
        for b in self.modules:
            out.extend(b._param_chunks(lo))
            lo += b._param_count()
        return out

    def push_params(self):
        for b in self.modules:
            b.push_params()

    def pull_params(self):
        for b in self.modules:
            b.pull_params()

    def zeroGradParameters(self):
        if self._flat is not None:
            self._flat[1][...] = 0
        for b in self.modules:
            b.zeroGradParameters()

    def _owner(self, module):
        for b in self.modules:
            if b._is_graph():
                o = b._owner(module)
                if o is not None:
                    return o
            elif any(m is module for m in b.leaves()):
                return b
        return None

    def __repr__(self):
        lines = [f"nn.Concat({self.dimension}) {{"]
        lines += [f"  ({i + 1}): {m!r}" for i, m in enumerate(self.modules)]
        return "\n".join(lines + ["}"])


class Copy(Module):
    """nn.Copy(intype, outtype): host<->device crossing of the reference models — a no-op here."""
    TYPENAME = "nn.Copy"

    def __init__(self, intype=None, outtype=None, forceCopy=None, dontCast=None):
        super().__init__()

    def leaves(self):
        return []


class _Param(Module):
    def __init__(self):
        super().__init__()
        self.weight = self.bias = self.gradWeight = self.gradBias = None

    def param_arrays(self):
        return [self.weight, self.bias]

    def set_param_arrays(self, arrays):
        self.weight, self.bias = arrays

    def set_grad_arrays(self, arrays):
        self.gradWeight, self.gradBias = arrays


class SpatialConvolution(_Param):
    """nn.SpatialConvolution(nInputPlane, nOutputPlane, kW, kH, dW, dH, padW, padH) — 3x3 s1 p1 (the one geometry
    models.lua uses on the G/R path) and 5x5 s1 p2 (the D network's createNxN(128, 64, 5, ..), models.lua:275,297)."""
    TYPENAME = "nn.SpatialConvolution"
    KIND = L.CONV3

    def __init__(self, nInputPlane, nOutputPlane, kW=3, kH=3, dW=1, dH=1, padW=1, padH=None):
        super().__init__()
        padH = padW if padH is None else padH
        if (kW, kH, dW, dH, padW, padH) not in ((3, 3, 1, 1, 1, 1), (5, 5, 1, 1, 2, 2)) or (kW != 3 and self.KIND != L.CONV3):
            raise L.GanrevError("only 3x3 stride-1 pad-1 (models.lua:409-436) and 5x5 stride-1 pad-2 (models.lua:297) "
                                "convolutions have a gfx950 kernel")
        self.nInputPlane, self.nOutputPlane, self.kW, self.kH = nInputPlane, nOutputPlane, kW, kH
        self.weight = np.zeros(self._wshape(), np.float32)
        self.bias = np.zeros(nOutputPlane, np.float32)
        self.reset()

    def _wshape(self):
        return (self.nOutputPlane, self.nInputPlane, self.kH, self.kW)

    def reset(self, stdv=None, rng=None):
        """nn.SpatialConvolution:reset — uniform(-stdv, stdv); a given stdv is scaled by sqrt(3) (upstream)."""
        rng = _rng(rng)
        stdv = stdv * math.sqrt(3) if stdv is not None else 1.0 / math.sqrt(self.kW * self.kH * self.nInputPlane)
        self.weight[...] = rng.uniform(-stdv, stdv, self.weight.shape)
        self.bias[...] = rng.uniform(-stdv, stdv, self.bias.shape)

    def desc(self, dims):
        c, h, w = dims
        if self.kW != 3:
            return [(L.CONVK, self.nInputPlane, self.nOutputPlane, self.kW, 0.0, 0)], (self.nOutputPlane, h, w)
        return [(self.KIND, self.nInputPlane, self.nOutputPlane, 0, 0.0, 0)], (self.nOutputPlane, h, w)

    def __repr__(self):
        p = (self.kW - 1) // 2
        return f"{self.typename}({self.nInputPlane} -> {self.nOutputPlane}, {self.kW}x{self.kH}, 1,1, {p},{p})"


class SpatialFullConvolution(SpatialConvolution):
    """nn.SpatialFullConvolution(nIn, nOut, 3,3,1,1,1,1): weight [nIn][nOut][3][3] (north_star names it; the
    reference itself up-samples with SpatialUpSamplingNearest + SpatialConvolution, models.lua:121-122)."""
    TYPENAME = "nn.SpatialFullConvolution"
    KIND = L.FULLCONV3

    def _wshape(self):
        return (self.nInputPlane, self.nOutputPlane, 3, 3)


class Linear(_Param):
    TYPENAME = "nn.Linear"

    def __init__(self, inputSize, outputSize):
        super().__init__()
        self.weight = np.zeros((outputSize, inputSize), np.float32)
        self.bias = np.zeros(outputSize, np.float32)
        self.reset()

    def reset(self, stdv=None, rng=None):
        rng = _rng(rng)
        stdv = stdv * math.sqrt(3) if stdv is not None else 1.0 / math.sqrt(self.weight.shape[1])
        self.weight[...] = rng.uniform(-stdv, stdv, self.weight.shape)
        self.bias[...] = rng.uniform(-stdv, stdv, self.bias.shape)

    def desc(self, dims):
        return [(L.LINEAR, self.weight.shape[1], self.weight.shape[0], 0, 0.0, 0)], (self.weight.shape[0], 1, 1)

    def __repr__(self):
        return f"nn.Linear({self.weight.shape[1]} -> {self.weight.shape[0]})"


class BatchNormalization(_Param):
    """nn.BatchNormalization(nFeature): eps 1e-5, momentum 0.1, affine (upstream defaults; the reference sets none)."""
    TYPENAME = "nn.BatchNormalization"

    def __init__(self, nFeature, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        if eps != 1e-5 or momentum != 0.1 or not affine:
            raise L.GanrevError("only eps=1e-5, momentum=0.1, affine BatchNormalization is implemented")
        self.nFeature = nFeature
        self.weight = np.zeros(nFeature, np.float32)
        self.bias = np.zeros(nFeature, np.float32)
        self.running_mean = np.zeros(nFeature, np.float32)
        self.running_var = np.ones(nFeature, np.float32)
        self.reset()

    def reset(self, rng=None):
        rng = _rng(rng)
        self.weight[...] = rng.uniform(0, 1, self.weight.shape)   # upstream: weight:uniform(), bias:zero()
        self.bias[...] = 0
        self.running_mean[...] = 0
        self.running_var[...] = 1

    def desc(self, dims):
        return [(L.BN, self.nFeature, 0, 0, 0.0, 0)], dims

    def __repr__(self):
        return f"{self.typename}({self.nFeature})"


class SpatialBatchNormalization(BatchNormalization):
    TYPENAME = "nn.SpatialBatchNormalization"


class _Simple(Module):
    KIND = None

    def desc(self, dims):
        return [(self.KIND, 0, 0, 0, 0.0, 0)], dims


class ELU(_Simple):
    TYPENAME = "nn.ELU"
    KIND = L.ELU

    def __init__(self, alpha=1.0, inplace=False):
        super().__init__()
        if alpha != 1.0:
            raise L.GanrevError("nn.ELU: only alpha=1 (the reference's nn.ELU()) is implemented")


class ReLU(_Simple):
    TYPENAME = "nn.ReLU"
    KIND = L.RELU

    def __init__(self, inplace=False):
        super().__init__()


class Sigmoid(_Simple):
    TYPENAME = "nn.Sigmoid"
    KIND = L.SIGMOID


class Tanh(_Simple):
    TYPENAME = "nn.Tanh"
    KIND = L.TANH


class LeakyReLU(Module):
    TYPENAME = "nn.LeakyReLU"

    def __init__(self, negval=0.01, inplace=False):
        super().__init__()
        self.negval = float(negval)

    def desc(self, dims):
        return [(L.LEAKYRELU, 0, 0, 0, self.negval, 0)], dims


class PReLU(Module):
    """nn.PReLU(nOutputPlane=0): y = x > 0 ? x : w * x with ONE learnable slope w, initial value 0.25 (models.lua:276 and
    every other activation of the D networks).  The slope is a parameter: it sits in getParameters()' flat vector where
    the module sits in the network."""
    TYPENAME = "nn.PReLU"

    def __init__(self, nOutputPlane=0):
        super().__init__()
        if nOutputPlane not in (0, None):
            raise L.GanrevError("nn.PReLU: only the shared slope (nn.PReLU() as models.lua writes it) is implemented")
        self.nOutputPlane = 0
        self.weight = np.full(1, 0.25, np.float32)
        self.gradWeight = None

    def param_arrays(self):
        return [self.weight]

    def set_param_arrays(self, arrays):
        (self.weight,) = arrays

    def set_grad_arrays(self, arrays):
        (self.gradWeight,) = arrays

    def desc(self, dims):
        return [(L.PRELU, 0, 0, 0, 0.0, 0)], dims


class Dropout(Module):
    """nn.Dropout(p=0.5, v1=false): v2 (default) scales kept units by 1/(1-p) while training and is the identity in
    evaluate(); v1 keeps the unscaled mask and multiplies by (1-p) in evaluate()."""
    TYPENAME = "nn.Dropout"

    def __init__(self, p=0.5, v1=False, inplace=False):
        super().__init__()
        self.p, self.v2, self.always_on = float(p), not v1, False

    def keepAlwaysOn(self):
        """models.lua:402-405:  drop:training(); drop.evaluate = function() end"""
        self.always_on = True
        self.train = True
        return self

    def desc(self, dims):
        flags = (L.DROPOUT_V2 if self.v2 else 0) | (L.DROPOUT_ALWAYS_ON if self.always_on else 0)
        return [(L.DROPOUT, 0, 0, 0, self.p, flags)], dims

    def __repr__(self):
        return f"nn.Dropout({self.p}{'' if self.v2 else ', v1'})"


class SpatialDropout(Module):
    TYPENAME = "nn.SpatialDropout"

    def __init__(self, p=0.5):
        super().__init__()
        self.p = float(p)

    def desc(self, dims):
        return [(L.SPATIAL_DROPOUT, 0, 0, 0, self.p, 0)], dims


class SpatialMaxPooling(Module):
    TYPENAME = "nn.SpatialMaxPooling"

    def __init__(self, kW, kH, dW=None, dH=None, padW=0, padH=0):
        super().__init__()
        dW, dH = dW or kW, dH or kH
        if (kW, kH, dW, dH, padW, padH) != (2, 2, 2, 2, 0, 0):
            raise L.GanrevError("only SpatialMaxPooling(2,2) is implemented (models.lua:422,440)")

    def desc(self, dims):
        c, h, w = dims
        return [(L.MAXPOOL2, 0, 0, 0, 0.0, 0)], (c, h // 2, w // 2)


class SpatialUpSamplingNearest(Module):
    TYPENAME = "nn.SpatialUpSamplingNearest"

    def __init__(self, scale):
        super().__init__()
        if scale != 2:
            raise L.GanrevError("only SpatialUpSamplingNearest(2) is implemented (models.lua:121,127)")

    def desc(self, dims):
        c, h, w = dims
        return [(L.UPSAMPLE2, 0, 0, 0, 0.0, 0)], (c, h * 2, w * 2)


class View(Module):
    TYPENAME = "nn.View"

    def __init__(self, *sizes):
        super().__init__()
        self.sizes = tuple(int(s) for s in sizes)

    def desc(self, dims):
        s = self.sizes + (1,) * (3 - len(self.sizes))
        if int(np.prod(s)) != int(np.prod(dims)):
            raise L.GanrevError(f"nn.View{self.sizes}: input has {int(np.prod(dims))} elements per sample")
        return [(L.VIEW, s[0], s[1], s[2], 0.0, 0)], s


class MSECriterion:
    """nn.MSECriterion (sizeAverage): train_r.lua:119,147,150."""

    def __init__(self, sizeAverage=True):
        if not sizeAverage:
            raise L.GanrevError("only sizeAverage=true is implemented")
        self.output = 0.0
        self.gradInput = None

    def forward(self, input, target):
        self.output, self._g = L.default_context().mse(input, target)
        self._key = (id(input), id(target))
        return self.output

    def backward(self, input, target):
        _, g = L.default_context().mse(input, target)
        self.gradInput = g.reshape(np.shape(input))
        return self.gradInput


class BCECriterion:
    """nn.BCECriterion (sizeAverage): train.lua:173's CRITERION, used by adversarial.lua."""

    def __init__(self, sizeAverage=True):
        if not sizeAverage:
            raise L.GanrevError("only sizeAverage=true is implemented")
        self.output = 0.0
        self.gradInput = None

    def forward(self, input, target):
        self.output, _ = L.default_context().bce(input, target, want_grad=False)
        return self.output

    def backward(self, input, target):
        _, g = L.default_context().bce(input, target)
        self.gradInput = g.reshape(np.shape(input))
        return self.gradInput


class CosineDistance:
    """nn.CosineDistance on a pair of vectors (apply_r.lua:396-400)."""

    def forward(self, pair):
        a, b = pair
        self.output = np.array([L.default_context().cosine_similarity(a, b)], dtype=np.float32)
        return self.output
