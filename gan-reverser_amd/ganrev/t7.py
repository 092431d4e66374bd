"""Torch7 binary serialisation (`torch.save` / `torch.load`, the `.net` / `.t7` files of the reference): reader, writer, and the
conversion between deserialised `nn.*` / `cudnn.*` module trees and the ganrev mirror modules.

Where the reference uses it:  train_r.lua:68 and apply_r.lua:62  `torch.load(OPT.G)` -> {G=<nn.Sequential>, opt={noiseDim, noiseMethod,
height, width, colorSpace, ...}, D=..., ...} (written by train.lua:256);  train_r.lua:234  `torch.save(filename, {R=MODEL_R,
opt=OPT})`;  apply_r.lua:92,101 load that file back.

The format is defined by Torch7's File.lua / the TH storages, not by the reference (which holds no sample file and no spec).
RESTATED FROM MEMORY [upstream: torch7 File.lua readObject/writeObject, default binary little-endian encoding]:
  object  := int32 type, payload
  type 0 nil | 1 number: float64 | 2 string: int32 length, bytes | 5 boolean: int32
  type 3 table: int32 index (reference id: a repeated index refers back to the first occurrence), int32 n, n x (key object, value object)
  type 4 torch object: int32 index, string version ("V 1"), string class name, then
        torch.XTensor : int32 ndim, int64 size[ndim], int64 stride[ndim], int64 storageOffset (1-based), object storage (or nil)
        torch.XStorage: int64 n, n raw elements
        any other class (nn.*, cudnn.*): one object - the table of its fields
  type 6 / 7 / 8 functions (dumped Lua bytecode): int32 index, then int32 length + bytes, then an upvalue table (7, 8); skipped
No Torch7 runtime exists in the build image: this module is verified against its own writer and against byte strings assembled
by hand from the layout above (tests/test_host_logic.py) - parity with real files is UNPINNED until a sample file is available.
"""
import struct

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN, TYPE_FUNCTION, TYPE_RECUR_FUNCTION_LEGACY, TYPE_RECUR_FUNCTION = range(9)

_DTYPES = {"Float": np.float32, "Double": np.float64, "Long": np.int64, "Int": np.int32, "Short": np.int16,
           "Char": np.int8, "Byte": np.uint8, "Cuda": np.float32, "Half": np.float16}


class TorchObject:
    """A deserialised non-tensor torch class instance (an nn module, a criterion ...): its class name and field table."""

    def __init__(self, typename, fields=None):
        self.typename = typename
        self.fields = fields if fields is not None else {}

    def __getattr__(self, k):
        try:
            return self.__dict__["fields"][k]
        except KeyError:
            raise AttributeError(k)

    def __repr__(self):
        return f"<{self.typename}>"


class LuaFunction:
    def __init__(self, dumped, upvalues=None):
        self.dumped, self.upvalues = dumped, upvalues


class Reader:
    def __init__(self, data):
        self.b = memoryview(data)
        self.p = 0
        self.memo = {}

    def _take(self, n):
        if self.p + n > len(self.b):
            raise EOFError("truncated Torch7 file")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def int(self):
        return struct.unpack("<i", self._take(4))[0]

    def long(self):
        return struct.unpack("<q", self._take(8))[0]

    def double(self):
        return struct.unpack("<d", self._take(8))[0]

    def string(self):
        return bytes(self._take(self.int())).decode("latin-1")

    def obj(self):
        t = self.int()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = self.double()
            return int(v) if v.is_integer() and abs(v) < 2 ** 53 else v
        if t == TYPE_STRING:
            return self.string()
        if t == TYPE_BOOLEAN:
            return self.int() == 1
        if t in (TYPE_TABLE, TYPE_TORCH, TYPE_FUNCTION, TYPE_RECUR_FUNCTION, TYPE_RECUR_FUNCTION_LEGACY):
            idx = self.int()
            if idx in self.memo:
                return self.memo[idx]
            if t == TYPE_TABLE:
                out = {}
                self.memo[idx] = out
                for _ in range(self.int()):
                    k = self.obj()
                    out[k] = self.obj()
                res = _listify(out)     # (a table that refers to itself keeps the dict form inside the cycle)
                self.memo[idx] = res
                return res
            if t == TYPE_TORCH:
                version = self.string()
                cls = self.string() if version.startswith("V ") else version
                return self._torch(idx, cls)
            fn = LuaFunction(bytes(self._take(self.int())))
            self.memo[idx] = fn
            if t != TYPE_FUNCTION:
                fn.upvalues = self.obj()
            return fn
        raise ValueError(f"unknown Torch7 type tag {t} at byte {self.p - 4}")

    def _torch(self, idx, cls):
        kind = cls[len("torch."):] if cls.startswith("torch.") else ""
        if kind.endswith("Tensor") and kind[:-6] in _DTYPES:
            nd = self.int()
            size = [self.long() for _ in range(nd)]
            stride = [self.long() for _ in range(nd)]
            off = self.long() - 1
            storage = self.obj()
            if storage is None or nd == 0:
                arr = np.zeros(size if nd else (0,), _DTYPES[kind[:-6]])
            else:
                arr = np.lib.stride_tricks.as_strided(np.asarray(storage)[off:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
            self.memo[idx] = arr
            return arr
        if kind.endswith("Storage") and kind[:-7] in _DTYPES:
            n = self.long()
            dt = np.dtype(_DTYPES[kind[:-7]])
            arr = np.frombuffer(self._take(n * dt.itemsize), dtype=dt.newbyteorder("<")).astype(dt).view(Storage)
            self.memo[idx] = arr
            return arr
        o = TorchObject(cls)
        self.memo[idx] = o
        fields = self.obj()
        o.fields = fields if isinstance(fields, dict) else ({} if fields is None else {i + 1: v for i, v in enumerate(fields)})
        return o


class Storage(np.ndarray):
    """A torch.XStorage (flat, typed, no shape) as opposed to a torch.XTensor: the reader returns storages that appear as OBJECTS
    (nn.Concat.size and nn.View.size are torch.LongStorage in Torch7's nn: `self.size:resize(...):copy(outs[1]:size())` needs a
    storage, a LongTensor there breaks the reference's first forward of a loaded D) as this ndarray view, and the writer emits a
    Storage as torch.XStorage.  Use storage(values, dtype) to make one."""


def storage(values, dtype=np.int64):
    return np.ascontiguousarray(values, dtype=dtype).reshape(-1).view(Storage)


def _listify(d):
    """A Lua table whose keys are exactly 1..n reads back as a Python list (nn.Sequential.modules, size tables ...)."""
    n = len(d)
    if n and all(isinstance(k, int) for k in d) and set(d) == set(range(1, n + 1)):
        return [d[i] for i in range(1, n + 1)]
    return d


def load(path_or_bytes):
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
    return Reader(data).obj()


# ----------------------------------------------------------------------------------------------------------------- writer
class Writer:
    def __init__(self):
        self.out = bytearray()
        self.next = 1
        self.seen = {}

    def int(self, v):
        self.out += struct.pack("<i", v)

    def long(self, v):
        self.out += struct.pack("<q", v)

    def string(self, s):
        b = s.encode("latin-1")
        self.int(len(b))
        self.out += b

    def _index(self, o):
        if id(o) in self.seen:
            self.int(self.seen[id(o)])
            return False
        self.seen[id(o)] = self.next
        self.int(self.next)
        self.next += 1
        return True

    def obj(self, o):
        if o is None:
            self.int(TYPE_NIL)
        elif isinstance(o, bool):
            self.int(TYPE_BOOLEAN); self.int(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.int(TYPE_NUMBER); self.out += struct.pack("<d", float(o))
        elif isinstance(o, str):
            self.int(TYPE_STRING); self.string(o)
        elif isinstance(o, Storage):
            self._storage(o)
        elif isinstance(o, np.ndarray):
            self._tensor(o)
        elif isinstance(o, (list, tuple)):
            self.int(TYPE_TABLE)
            if self._index(o):
                self.int(len(o))
                for i, v in enumerate(o):
                    self.obj(i + 1); self.obj(v)
        elif isinstance(o, dict):
            self.int(TYPE_TABLE)
            if self._index(o):
                self.int(len(o))
                for k, v in o.items():
                    self.obj(k); self.obj(v)
        elif isinstance(o, TorchObject):
            self.int(TYPE_TORCH)
            if self._index(o):
                self.string("V 1"); self.string(o.typename)
                self.obj(o.fields)
        else:
            raise TypeError(f"cannot serialise {type(o).__name__}")

    _NAMES = {np.dtype(np.float32): "Float", np.dtype(np.float64): "Double", np.dtype(np.int64): "Long", np.dtype(np.int32): "Int",
              np.dtype(np.uint8): "Byte"}

    def _storage(self, a):
        """a torch.XStorage object of its own (not the storage behind a tensor): int64 n, n raw elements"""
        self.int(TYPE_TORCH)
        if not self._index(a):
            return
        self.string("V 1"); self.string(f"torch.{self._NAMES[a.dtype]}Storage")
        self.long(a.size)
        self.out += np.asarray(a).astype(a.dtype.newbyteorder("<")).tobytes()

    def _tensor(self, a):
        name = {np.dtype(np.float32): "Float", np.dtype(np.float64): "Double", np.dtype(np.int64): "Long", np.dtype(np.int32): "Int",
                np.dtype(np.uint8): "Byte"}[a.dtype]
        self.int(TYPE_TORCH)
        if not self._index(a):
            return
        self.string("V 1"); self.string(f"torch.{name}Tensor")
        a = np.ascontiguousarray(a)
        self.int(a.ndim)
        for s in a.shape:
            self.long(s)
        for s in a.strides:
            self.long(s // a.itemsize)
        self.long(1)
        if a.size == 0:
            self.int(TYPE_NIL)
            return
        self.int(TYPE_TORCH); self.int(self.next); self.next += 1
        self.string("V 1"); self.string(f"torch.{name}Storage")
        self.long(a.size)
        self.out += a.astype(a.dtype.newbyteorder("<")).tobytes()


def dumps(o):
    w = Writer()
    w.obj(o)
    return bytes(w.out)


def save(path, o):
    with open(path, "wb") as f:
        f.write(dumps(o))


# ------------------------------------------------------------------------------------- nn module trees <-> ganrev modules
def to_model(obj):
    """Deserialised nn.Sequential (models.lua:104-143, 389-464 as saved by train.lua:256 / train_r.lua:234) -> ganrev.nn.Sequential
    with the stored weights and running statistics.  cudnn.* classes map to their nn.* counterparts (what the reference's
    `cudnn.convert(model, nn)` does); nn.Copy (host<->device transfer) is dropped."""
    from . import nn
    t = obj.typename
    f = obj.fields
    if t == "nn.Sequential":
        seq = nn.Sequential()
        for m in f.get("modules", []):
            sub = to_model(m)
            if sub is not None:
                seq.add(sub)
        return seq
    if t == "nn.Copy":
        return None
    if t == "nn.Concat":                     # the D network (models.lua:293-321; train.lua:256 saves D next to G)
        cat = nn.Concat(int(f["dimension"]))
        for m in f.get("modules", []):
            sub = to_model(m)
            if sub is not None:
                cat.add(sub)
        return cat
    if t == "nn.PReLU":
        m = nn.PReLU(int(f.get("nOutputPlane", 0)))
        m.weight[...] = np.asarray(f["weight"], np.float32).reshape(m.weight.shape)
        return m
    if t in ("nn.SpatialConvolution", "cudnn.SpatialConvolution", "nn.SpatialConvolutionMM"):
        m = nn.SpatialConvolution(f["nInputPlane"], f["nOutputPlane"], f["kW"], f["kH"], f.get("dW", 1), f.get("dH", 1), f.get("padW", 0), f.get("padH", f.get("padW", 0)))
        m.weight[...] = np.asarray(f["weight"], np.float32).reshape(m.weight.shape)      # SpatialConvolutionMM stores it 2-D
        m.bias[...] = f["bias"]
        return m
    if t == "nn.SpatialFullConvolution":
        m = nn.SpatialFullConvolution(f["nInputPlane"], f["nOutputPlane"], f["kW"], f["kH"], f.get("dW", 1), f.get("dH", 1), f.get("padW", 0), f.get("padH", f.get("padW", 0)))
        m.weight[...] = np.asarray(f["weight"], np.float32).reshape(m.weight.shape); m.bias[...] = f["bias"]
        return m
    if t == "nn.Linear":
        w = np.asarray(f["weight"], np.float32)
        m = nn.Linear(w.shape[1], w.shape[0])
        m.weight[...] = w; m.bias[...] = f["bias"]
        return m
    if t in ("nn.BatchNormalization", "nn.SpatialBatchNormalization", "cudnn.SpatialBatchNormalization", "cudnn.BatchNormalization"):
        cls = nn.SpatialBatchNormalization if "Spatial" in t else nn.BatchNormalization
        rm = np.asarray(f["running_mean"], np.float32)
        m = cls(rm.size, f.get("eps", 1e-5), f.get("momentum", 0.1), f.get("affine", True))
        m.weight[...] = f["weight"]; m.bias[...] = f["bias"]; m.running_mean[...] = rm
        if "running_var" in f:
            m.running_var[...] = f["running_var"]
        else:                                     # older nn revisions stored running_std = 1 / sqrt(var + eps)
            m.running_var[...] = 1.0 / np.square(np.asarray(f["running_std"], np.float64)) - f.get("eps", 1e-5)
        return m
    simple = {"nn.ELU": nn.ELU, "nn.ReLU": nn.ReLU, "cudnn.ReLU": nn.ReLU, "nn.Sigmoid": nn.Sigmoid, "cudnn.Sigmoid": nn.Sigmoid,
              "nn.Tanh": nn.Tanh, "cudnn.Tanh": nn.Tanh}
    if t in simple:
        return simple[t]()
    if t == "nn.LeakyReLU":
        return nn.LeakyReLU(f.get("negval", 0.01))
    if t == "nn.Dropout":
        m = nn.Dropout(f.get("p", 0.5), not f.get("v2", True))
        # models.lua:404 replaces :evaluate with a no-op closure on the fixer's first layer (a dumped Lua function in the file);
        # this package's own writer cannot dump Lua bytecode and records the fact in a field of its own
        if isinstance(f.get("evaluate"), LuaFunction) or f.get("ganrev_always_on"):
            m.keepAlwaysOn()
        return m
    if t == "nn.SpatialDropout":
        return nn.SpatialDropout(f.get("p", 0.5))
    if t in ("nn.SpatialMaxPooling", "cudnn.SpatialMaxPooling"):
        return nn.SpatialMaxPooling(f["kW"], f["kH"], f.get("dW"), f.get("dH"), f.get("padW", 0), f.get("padH", 0))
    if t == "nn.SpatialUpSamplingNearest":
        return nn.SpatialUpSamplingNearest(f["scale_factor"])
    if t == "nn.View":
        size = f.get("size")
        return nn.View(*[int(v) for v in np.asarray(size).ravel()])
    raise ValueError(f"no ganrev counterpart for {t}")


def from_model(model):
    """ganrev module tree -> TorchObject tree with the field names nn's constructors create (weights, biases, running statistics,
    geometry); gradient and buffer tensors are written empty, as NN_UTILS.prepareNetworkForSave (utils/nn_utils.lua:395-413)
    leaves them."""
    from . import nn
    empty = np.zeros(0, np.float32)
    base = {"_type": "torch.FloatTensor", "train": bool(getattr(model, "train", True)), "output": empty, "gradInput": empty}
    t = model.typename
    if isinstance(model, nn.Sequential):
        return TorchObject("nn.Sequential", dict(base, modules=[from_model(m) for m in model.modules]))
    if isinstance(model, nn.Concat):
        return TorchObject("nn.Concat", dict(base, dimension=model.dimension, size=storage([]), modules=[from_model(m) for m in model.modules]))
    f = dict(base)
    if isinstance(model, nn.PReLU):
        f.update(nOutputPlane=0, weight=model.weight, gradWeight=empty, gradWeightBuf=empty, gradWeightBuf2=empty)
    elif isinstance(model, nn.SpatialConvolution):
        pad = (model.kW - 1) // 2
        f.update(nInputPlane=model.nInputPlane, nOutputPlane=model.nOutputPlane, kW=model.kW, kH=model.kH, dW=1, dH=1, padW=pad, padH=pad,
                 weight=model.weight, bias=model.bias, gradWeight=empty, gradBias=empty, finput=empty, fgradInput=empty)
        if isinstance(model, nn.SpatialFullConvolution):
            f.update(adjW=0, adjH=0)
        t = "nn.SpatialFullConvolution" if isinstance(model, nn.SpatialFullConvolution) else "nn.SpatialConvolution"
    elif isinstance(model, nn.Linear):
        f.update(weight=model.weight, bias=model.bias, gradWeight=empty, gradBias=empty)
    elif isinstance(model, nn.BatchNormalization):
        f.update(weight=model.weight, bias=model.bias, gradWeight=empty, gradBias=empty, running_mean=model.running_mean,
                 running_var=model.running_var, eps=1e-5, momentum=0.1, affine=True, nDim=4 if isinstance(model, nn.SpatialBatchNormalization) else 2)
    elif isinstance(model, nn.LeakyReLU):
        f.update(negval=model.negval, inplace=False)
    elif isinstance(model, nn.Dropout):
        f.update(p=model.p, v2=model.v2, inplace=False, noise=empty)
        if model.always_on:
            f["ganrev_always_on"] = True      # a Torch7 loader must re-apply `drop.evaluate = function() end` (models.lua:402-405)
    elif isinstance(model, nn.SpatialDropout):
        f.update(p=model.p, noise=empty)
    elif isinstance(model, nn.SpatialMaxPooling):
        f.update(kW=2, kH=2, dW=2, dH=2, padW=0, padH=0, ceil_mode=False, indices=empty)
    elif isinstance(model, nn.SpatialUpSamplingNearest):
        f.update(scale_factor=2, inputSize=np.zeros(4, np.int64), outputSize=np.zeros(4, np.int64))
    elif isinstance(model, nn.View):
        f.update(size=storage(model.sizes), numElements=int(np.prod(model.sizes)))            # nn.View keeps a torch.LongStorage
    elif isinstance(model, nn.ELU):
        f.update(alpha=1, inplace=False)
    return TorchObject(t if t.startswith("nn.") else "nn." + t.split(".")[-1], f)


def load_checkpoint(path):
    """torch.load of a reference checkpoint -> dict with every nn.Sequential value converted (keys G, R, D ... as saved) and the
    rest (opt table ...) as plain Python values."""
    top = load(path)
    out = {}
    for k, v in (top.items() if isinstance(top, dict) else enumerate(top)):
        if isinstance(v, TorchObject) and v.typename == "nn.Sequential":
            try:
                out[k] = to_model(v)
            except Exception as e:          # a layer this path has no kernel for (e.g. SpatialAveragePooling of create_D_default): keep the raw tree
                out[k] = v
                out.setdefault("_unconverted", {})[k] = str(e)
        else:
            out[k] = v
    return out


def save_checkpoint(path, **entries):
    """train_r.lua:234  torch.save(filename, {R=MODEL_R, opt=OPT}) - modules are converted with from_model."""
    from . import nn
    save(path, {k: (from_model(v) if isinstance(v, nn.Module) else v) for k, v in entries.items()})
