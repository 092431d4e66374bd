"""ctypes binding of the CPU oracle (oracle/libganrev_oracle.so).  TEST INFRASTRUCTURE ONLY — see ganrev_oracle.h.
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GANREV_ORACLE_SO: load another build of the same sources instead (tests/test_sanitizers.py: the ASan + UBSan build)
_SO = os.environ.get("GANREV_ORACLE_SO") or os.path.join(_HERE, "libganrev_oracle.so")


def build(force=False):
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class GoLayer(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("p", C.c_float), ("flags", C.c_int32)]


class GoHyper(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("l1", C.c_double), ("l2", C.c_double), ("clamp", C.c_double)]

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, l1=0.0, l2=1e-4, clamp=1.0):
        super().__init__(lr, beta1, beta2, eps, l1, l2, clamp)


_lib = None
_P = C.c_void_p


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.go_net_create.restype = _P
        L.go_net_create.argtypes = [C.POINTER(GoLayer), C.c_int, C.c_int, C.c_int, C.c_int]
        L.go_net_destroy.argtypes = [_P]
        L.go_net_param_count.restype = C.c_int64
        L.go_net_param_count.argtypes = [_P]
        for f in ("go_net_params", "go_net_grads"):
            getattr(L, f).restype = C.POINTER(C.c_float)
            getattr(L, f).argtypes = [_P]
        L.go_net_out_dim.argtypes = [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.go_net_n_bn.argtypes = [_P]
        for f in ("go_net_bn_running_mean", "go_net_bn_running_var"):
            getattr(L, f).restype = C.POINTER(C.c_float)
            getattr(L, f).argtypes = [_P, C.c_int, C.POINTER(C.c_int)]
        L.go_net_set_training.argtypes = [_P, C.c_int]
        L.go_net_set_bn_groups.argtypes = [_P, C.c_int]
        L.go_net_set_mask.argtypes = [_P, C.c_int, _P, C.c_int64]
        L.go_net_mask_size.restype = C.c_int64
        L.go_net_mask_size.argtypes = [_P, C.c_int, C.c_int]
        L.go_net_zero_grads.argtypes = [_P]
        L.go_net_set_lean.argtypes = [_P, C.c_int]
        L.go_net_forward.argtypes = [_P, _P, C.c_int, _P]
        L.go_net_backward.argtypes = [_P, _P, _P, C.c_int, _P]
        L.go_net_layer_output.restype = C.POINTER(C.c_float)
        L.go_net_layer_output.argtypes = [_P, C.c_int, C.POINTER(C.c_int64)]
        L.go_net_get_pool_index.restype = C.c_int64
        L.go_net_get_pool_index.argtypes = [_P, C.c_int, _P, C.c_int64]
        L.go_net_force_pool_index.argtypes = [_P, C.c_int, _P, C.c_int64]
        L.go_net_force_act_side.argtypes = [_P, C.c_int, _P, C.c_int64]
        L.go_mse.restype = C.c_double
        L.go_mse.argtypes = [_P, _P, C.c_int64, _P]
        L.go_mse_scaled.restype = C.c_double
        L.go_mse_scaled.argtypes = [_P, _P, C.c_int64, C.c_int64, _P]
        L.go_bce.restype = C.c_double
        L.go_bce.argtypes = [_P, _P, C.c_int64, _P]
        L.go_penalty_clamp_adam.argtypes = [_P, _P, _P, _P, C.c_int64, C.POINTER(GoHyper), C.c_int, C.POINTER(C.c_double)]
        L.go_train_r_step.argtypes = [_P, _P, _P, C.c_int, C.POINTER(GoHyper), _P, _P, C.c_int, C.POINTER(C.c_double), _P]
        L.go_cosine_similarity.restype = C.c_float
        L.go_cosine_similarity.argtypes = [_P, _P, C.c_int, C.c_int]
        L.go_cosine_topk.argtypes = [_P, C.c_int64, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int]
        L.go_l2_distance_rows.argtypes = [_P, _P, C.c_int64, C.c_int64, _P]
        L.go_kmeans.argtypes = [_P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P]
        L.go_cosine_assign.argtypes = [_P, C.c_int64, C.c_int, _P, C.c_int, C.c_int, _P, _P]
        L.go_set_conv_impl.argtypes = [C.c_int]
        L.go_conv3_forward.argtypes = [_P, _P, _P, _P] + [C.c_int] * 5
        L.go_conv3_backward_data.argtypes = [_P, _P, _P] + [C.c_int] * 5
        L.go_conv3_backward_weight.argtypes = [_P, _P, _P, _P] + [C.c_int] * 5
        L.go_convk_forward.argtypes = [_P, _P, _P, _P] + [C.c_int] * 6
        L.go_convk_backward_data.argtypes = [_P, _P, _P] + [C.c_int] * 6
        L.go_convk_backward_weight.argtypes = [_P, _P, _P, _P] + [C.c_int] * 6
        L.go_linear_forward.argtypes = [_P, _P, _P, _P] + [C.c_int] * 3
        L.go_linear_backward_data.argtypes = [_P, _P, _P] + [C.c_int] * 3
        L.go_linear_backward_weight.argtypes = [_P, _P, _P, _P] + [C.c_int] * 3
        L.go_bn_forward_train.argtypes = [_P] * 8 + [C.c_int] * 4
        L.go_bn_forward_eval.argtypes = [_P] * 6 + [C.c_int] * 3
        L.go_bn_backward_train.argtypes = [_P] * 8 + [C.c_int] * 4
        _lib = L
    return _lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Net:
    """nn.Sequential restatement (go_net)."""

    def __init__(self, descs, in_dims):
        self.L = lib()
        arr = (GoLayer * len(descs))(*[GoLayer(*d) for d in descs])
        c, h, w = in_dims
        self.h = self.L.go_net_create(arr, len(descs), int(c), int(h), int(w))
        if not self.h:
            raise ValueError("oracle: shape mismatch in layer list")
        self.in_dims = tuple(in_dims)
        self.n_layers = len(descs)
        oc, oh, ow = C.c_int(), C.c_int(), C.c_int()
        self.L.go_net_out_dim(self.h, C.byref(oc), C.byref(oh), C.byref(ow))
        self.out_dims = (oc.value, oh.value, ow.value)
        self.n_params = int(self.L.go_net_param_count(self.h))
        self.params = np.ctypeslib.as_array(self.L.go_net_params(self.h), shape=(max(self.n_params, 1),))[:self.n_params]
        self.grads = np.ctypeslib.as_array(self.L.go_net_grads(self.h), shape=(max(self.n_params, 1),))[:self.n_params]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_net_destroy(self.h)
            self.h = None

    def bn_running(self, i):
        n = C.c_int()
        m = self.L.go_net_bn_running_mean(self.h, i, C.byref(n))
        v = self.L.go_net_bn_running_var(self.h, i, C.byref(n))
        return np.ctypeslib.as_array(m, shape=(n.value,)), np.ctypeslib.as_array(v, shape=(n.value,))

    def n_bn(self):
        return self.L.go_net_n_bn(self.h)

    def set_training(self, t):
        self.L.go_net_set_training(self.h, int(bool(t)))

    def set_bn_groups(self, g):
        self.L.go_net_set_bn_groups(self.h, int(g))

    def mask_size(self, layer, batch):
        return int(self.L.go_net_mask_size(self.h, layer, batch))

    def set_mask(self, layer, keep):
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        rc = self.L.go_net_set_mask(self.h, layer, _p(keep), keep.size)
        assert rc == 0, rc

    def zero_grads(self):
        self.L.go_net_zero_grads(self.h)

    def set_lean(self, lean=True):
        """backward frees every buffer it has consumed: layer outputs are unavailable after a backward"""
        self.L.go_net_set_lean(self.h, int(bool(lean)))

    @staticmethod
    def _shape(d):
        c, h, w = d
        return (c,) if (h == 1 and w == 1) else (c, h, w)

    def forward(self, x):
        x = f32(x)
        out = np.empty((x.shape[0],) + self._shape(self.out_dims), np.float32)
        rc = self.L.go_net_forward(self.h, _p(x), x.shape[0], _p(out))
        assert rc == 0, f"oracle forward rc={rc}"
        return out

    def backward(self, x, gout, want_gin=True):
        x, gout = f32(x), f32(gout)
        gin = np.empty_like(x) if want_gin else None
        rc = self.L.go_net_backward(self.h, _p(x), _p(gout), x.shape[0], _p(gin))
        assert rc == 0, f"oracle backward rc={rc}"
        return gin

    def pool_index(self, layer):
        """argmax (0..3) nn.SpatialMaxPooling `layer` took in the last forward"""
        n = int(self.L.go_net_get_pool_index(self.h, layer, None, 0))
        assert n > 0, f"layer {layer} is not a max-pool with a recorded forward"
        out = np.empty(n, np.uint8)
        assert self.L.go_net_get_pool_index(self.h, layer, _p(out), n) == n
        return out

    def force_pool_index(self, layer, idx):
        """use `idx` instead of computing the argmax in the following forwards (None: compute again)"""
        if idx is None:
            rc = self.L.go_net_force_pool_index(self.h, layer, None, 0)
        else:
            idx = np.ascontiguousarray(idx, dtype=np.uint8)
            rc = self.L.go_net_force_pool_index(self.h, layer, _p(idx), idx.size)
        assert rc == 0, rc

    def force_act_side(self, layer, side):
        """ReLU / LeakyReLU `layer`: side[k] != 0 = treat input k as positive in the following backwards (None: its own sign)"""
        if side is None:
            rc = self.L.go_net_force_act_side(self.h, layer, None, 0)
        else:
            side = np.ascontiguousarray(side, dtype=np.uint8)
            rc = self.L.go_net_force_act_side(self.h, layer, _p(side), side.size)
        assert rc == 0, rc

    def layer_output(self, layer):
        n = C.c_int64()
        p = self.L.go_net_layer_output(self.h, layer, C.byref(n))
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def set_threads(n):
    lib().go_set_threads(int(n))
    return lib().go_get_max_threads()


def set_conv_impl(impl):
    """3x3 convolutions of every later call: "direct" (0; the parity oracle's loop nests, oracle_blas.c) or "mm" (1; im2col +
    blocked sgemm per sample, oracle_mm.c: the structure of THNN's SpatialConvolutionMM - the CPU BASELINE bench.py reports).
    Returns the previous setting."""
    prev = lib().go_get_conv_impl()
    lib().go_set_conv_impl({"direct": 0, "mm": 1}.get(impl, impl))
    return prev


def mse(x, t, n_global=None):
    x, t = f32(x), f32(t)
    g = np.empty_like(x)
    loss = lib().go_mse_scaled(_p(x), _p(t), x.size, int(n_global or x.size), _p(g))
    return loss, g


def bce(x, t):
    """nn.BCECriterion (sizeAverage): (loss, gradInput)"""
    x, t = f32(x), f32(t)
    g = np.empty_like(x)
    loss = lib().go_bce(_p(x), _p(t), x.size, _p(g))
    return loss, g


def penalty_clamp_adam(theta, g, m, v, hyper, t):
    pen = C.c_double()
    lib().go_penalty_clamp_adam(_p(theta), _p(g), _p(m), _p(v), theta.size, C.byref(hyper), int(t), C.byref(pen))
    return pen.value


def train_r_step(gnet, rnet, noise, hyper, m, v, t, want_images=False):
    noise = f32(noise)
    B = noise.shape[0]
    loss = C.c_double()
    images = np.empty((B,) + Net._shape(gnet.out_dims), np.float32) if want_images else None
    rc = lib().go_train_r_step(gnet.h, rnet.h, _p(noise), B, C.byref(hyper), _p(m), _p(v), int(t), C.byref(loss), _p(images))
    assert rc == 0, f"oracle train_r_step rc={rc}"
    return loss.value, images


def cosine_similarity(a, b, accumulate_in_float=False):
    a, b = f32(a).ravel(), f32(b).ravel()
    return float(lib().go_cosine_similarity(_p(a), _p(b), a.size, int(accumulate_in_float)))


def cosine_topk(emb, query_rows, k, accumulate_in_float=False):
    emb = f32(emb)
    q = np.ascontiguousarray(query_rows, dtype=np.int64)
    n, d = emb.shape
    k = min(k, n)
    idx = np.empty((q.size, k), np.int64)
    sc = np.empty((q.size, k), np.float32)
    lib().go_cosine_topk(_p(emb), n, d, _p(q), q.size, k, _p(idx), _p(sc), int(accumulate_in_float))
    return idx, sc


def l2_distance_rows(a, b):
    a, b = f32(a), f32(b)
    n = a.shape[0]
    a2, b2 = a.reshape(n, -1), b.reshape(n, -1)
    out = np.empty(n, np.float64)
    lib().go_l2_distance_rows(_p(a2), _p(b2), n, a2.shape[1], _p(out))
    return out


def kmeans(x, k, niter, centroids0):
    """unsup.kmeans(x, k, niter) from the given initial centroids -> (centroids, totalcounts, labels of the last iteration)."""
    x = f32(x)
    n, d = x.shape
    cent = np.array(centroids0, dtype=np.float32, order="C", copy=True).reshape(k, d)
    tot = np.zeros(k, np.float32)
    lab = np.zeros(n, np.int32)
    lib().go_kmeans(_p(x), n, d, k, niter, _p(cent), _p(tot), _p(lab))
    return cent, tot, lab


def cosine_assign(x, centroids, take_min=True):
    x, cent = f32(x), f32(centroids)
    n, d = x.shape
    lab = np.zeros(n, np.int32)
    sim = np.zeros(n, np.float32)
    lib().go_cosine_assign(_p(x), n, d, _p(cent), cent.shape[0], int(take_min), _p(lab), _p(sim))
    return lab, sim


# ---- single operators
def conv3_forward(x, w, b):
    x, w = f32(x), f32(w)
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    out = np.empty((B, Cout, H, W), np.float32)
    lib().go_conv3_forward(_p(x), _p(w), _p(f32(b)) if b is not None else None, _p(out), B, Cin, Cout, H, W)
    return out


def conv3_backward_data(gout, w):
    gout, w = f32(gout), f32(w)
    B, Cout, H, W = gout.shape
    Cin = w.shape[1]
    gin = np.empty((B, Cin, H, W), np.float32)
    lib().go_conv3_backward_data(_p(gout), _p(w), _p(gin), B, Cin, Cout, H, W)
    return gin


def conv3_backward_weight(x, gout):
    x, gout = f32(x), f32(gout)
    B, Cin, H, W = x.shape
    Cout = gout.shape[1]
    gw = np.zeros((Cout, Cin, 3, 3), np.float32)
    gb = np.zeros(Cout, np.float32)
    lib().go_conv3_backward_weight(_p(x), _p(gout), _p(gw), _p(gb), B, Cin, Cout, H, W)
    return gw, gb


def convk_forward(x, w, b):
    """nn.SpatialConvolution(Cin, Cout, K, K, 1, 1, (K-1)/2, (K-1)/2) with the window size taken from w."""
    x, w = f32(x), f32(w)
    B, Cin, H, W = x.shape
    Cout, K = w.shape[0], w.shape[2]
    out = np.empty((B, Cout, H, W), np.float32)
    lib().go_convk_forward(_p(x), _p(w), _p(f32(b)) if b is not None else None, _p(out), B, Cin, Cout, H, W, K)
    return out


def convk_backward_data(gout, w):
    gout, w = f32(gout), f32(w)
    B, Cout, H, W = gout.shape
    Cin, K = w.shape[1], w.shape[2]
    gin = np.empty((B, Cin, H, W), np.float32)
    lib().go_convk_backward_data(_p(gout), _p(w), _p(gin), B, Cin, Cout, H, W, K)
    return gin


def convk_backward_weight(x, gout, K):
    x, gout = f32(x), f32(gout)
    B, Cin, H, W = x.shape
    Cout = gout.shape[1]
    gw = np.zeros((Cout, Cin, K, K), np.float32)
    gb = np.zeros(Cout, np.float32)
    lib().go_convk_backward_weight(_p(x), _p(gout), _p(gw), _p(gb), B, Cin, Cout, H, W, K)
    return gw, gb


def linear_forward(x, w, b):
    x, w, b = f32(x), f32(w), f32(b)
    out = np.empty((x.shape[0], w.shape[0]), np.float32)
    lib().go_linear_forward(_p(x), _p(w), _p(b), _p(out), x.shape[0], w.shape[1], w.shape[0])
    return out


def linear_backward(x, gout, w):
    x, gout, w = f32(x), f32(gout), f32(w)
    gin = np.empty_like(x)
    gw = np.zeros_like(w)
    gb = np.zeros(w.shape[0], np.float32)
    lib().go_linear_backward_data(_p(gout), _p(w), _p(gin), x.shape[0], w.shape[1], w.shape[0])
    lib().go_linear_backward_weight(_p(x), _p(gout), _p(gw), _p(gb), x.shape[0], w.shape[1], w.shape[0])
    return gin, gw, gb


def bn_forward_train(x, gamma, beta, run_mean, run_var, groups=1):
    x = f32(x)
    B, Cc = x.shape[:2]
    HW = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    out = np.empty_like(x)
    sm = np.empty(groups * Cc, np.float32)
    si = np.empty(groups * Cc, np.float32)
    lib().go_bn_forward_train(_p(x), _p(f32(gamma)), _p(f32(beta)), _p(out), _p(sm), _p(si), _p(run_mean), _p(run_var), B, Cc, HW, groups)
    return out, sm, si


def bn_forward_eval(x, gamma, beta, run_mean, run_var):
    x = f32(x)
    B, Cc = x.shape[:2]
    HW = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    out = np.empty_like(x)
    lib().go_bn_forward_eval(_p(x), _p(f32(gamma)), _p(f32(beta)), _p(out), _p(f32(run_mean)), _p(f32(run_var)), B, Cc, HW)
    return out


def bn_backward_train(x, gout, gamma, sm, si, groups=1):
    x, gout = f32(x), f32(gout)
    B, Cc = x.shape[:2]
    HW = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
    gin = np.empty_like(x)
    gg = np.zeros(Cc, np.float32)
    gb = np.zeros(Cc, np.float32)
    lib().go_bn_backward_train(_p(x), _p(gout), _p(f32(gamma)), _p(gin), _p(gg), _p(gb), _p(sm), _p(si), B, Cc, HW, groups)
    return gin, gg, gb


def from_model(model, in_dims):
    """Build the oracle twin of a ganrev nn.Sequential (same layer descriptors), copying parameters and BN running stats."""
    descs, index = model._descs(tuple(in_dims))
    net = Net(descs, in_dims)
    flat = model._flat[0] if model._flat is not None else model._flat_host()
    net.params[...] = flat
    bi = 0
    for m in model.leaves():
        if hasattr(m, "running_mean"):
            rm, rv = net.bn_running(bi)
            rm[...] = m.running_mean
            rv[...] = m.running_var
            bi += 1
    net.layer_index = index
    # per-sample input dims of every max-pool layer (for the tests' tie-conditioning check)
    net.pool_in_dims = {}
    d = tuple(in_dims)
    for m in model.leaves():
        ds, nd_ = m.desc(d)
        if m.typename == "nn.SpatialMaxPooling":
            net.pool_in_dims[index[id(m)]] = d
        d = nd_
    return net


# ---------------------------------------------------------------------------------------------------------------------
# Noise.  The reference draws createNoiseInputs (utils/nn_utils.lua:39-51: torch.Tensor:normal(0, 1) / :uniform(-1, 1)) and the
# nn.Dropout / nn.SpatialDropout masks (nn's bernoulli) from Torch7's CPU MT19937 stream, which the device cannot share; the
# library draws them from a counter-based generator instead (include/ganrev.h: gr_fill_normal_dev, gr_fill_uniform_dev,
# gr_net_set_seed).  Restated here in numpy integer arithmetic so that the DEVICE's noise is checked value by value, not only
# statistically: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
# 1, 2, 3", SC'11 - the Random123 library), pinned below by that library's published known-answer vectors
# (tests/test_oracle_vs_torch.py::test_philox_known_answers), then the same counter / key layout and the same Box-Muller as
# csrc/elem.hip.
_PHILOX_M0, _PHILOX_M1, _PHILOX_W0, _PHILOX_W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds on arrays of counters (uint32 each); returns the four output words as uint32 arrays."""
    c = [np.asarray(v, np.uint64) & np.uint64(0xFFFFFFFF) for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(int(k0) & 0xFFFFFFFF), np.uint64(int(k1) & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(_PHILOX_M0) * c[0]
        p1 = np.uint64(_PHILOX_M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask]
        k0 = (k0 + np.uint64(_PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(_PHILOX_W1)) & mask
    return [v.astype(np.uint32) for v in c]


def _noise_words(n, seed, tag):
    """the four Philox words of every group of four outputs: counter (i lo, i hi, tag, 0), key = the 64-bit seed"""
    i = np.arange((n + 3) // 4, dtype=np.uint64)
    return philox4x32_10(i & np.uint64(0xFFFFFFFF), i >> np.uint64(32), tag, 0, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def fill_uniform(n, seed, lo=-1.0, hi=1.0):
    """gr_fill_uniform_dev: lo + (hi - lo) * (top 24 bits of a Philox word) / 2^24 in fp32, element 4 i + k from word k of counter i"""
    r = _noise_words(n, int(seed), 0x756e6966)
    u = np.stack(r, axis=1).reshape(-1)[:n]
    lo, hi = np.float32(lo), np.float32(hi)
    return lo + (hi - lo) * ((u >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0))


def fill_normal(n, seed):
    """gr_fill_normal_dev: Box-Muller on the words (x, y) and (z, w) of counter i, radius from a uniform in (0, 1], angle from one in
    [0, 1): elements 4i, 4i+1 = r cos / r sin of the first pair, 4i+2, 4i+3 of the second - fp32 throughout (libm's logf / cosf /
    sinf here, the device's there: the comparison allows a few ulp)."""
    x, y, z, w = _noise_words(n, int(seed), 0x6e6f6973)
    f = np.float32
    inv = f(1.0 / 16777216.0)
    u1 = ((x >> np.uint32(8)).astype(f) + f(1)) * inv; u2 = (y >> np.uint32(8)).astype(f) * inv
    u3 = ((z >> np.uint32(8)).astype(f) + f(1)) * inv; u4 = (w >> np.uint32(8)).astype(f) * inv
    ra = np.sqrt(f(-2) * np.log(u1)).astype(f); rb = np.sqrt(f(-2) * np.log(u3)).astype(f)
    tp = f(6.2831853071795864)
    out = np.stack([ra * np.cos(tp * u2).astype(f), ra * np.sin(tp * u2).astype(f), rb * np.cos(tp * u4).astype(f), rb * np.sin(tp * u4).astype(f)], axis=1)
    return out.reshape(-1)[:n].astype(f)


def dropout_keep(n_elems, p_drop, seed, counter, layer):
    """Keep flags (0/1 bytes) of one Dropout / SpatialDropout layer as gr_net draws them: keyed by (seed, forward counter - 1 for the
    first forward after gr_net_set_seed -, layer index of the module, element).  p = 0.5 takes the Philox bits themselves (128 per
    counter); any other p compares one 32-bit word per element with p * 2^32 (keep when word >= threshold)."""
    nwords = (n_elems + 31) // 32
    s0, s1, c0, c1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF
    if np.float32(p_drop) == np.float32(0.5):
        i = np.arange((nwords + 3) // 4, dtype=np.uint64)
        words = np.stack(philox4x32_10(i, layer, c0, c1, s0, s1), axis=1).reshape(-1)[:nwords]
    else:
        thresh = np.uint32(min(4294967295.0, float(np.float32(p_drop)) * 4294967296.0))
        i = np.arange(nwords, dtype=np.uint64)
        words = np.zeros(nwords, np.uint32)
        for j in range(8):
            r = philox4x32_10(i, j | (layer << 8), c0, c1, s0, s1)
            for k in range(4):
                words |= (r[k] >= thresh).astype(np.uint32) << np.uint32(j * 4 + k)
    bits = (words[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & np.uint32(1)
    return bits.reshape(-1)[:n_elems].astype(np.uint8)
