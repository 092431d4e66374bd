"""ctypes binding of libganrev.so (include/ganrev.h) — the only way this package computes anything.

There is NO CPU fallback: if the HIP library is missing, or no gfx950 GPU is visible, every entry point raises
GanrevError.  (The CPU oracle under /oracle is test infrastructure and is never imported from here.)

PyTorch is imported first, when present, only so that this process ends up with ONE HIP runtime / ONE RCCL
(torch ships its own libamdhip64.so.7 / librccl.so.1 with the same sonames as /opt/rocm's): plumbing, not compute.
"""
import ctypes as C
import os

import numpy as np

try:  # noqa: SIM105  (settle which libamdhip64 / librccl the process binds before loading ours)
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional plumbing
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GANREV_LIB") or os.path.join(_HERE, "libganrev.so")   # GANREV_LIB: A/B-testing hook


class GanrevError(RuntimeError):
    pass


GR_OK = 0
STATUS = {0: "GR_OK", -1: "GR_ERR_INVALID", -2: "GR_ERR_UNSUPPORTED", -3: "GR_ERR_HIP", -4: "GR_ERR_NO_DEVICE",
          -5: "GR_ERR_COMM", -6: "GR_ERR_STATE"}

# layer kinds (shared numeric values with the oracle's go_layer)
CONV3, BN, ELU, RELU, LEAKYRELU, SIGMOID, TANH, DROPOUT, SPATIAL_DROPOUT, MAXPOOL2, UPSAMPLE2, VIEW, LINEAR, FULLCONV3 = range(1, 15)
CONVK, PRELU = 15, 16       # the D network's extra module types (models.lua:272-337)
DROPOUT_V2, DROPOUT_ALWAYS_ON = 1, 2
COMM_ID_BYTES = 128


class LayerDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("p", C.c_float), ("flags", C.c_int32)]


class Hyper(C.Structure):
    """optim.adam defaults + train_r.lua:22-24 defaults."""
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("l1", C.c_double), ("l2", C.c_double), ("clamp", C.c_double)]

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, l1=0.0, l2=1e-4, clamp=1.0):
        super().__init__(lr, beta1, beta2, eps, l1, l2, clamp)


_P = C.c_void_p
_F = C.POINTER(C.c_float)
_SIGS = {
    "gr_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gr_shutdown": (C.c_int, [_P]),
    "gr_last_error": (C.c_char_p, [_P]),
    "gr_version": (C.c_char_p, []),
    "gr_stream": (_P, [_P]),
    "gr_synchronize": (C.c_int, [_P]),
    "gr_device_info": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "gr_net_create": (C.c_int, [_P, C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "gr_net_destroy": (C.c_int, [_P]),
    "gr_net_out_dim": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gr_net_param_count": (C.c_int64, [_P]),
    "gr_net_get_params": (C.c_int, [_P, _P]),
    "gr_net_set_params": (C.c_int, [_P, _P]),
    "gr_net_get_grads": (C.c_int, [_P, _P]),
    "gr_net_set_grads": (C.c_int, [_P, _P]),
    "gr_net_zero_grads": (C.c_int, [_P]),
    "gr_net_params_dev": (_P, [_P]),
    "gr_net_grads_dev": (_P, [_P]),
    "gr_net_n_bn": (C.c_int, [_P]),
    "gr_net_bn_features": (C.c_int, [_P, C.c_int]),
    "gr_net_get_bn_running": (C.c_int, [_P, C.c_int, _P, _P]),
    "gr_net_set_bn_running": (C.c_int, [_P, C.c_int, _P, _P]),
    "gr_net_set_training": (C.c_int, [_P, C.c_int]),
    "gr_net_set_seed": (C.c_int, [_P, C.c_uint64]),
    "gr_net_mask_size": (C.c_int64, [_P, C.c_int, C.c_int]),
    "gr_net_set_mask": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gr_net_get_mask": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gr_net_forward_host": (C.c_int, [_P, _P, C.c_int, _P]),
    "gr_net_forward_dev": (C.c_int, [_P, _P, C.c_int, _P]),
    "gr_net_output_dev": (_P, [_P]),
    "gr_net_forward_batched_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "gr_embed_dev": (C.c_int, [_P, C.POINTER(_P), C.c_int, _P, C.c_int64, C.c_int, _P, C.POINTER(_P)]),
    "gr_net_backward_host": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "gr_net_backward_dev": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "gr_net_layer_output": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gr_net_get_pool_index": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "gr_mse_host": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.POINTER(C.c_double), _P]),
    "gr_mse_dev": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _P, _P]),
    "gr_bce_host": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double), _P]),
    "gr_bce_dev": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P]),
    "gr_adam_step": (C.c_int, [_P, C.POINTER(Hyper), C.c_int]),
    "gr_adam_reset": (C.c_int, [_P]),
    "gr_adam_get_state": (C.c_int, [_P, _P, _P]),
    "gr_adam_set_state": (C.c_int, [_P, _P, _P]),
    "gr_comm_unique_id": (C.c_int, [_P, _P]),
    "gr_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "gr_comm_destroy": (C.c_int, [_P]),
    "gr_comm_ranks": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gr_comm_set_host_exchange": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "gr_allreduce_grads": (C.c_int, [_P]),
    "gr_allreduce_dev": (C.c_int, [_P, _P, C.c_int64]),
    "gr_allgather_dev": (C.c_int, [_P, _P, _P, C.c_int64]),
    "gr_broadcast_params": (C.c_int, [_P, C.c_int]),
    "gr_train_r_step": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.POINTER(Hyper), C.c_int, C.POINTER(C.c_double)]),
    "gr_set_conv_mode": (C.c_int, [_P, C.c_int]),
    "gr_get_conv_mode": (C.c_int, [_P]),
    "gr_set_tuning": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "gr_range_guard_stats": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "gr_range_guard_scan_params": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "gr_search_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "gr_debug_stamps": (C.c_int, [_P, _P]),
    "gr_set_timing": (C.c_int, [_P, C.c_int]),
    "gr_last_step_times": (C.c_int, [_P, _P]),
    "gr_event_record": (C.c_int, [_P, C.c_int]),
    "gr_event_elapsed_ms": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "gr_kernel_times": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "gr_cosine_topk_host": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int]),
    "gr_cosine_topk_dev": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int]),
    "gr_cosine_similarity_host": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_float)]),
    "gr_l2_distance_rows_host": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, _P]),
    "gr_kmeans_host": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "gr_cosine_assign_host": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, C.c_int, C.c_int, _P, _P]),
    "gr_malloc": (C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    "gr_free": (C.c_int, [_P, _P]),
    "gr_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_int64]),
    "gr_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_int64]),
    "gr_fill_normal_dev": (C.c_int, [_P, _P, C.c_int64, C.c_uint64]),
    "gr_fill_uniform_dev": (C.c_int, [_P, _P, C.c_int64, C.c_float, C.c_float, C.c_uint64]),
    "gr_copy2d_dev": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64]),
    "gr_add_dev": (C.c_int, [_P, _P, _P, C.c_int64]),
    "gr_conv3_forward_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gr_conv3_backward_data_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gr_conv3_backward_weight_dev": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gr_bench_mfma_loop": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "gr_bench_conv3": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def load_library():
    """dlopen libganrev.so and declare every prototype of include/ganrev.h.  No compute, no GPU needed."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GanrevError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C gan-reverser_amd/csrc`).  This package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError here == header and library disagree
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if isinstance(a, int):
        return C.c_void_p(a)
    return a


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context:
    """One per process per GPU (gr_ctx)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _P()
        rc = self.lib.gr_init(int(device), C.byref(h))
        if rc != GR_OK:
            raise GanrevError(f"gr_init(device={device}) -> {STATUS.get(rc, rc)}: no usable gfx950 GPU "
                              "(libganrev has no CPU path; the oracle under /oracle is for tests only)")
        self.h = h
        self.device = device

    def check(self, rc, what=""):
        if rc != GR_OK:
            msg = self.lib.gr_last_error(self.h)
            raise GanrevError(f"{what}: {STATUS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.gr_shutdown(self.h)
            self.h = None

    def synchronize(self):
        self.check(self.lib.gr_synchronize(self.h), "gr_synchronize")

    def info(self):
        buf = C.create_string_buffer(512)
        self.check(self.lib.gr_device_info(self.h, buf, 512), "gr_device_info")
        return buf.value.decode()

    # ---- raw device memory
    def malloc(self, nbytes):
        p = _P()
        self.check(self.lib.gr_malloc(self.h, int(nbytes), C.byref(p)), "gr_malloc")
        return p.value

    def free(self, p):
        self.check(self.lib.gr_free(self.h, _ptr(p)), "gr_free")

    def upload(self, arr, dptr=None):
        arr = np.ascontiguousarray(arr)
        if dptr is None:
            dptr = self.malloc(arr.nbytes)
        self.check(self.lib.gr_memcpy_h2d(self.h, _ptr(dptr), _ptr(arr), arr.nbytes), "gr_memcpy_h2d")
        return dptr

    def download(self, dptr, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.gr_memcpy_d2h(self.h, _ptr(out), _ptr(dptr), out.nbytes), "gr_memcpy_d2h")
        return out

    def fill_normal(self, dptr, n, seed):
        self.check(self.lib.gr_fill_normal_dev(self.h, _ptr(dptr), int(n), int(seed)), "gr_fill_normal_dev")

    def fill_uniform(self, dptr, n, seed, lo=-1.0, hi=1.0):
        self.check(self.lib.gr_fill_uniform_dev(self.h, _ptr(dptr), int(n), float(lo), float(hi), int(seed)), "gr_fill_uniform_dev")

    def copy2d(self, dst, dst_pitch, src, src_pitch, rows, cols):
        """rows x cols floats between two row-major device matrices (pitches in floats): nn.Concat's join / slice"""
        self.check(self.lib.gr_copy2d_dev(self.h, _ptr(dst), int(dst_pitch), _ptr(src), int(src_pitch), int(rows), int(cols)), "gr_copy2d_dev")

    def add(self, y, x, n):
        self.check(self.lib.gr_add_dev(self.h, _ptr(y), _ptr(x), int(n)), "gr_add_dev")

    def bce_dev(self, x, t, n, loss_dev, grad_dev=None):
        self.check(self.lib.gr_bce_dev(self.h, _ptr(x), _ptr(t), int(n), _ptr(loss_dev), _ptr(grad_dev)), "gr_bce_dev")

    # ---- criterion / search
    def mse(self, x, t, n_global=None, want_grad=True):
        x, t = f32(x), f32(t)
        loss = C.c_double()
        g = np.empty_like(x) if want_grad else None
        self.check(self.lib.gr_mse_host(self.h, _ptr(x), _ptr(t), x.size, int(n_global or x.size), C.byref(loss), _ptr(g)), "gr_mse_host")
        return loss.value, g

    def bce(self, x, t, want_grad=True):
        """nn.BCECriterion (sizeAverage): (loss, gradInput)"""
        x, t = f32(x), f32(t)
        loss = C.c_double()
        g = np.empty_like(x) if want_grad else None
        self.check(self.lib.gr_bce_host(self.h, _ptr(x), _ptr(t), x.size, C.byref(loss), _ptr(g)), "gr_bce_host")
        return loss.value, g

    def cosine_topk(self, emb, query_rows, k, accumulate_in_float=False, emb_dev=None, n=None, d=None):
        q = np.ascontiguousarray(query_rows, dtype=np.int64)
        if emb_dev is None:
            emb = f32(emb)
            n, d = emb.shape
        k = min(int(k), int(n))
        idx = np.empty((q.size, k), dtype=np.int64)
        sc = np.empty((q.size, k), dtype=np.float32)
        if emb_dev is None:
            rc = self.lib.gr_cosine_topk_host(self.h, _ptr(emb), n, d, _ptr(q), q.size, k, _ptr(idx), _ptr(sc), int(accumulate_in_float))
        else:
            rc = self.lib.gr_cosine_topk_dev(self.h, _ptr(emb_dev), n, d, _ptr(q), q.size, k, _ptr(idx), _ptr(sc), int(accumulate_in_float))
        self.check(rc, "gr_cosine_topk")
        return idx, sc

    def cosine_similarity(self, a, b):
        a, b = f32(a).ravel(), f32(b).ravel()
        out = C.c_float()
        self.check(self.lib.gr_cosine_similarity_host(self.h, _ptr(a), _ptr(b), a.size, C.byref(out)), "gr_cosine_similarity_host")
        return out.value

    def l2_distance_rows(self, a, b):
        """torch.dist(a[i], b[i]) for every row i (apply_r.lua:369)."""
        a, b = f32(a), f32(b)
        n = a.shape[0]
        a2, b2 = a.reshape(n, -1), b.reshape(n, -1)
        out = np.empty(n, dtype=np.float64)
        self.check(self.lib.gr_l2_distance_rows_host(self.h, _ptr(a2), _ptr(b2), n, a2.shape[1], _ptr(out)), "gr_l2_distance_rows_host")
        return out

    def kmeans(self, x, k, niter, centroids0):
        """unsup.kmeans(x, k, niter) (apply_r.lua:198) from the given initial centroids -> (centroids, totalcounts, labels)."""
        x = f32(x)
        n, d = x.shape
        cent = np.array(centroids0, dtype=np.float32, order="C", copy=True).reshape(k, d)
        tot = np.zeros(k, np.float32)
        lab = np.zeros(n, np.int32)
        self.check(self.lib.gr_kmeans_host(self.h, _ptr(x), n, d, int(k), int(niter), _ptr(cent), _ptr(tot), _ptr(lab)), "gr_kmeans_host")
        return cent, tot, lab

    def cosine_assign(self, x, centroids, take_min=True):
        """apply_r.lua:205-217: per row the centroid with the minimum (reference behaviour) or maximum cosine similarity."""
        x, cent = f32(x), f32(centroids)
        n, d = x.shape
        lab = np.zeros(n, np.int32)
        sim = np.zeros(n, np.float32)
        self.check(self.lib.gr_cosine_assign_host(self.h, _ptr(x), n, d, _ptr(cent), cent.shape[0], int(take_min), _ptr(lab), _ptr(sim)),
                   "gr_cosine_assign_host")
        return lab, sim

    # ---- data parallel
    def comm_unique_id(self):
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        self.check(self.lib.gr_comm_unique_id(self.h, buf), "gr_comm_unique_id")
        return bytes(buf)

    def comm_init(self, uid, nranks, rank):
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(uid)
        self.check(self.lib.gr_comm_init(self.h, buf, int(nranks), int(rank)), "gr_comm_init")

    def comm_destroy(self):
        self.check(self.lib.gr_comm_destroy(self.h), "gr_comm_destroy")

    EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)

    def set_host_exchange(self, nranks, rank, fn):
        """Install fn(dev_ptr, count, kind) -> 0 as this context's collectives (gr_comm_set_host_exchange: kind 0 fp32 SUM, 1 fp64 SUM,
        2 uint32 MAX over the ranks, result left in the device buffer); fn = None removes it.  The "fake comm" of SURVEY.md section 4."""
        if fn is None:
            self._xchg_cb = None
            self.check(self.lib.gr_comm_set_host_exchange(self.h, 1, 0, None, None), "gr_comm_set_host_exchange")
            return

        def tramp(user, buf, count, kind):
            try:
                return int(fn(buf, int(count), int(kind)) or 0)
            except Exception:  # noqa: BLE001 - an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._xchg_cb = self.EXCHANGE_FN(tramp)           # keep the trampoline alive as long as the library may call it
        self.check(self.lib.gr_comm_set_host_exchange(self.h, int(nranks), int(rank), C.cast(self._xchg_cb, C.c_void_p), None),
                   "gr_comm_set_host_exchange")

    def allreduce(self, dptr, n):
        self.check(self.lib.gr_allreduce_dev(self.h, _ptr(dptr), int(n)), "gr_allreduce_dev")

    def allgather(self, send_dev, recv_dev, nbytes):
        self.check(self.lib.gr_allgather_dev(self.h, _ptr(send_dev), _ptr(recv_dev), int(nbytes)), "gr_allgather_dev")

    def set_conv_mode(self, mode):
        """0 / "f32": exact fp32 MFMA; 1 / "bf16x6": fp32-accurate 3-term bf16 split on the bf16 MFMA (6 products);
        2 / "f16x3": fp32-accurate 2-term fp16 split of power-of-two-scaled operands on the f16 MFMA (3 products)."""
        mode = {"f32": 0, "bf16x6": 1, "f16x3": 2}.get(mode, mode)
        self.check(self.lib.gr_set_conv_mode(self.h, int(mode)), "gr_set_conv_mode")

    def set_tuning(self, key, value):
        self.check(self.lib.gr_set_tuning(self.h, key.encode(), int(value)), "gr_set_tuning")

    def search_reruns(self):
        """searches whose sample-bound filter overflowed and ran again on every key, since gr_init"""
        a = C.c_int64(0)
        self.check(self.lib.gr_search_stats(self.h, C.byref(a)), "gr_search_stats")
        return a.value

    def range_guard_stats(self):
        """(scan launches, passes sent to bf16x6) of the f16x3 range guard since gr_init"""
        a, b = C.c_int64(0), C.c_int64(0)
        self.check(self.lib.gr_range_guard_stats(self.h, C.byref(a), C.byref(b)), "gr_range_guard_stats")
        return a.value, b.value

    def conv_mode(self):
        return ("f32", "bf16x6", "f16x3")[self.lib.gr_get_conv_mode(self.h)]

    def set_timing(self, mode):
        self.check(self.lib.gr_set_timing(self.h, int(mode)), "gr_set_timing")

    def kernel_times(self):
        import json
        buf = C.create_string_buffer(1 << 18)
        self.check(self.lib.gr_kernel_times(self.h, buf, 1 << 18), "gr_kernel_times")
        return json.loads(buf.value.decode())

    def event_record(self, slot):
        self.check(self.lib.gr_event_record(self.h, int(slot)), "gr_event_record")

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self.check(self.lib.gr_event_elapsed_ms(self.h, int(a), int(b), C.byref(ms)), "gr_event_elapsed_ms")
        return ms.value

    def comm_ranks(self):
        n, r = C.c_int(), C.c_int()
        self.check(self.lib.gr_comm_ranks(self.h, C.byref(n), C.byref(r)), "gr_comm_ranks")
        return n.value, r.value

    def last_step_times(self):
        t = np.zeros(6, dtype=np.float32)
        self.check(self.lib.gr_last_step_times(self.h, _ptr(t)), "gr_last_step_times")
        return dict(zip(("g_fwd", "r_fwd", "loss", "r_bwd", "allreduce", "adam"), t.tolist()))

    def bench_mfma_loop(self, shape=0, launches=100):
        """fp32-accurate TFLOP/s the bare f16x3 inner loop sustains on this device (0: 32x32x16, 1: 16x16x32 MFMA shape)"""
        t = C.c_float()
        self.check(self.lib.gr_bench_mfma_loop(self.h, int(shape), int(launches), C.byref(t)), "gr_bench_mfma_loop")
        return t.value

    def bench_conv3(self, which, batch, cin, cout, h, w, iters):
        ms = C.c_float()
        self.check(self.lib.gr_bench_conv3(self.h, which, batch, cin, cout, h, w, iters, C.byref(ms)), "gr_bench_conv3")
        return ms.value


_default_ctx = None


def default_context():
    """Lazily created process-wide context on LOCAL_RANK (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx


class Net:
    """gr_net handle: a compiled nn.Sequential."""

    def __init__(self, ctx, descs, in_dims):
        self.ctx, self.lib = ctx, ctx.lib
        arr = (LayerDesc * len(descs))(*[LayerDesc(*d) for d in descs])
        h = _P()
        c, hh, w = in_dims
        ctx.check(self.lib.gr_net_create(ctx.h, arr, len(descs), int(c), int(hh), int(w), C.byref(h)), "gr_net_create")
        self.h = h
        self.in_dims = tuple(int(v) for v in in_dims)
        oc, oh, ow = C.c_int(), C.c_int(), C.c_int()
        ctx.check(self.lib.gr_net_out_dim(h, C.byref(oc), C.byref(oh), C.byref(ow)), "gr_net_out_dim")
        self.out_dims = (oc.value, oh.value, ow.value)
        self.n_params = int(self.lib.gr_net_param_count(h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.gr_net_destroy(self.h)
            self.h = None

    def _c(self, rc, what):
        self.ctx.check(rc, what)

    def get_params(self):
        a = np.empty(self.n_params, dtype=np.float32)
        self._c(self.lib.gr_net_get_params(self.h, _ptr(a)), "gr_net_get_params")
        return a

    def set_params(self, a):
        a = f32(a)
        assert a.size == self.n_params
        self._c(self.lib.gr_net_set_params(self.h, _ptr(a)), "gr_net_set_params")

    def get_grads(self):
        a = np.empty(self.n_params, dtype=np.float32)
        self._c(self.lib.gr_net_get_grads(self.h, _ptr(a)), "gr_net_get_grads")
        return a

    def set_grads(self, a):
        a = f32(a)
        self._c(self.lib.gr_net_set_grads(self.h, _ptr(a)), "gr_net_set_grads")

    def zero_grads(self):
        self._c(self.lib.gr_net_zero_grads(self.h), "gr_net_zero_grads")

    def n_bn(self):
        return self.lib.gr_net_n_bn(self.h)

    def get_bn_running(self, i):
        n = self.lib.gr_net_bn_features(self.h, i)
        m, v = np.empty(n, np.float32), np.empty(n, np.float32)
        self._c(self.lib.gr_net_get_bn_running(self.h, i, _ptr(m), _ptr(v)), "gr_net_get_bn_running")
        return m, v

    def set_bn_running(self, i, m, v):
        m, v = f32(m), f32(v)
        self._c(self.lib.gr_net_set_bn_running(self.h, i, _ptr(m), _ptr(v)), "gr_net_set_bn_running")

    def set_training(self, t):
        self._c(self.lib.gr_net_set_training(self.h, int(bool(t))), "gr_net_set_training")

    def set_seed(self, s):
        self._c(self.lib.gr_net_set_seed(self.h, int(s)), "gr_net_set_seed")

    def mask_size(self, layer, batch):
        return int(self.lib.gr_net_mask_size(self.h, layer, batch))

    def set_mask(self, layer, keep):
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        self._c(self.lib.gr_net_set_mask(self.h, layer, _ptr(keep), keep.size), "gr_net_set_mask")

    def get_mask(self, layer, n):
        keep = np.empty(n, dtype=np.uint8)
        self._c(self.lib.gr_net_get_mask(self.h, layer, _ptr(keep), n), "gr_net_get_mask")
        return keep

    def forward(self, x, out=None):
        x = f32(x)
        b = x.shape[0]
        if out is None:
            out = np.empty((b,) + self._shape(self.out_dims), dtype=np.float32)
        self._c(self.lib.gr_net_forward_host(self.h, _ptr(x), b, _ptr(out)), "gr_net_forward_host")
        return out

    def forward_dev(self, x_dev, batch, out_dev=None):
        self._c(self.lib.gr_net_forward_dev(self.h, _ptr(x_dev), int(batch), _ptr(out_dev)), "gr_net_forward_dev")
        return self.lib.gr_net_output_dev(self.h)

    def forward_batched_dev(self, x_dev, rows, batch, out_dev):
        """utils/nn_utils.lua:5-33 on device-resident rows: chunks of `batch` rows, each written straight into out_dev"""
        self._c(self.lib.gr_net_forward_batched_dev(self.h, _ptr(x_dev), int(rows), int(batch), _ptr(out_dev)), "gr_net_forward_batched_dev")

    def backward(self, x, gout, want_gin=True):
        x, gout = f32(x), f32(gout)
        b = x.shape[0]
        gin = np.empty_like(x) if want_gin else None
        self._c(self.lib.gr_net_backward_host(self.h, _ptr(x), _ptr(gout), b, _ptr(gin)), "gr_net_backward_host")
        return gin

    def backward_dev(self, x_dev, gout_dev, batch, gin_dev=None):
        self._c(self.lib.gr_net_backward_dev(self.h, _ptr(x_dev), _ptr(gout_dev), int(batch), _ptr(gin_dev)), "gr_net_backward_dev")

    def layer_output(self, layer, shape):
        a = np.empty(shape, dtype=np.float32)
        self._c(self.lib.gr_net_layer_output(self.h, layer, _ptr(a), a.size), "gr_net_layer_output")
        return a

    def pool_index(self, layer, n):
        """nn.SpatialMaxPooling.indices of the last forward: n bytes, 0..3 = (dy, dx) scan position in the window"""
        a = np.empty(n, dtype=np.uint8)
        self._c(self.lib.gr_net_get_pool_index(self.h, int(layer), _ptr(a), a.size), "gr_net_get_pool_index")
        return a

    def adam_step(self, hyper, t):
        self._c(self.lib.gr_adam_step(self.h, C.byref(hyper), int(t)), "gr_adam_step")

    def adam_reset(self):
        self._c(self.lib.gr_adam_reset(self.h), "gr_adam_reset")

    def adam_state(self):
        m, v = np.empty(self.n_params, np.float32), np.empty(self.n_params, np.float32)
        self._c(self.lib.gr_adam_get_state(self.h, _ptr(m), _ptr(v)), "gr_adam_get_state")
        return m, v

    def set_adam_state(self, m, v):
        m, v = f32(m), f32(v)
        self._c(self.lib.gr_adam_set_state(self.h, _ptr(m), _ptr(v)), "gr_adam_set_state")

    def range_guard_scan(self):
        """f16x3 range guard for loops built from the *_dev calls: synchronous scan of this net's weights / BatchNorm scales; True
        when the context's guard has tripped (it then stays on bf16x6)."""
        t = C.c_int(0)
        self._c(self.lib.gr_range_guard_scan_params(self.h, C.byref(t)), "gr_range_guard_scan_params")
        return bool(t.value)

    def allreduce_grads(self):
        self._c(self.lib.gr_allreduce_grads(self.h), "gr_allreduce_grads")

    def broadcast_params(self, root=0):
        self._c(self.lib.gr_broadcast_params(self.h, int(root)), "gr_broadcast_params")

    @staticmethod
    def _shape(d):
        c, h, w = d
        return (c,) if (h == 1 and w == 1) else (c, h, w)


def embed_dev(gnet, rnets, noise_dev, rows, batch, attr_out_devs, images_out_dev=None):
    """apply_r.lua:145-153 on the device: noise -> G -> images -> every net of `rnets` -> attr_out_devs[k]; nothing visits the host"""
    n = len(rnets)
    nets = (_P * max(n, 1))(*[r.h for r in rnets])
    outs = (_P * max(n, 1))(*[_ptr(p) for p in attr_out_devs])
    rc = gnet.lib.gr_embed_dev(gnet.h, nets, n, _ptr(noise_dev), int(rows), int(batch), _ptr(images_out_dev), outs)
    gnet.ctx.check(rc, "gr_embed_dev")


def train_r_step(gnet, rnet, noise_dev, batch, global_batch, hyper, t, want_loss=True):
    loss = C.c_double()
    rc = rnet.lib.gr_train_r_step(gnet.h, rnet.h, _ptr(noise_dev), int(batch), int(global_batch), C.byref(hyper), int(t),
                                  C.byref(loss) if want_loss else None)
    rnet.ctx.check(rc, "gr_train_r_step")
    return loss.value if want_loss else None
