"""ctypes loader for libnd4hip.so (the C ABI of include/nd4hip.h).

There is NO CPU fallback: if the HIP library is missing, fails to load, or no GPU is usable, every
entry point raises. `load()` only dlopens the library (works without a GPU, used by the CPU test
that checks the exported symbols); `handle()` needs a real device.
"""
import ctypes
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnd4hip.so")

c_dp = ctypes.c_void_p   # device or host pointer to double
c_i64 = ctypes.c_int64
c_int = ctypes.c_int

# name -> (restype, argtypes); must list every symbol include/nd4hip.h declares
SIGNATURES = {
    "nd4hip_device_count": (c_int, []),
    "nd4hip_create": (c_int, [ctypes.POINTER(ctypes.c_void_p), c_int]),
    "nd4hip_destroy": (None, [ctypes.c_void_p]),
    "nd4hip_create_multi": (c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_int), c_int]),
    "nd4hip_device_list": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_int), c_int]),
    "nd4hip_partition": (c_int, [c_i64, c_int, c_int, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "nd4hip_set_stream": (c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "nd4hip_reset_stream": (c_int, [ctypes.c_void_p]),
    "nd4hip_synchronize": (c_int, [ctypes.c_void_p]),
    "nd4hip_last_error": (ctypes.c_char_p, []),
    "nd4hip_version": (ctypes.c_char_p, []),
    "nd4hip_malloc": (c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "nd4hip_free": (c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "nd4hip_memcpy_h2d": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "nd4hip_memcpy_d2h": (c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "nd4hip_timer_start": (c_int, [ctypes.c_void_p]),
    "nd4hip_timer_stop": (c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "nd4hip_fill_uniform_dev": (c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, c_i64, c_dp]),
    "nd4hip_dgemm_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dgemm_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dgemm_ex_dev": (c_int, [ctypes.c_void_p, c_int, c_int, c_i64, c_i64, c_i64, ctypes.c_double, c_dp, c_i64,
                                    c_dp, c_i64, ctypes.c_double, c_dp, c_i64]),
    "nd4hip_dgetrf_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp, ctypes.c_void_p]),
    "nd4hip_dgetrf_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp, ctypes.c_void_p]),
    "nd4hip_dgetrs_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_i64, ctypes.c_void_p, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dgetrs_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_i64, ctypes.c_void_p, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dtrsm_batched_dev": (c_int, [ctypes.c_void_p, c_int, c_int, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dtrsm_batched": (c_int, [ctypes.c_void_p, c_int, c_int, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dpotrf_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp]),
    "nd4hip_dpotrf_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp]),
    "nd4hip_dpotrs_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dpotrs_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dldltrf_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp]),
    "nd4hip_dldltrf_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp]),
    "nd4hip_dldltrs_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dldltrs_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dgebrd_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp, c_dp]),
    "nd4hip_dgebrd_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp, c_dp]),
    "nd4hip_dgehrd_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_dgehrd_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_dqrls_batched_dev": (c_int, [ctypes.c_void_p] + [c_i64] * 5 + [c_dp, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dqrls_batched": (c_int, [ctypes.c_void_p] + [c_i64] * 5 + [c_dp, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dsvdls_batched_dev": (c_int, [ctypes.c_void_p] + [c_i64] * 5 + [c_dp, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dsvdls_batched": (c_int, [ctypes.c_void_p] + [c_i64] * 5 + [c_dp, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "nd4hip_dgeqrf_q_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_dgeqrf_q_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_dgeqrf_full_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_dgeqrf_full_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_dgeqrf_qty_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_i64, c_dp, c_dp]),
    "nd4hip_dgeqrf_qty_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_i64, c_dp, c_dp]),
    "nd4hip_dgesvdj_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp, c_dp,
                                           ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)]),
    "nd4hip_dgesvdj_batched": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp, c_dp,
                                       ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_double)]),
    "nd4hip_dgeqr2_panel_batched_dev": (c_int, [ctypes.c_void_p, c_i64, c_i64, c_i64, c_dp, c_dp, c_dp]),
    "nd4hip_profile_enable": (c_int, [ctypes.c_void_p, c_int]),
    "nd4hip_profile_last": (c_int, [ctypes.c_void_p, ctypes.c_void_p, c_int]),
    "nd4hip_dgesvdj_last_info": (c_int, [ctypes.c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_ulonglong),
                                         ctypes.POINTER(ctypes.c_double)]),
}


class Prof(ctypes.Structure):
    """nd4hip_prof of include/nd4hip.h"""
    _fields_ = [("kernel_ms", ctypes.c_double), ("flops", ctypes.c_double), ("bytes", ctypes.c_double),
                ("device", c_int), ("valid", c_int), ("op", ctypes.c_char * 32)]


class Nd4HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nd4hip error %d: %s" % (code, msg))
        self.code = code


_lib = None
_lock = threading.Lock()


def load():
    """dlopen libnd4hip.so and bind every symbol. Raises if the library is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError("%s not built: run `python -m nd4js_amd.build` (hipcc, gfx950). "
                                  "There is no CPU fallback." % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)      # AttributeError if the ABI lost a symbol
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise Nd4HipError(rc, load().nd4hip_last_error().decode("utf-8", "replace"))


class Handle:
    """One GPU + stream + workspace (include/nd4hip.h). device=None -> current HIP device."""

    def __init__(self, device=None):
        self.lib = load()
        self._h = ctypes.c_void_p()
        if isinstance(device, (list, tuple)):       # one handle over several devices: host-pointer batches are sharded
            ids = (c_int * len(device))(*[int(d) for d in device])
            check(self.lib.nd4hip_create_multi(ctypes.byref(self._h), ids, len(device)))
        else:
            check(self.lib.nd4hip_create(ctypes.byref(self._h), -1 if device is None else int(device)))

    def devices(self):
        ids = (c_int * 64)()
        n = self.lib.nd4hip_device_list(self._h, ids, 64)
        return [ids[i] for i in range(n)]

    @property
    def ptr(self):
        return self._h

    def set_stream(self, stream_ptr):
        check(self.lib.nd4hip_set_stream(self._h, ctypes.c_void_p(stream_ptr or 0)))

    def reset_stream(self):
        check(self.lib.nd4hip_reset_stream(self._h))

    def synchronize(self):
        check(self.lib.nd4hip_synchronize(self._h))

    def timer_start(self):
        check(self.lib.nd4hip_timer_start(self._h))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(self.lib.nd4hip_timer_stop(self._h, ctypes.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        check(self.lib.nd4hip_profile_enable(self._h, 1 if on else 0))

    def profile_last(self):
        """one record per device of the handle: kernel ms, algorithmic flops / bytes of the last call (nd4hip_profile_last)"""
        recs = (Prof * 64)()
        n = self.lib.nd4hip_profile_last(self._h, ctypes.cast(recs, ctypes.c_void_p), 64)
        if n < 0:
            check(n)
        return [{"device": r.device, "valid": bool(r.valid), "op": r.op.decode(), "kernel_ms": r.kernel_ms, "flops": r.flops,
                 "bytes": r.bytes} for r in recs[:n]]

    def svd_last_info(self):
        """sweeps, rotations applied and off-norm of the last svd_decomp on this handle (executed-work audit)"""
        sw, rot, off = c_int(0), ctypes.c_ulonglong(0), ctypes.c_double(0.0)
        check(self.lib.nd4hip_dgesvdj_last_info(self._h, ctypes.byref(sw), ctypes.byref(rot), ctypes.byref(off)))
        return {"sweeps": sw.value, "rotations": rot.value, "offnorm": off.value}

    def close(self):
        if self._h:
            self.lib.nd4hip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = {}


def handle(device=None):
    """Lazily created per-device default handle (the reference has no init step: SURVEY.md §3.5)."""
    if isinstance(device, Handle):
        return device
    key = -1 if device is None else (tuple(device) if isinstance(device, (list, tuple)) else int(device))
    with _lock:
        h = _default.get(key)
    if h is None:
        h = Handle(device)
        with _lock:
            _default[key] = h
    return h


def partition(batch, n_dev, index):
    """[lo, hi) of the batch that device number `index` of an n_dev-device handle processes (include/nd4hip.h)"""
    lo, hi = c_i64(0), c_i64(0)
    check(load().nd4hip_partition(batch, n_dev, index, ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value
