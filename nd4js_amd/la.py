"""Python host mirror of the reference's `nd.la` hot-path functions (same names, argument meaning and
error text) on top of the C ABI — used by the parity tests and bench.py. The production host is the
JS wrapper nd4js_amd/js/index.js over the N-API shim; both call the same libnd4hip.so entry points.

  matmul2(a, b)        src/la/matmul.js:91-147      -> nd4hip_dgemm_batched
  matmul(*ms)          src/la/matmul.js:150-236     (chain ordering stays on the host)
  qr_decomp(A)         src/la/qr.js:80-145          -> nd4hip_dgeqrf_q_batched
  lu_decomp(A)         src/la/lu.js:24-81           -> nd4hip_dgetrf_batched
  svd_decomp(A)        src/la/svd.js:25             -> nd4hip_dgesvdj_batched

Inputs are numpy arrays / nested lists (NDArray analogue: dense, row-major, leading axes = batch).
Only float64 (and int32 promoted to float64 for QR/LU/SVD, as qr.js:31-37, lu.js:27, svd_dc.js:904
do) runs here; other dtypes raise TypeError — there is NO CPU fallback in this package.
"""
import ctypes

import numpy as np

from . import _lib


def _asarray(a, what):
    a = np.asarray(a)
    if a.dtype == np.int32 or a.dtype == np.int64 or a.dtype == np.bool_:
        a = a.astype(np.float64)
    if a.dtype != np.float64:
        raise TypeError("%s: only float64 (or int32 promoted to float64) runs on the GPU path, got %s" % (what, a.dtype))
    return np.ascontiguousarray(a)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _bcast_groups(lead, la, lb, IK, KJ):
    """Flatten NumPy-style broadcasting of the leading axes into (count, offA, sA, offB, sB, offC)
    groups with batch strides in {0, dense} — the job of the odometer at matmul.js:44-70."""
    nb = len(lead)
    la = (1,) * (nb - len(la)) + tuple(la)
    lb = (1,) * (nb - len(lb)) + tuple(lb)
    total = int(np.prod(lead, dtype=np.int64)) if nb else 1
    offA = np.broadcast_to((np.arange(int(np.prod(la, dtype=np.int64)), dtype=np.int64) * IK).reshape(la), lead).reshape(-1)
    offB = np.broadcast_to((np.arange(int(np.prod(lb, dtype=np.int64)), dtype=np.int64) * KJ).reshape(lb), lead).reshape(-1)
    groups, b0 = [], 0
    while b0 < total:
        b1 = b0 + 1
        if b1 < total:
            sA, sB = int(offA[b1] - offA[b0]), int(offB[b1] - offB[b0])
            if sA in (0, IK) and sB in (0, KJ):
                while b1 < total and offA[b1] - offA[b1 - 1] == sA and offB[b1] - offB[b1 - 1] == sB:
                    b1 += 1
            else:
                sA = sB = 0
        else:
            sA = sB = 0
        groups.append((b1 - b0, int(offA[b0]), sA, int(offB[b0]), sB, b0))
        b0 = b1
    return groups


def matmul2(a, b, device=None):
    a = _asarray(a, "matmul2(a,b)")
    b = _asarray(b, "matmul2(a,b)")
    if a.ndim < 2:
        raise ValueError("A must be at least 2D.")
    if b.ndim < 2:
        raise ValueError("B must be at least 2D.")
    I, K = a.shape[-2:]
    J = b.shape[-1]
    if b.shape[-2] != K:
        raise ValueError("The last dimension of A and the 2nd to last dimension of B do not match.")
    try:
        lead = np.broadcast_shapes(a.shape[:-2], b.shape[:-2])
    except ValueError:
        raise ValueError("Shapes are not broadcast-compatible.")
    c = np.empty(tuple(lead) + (I, J), dtype=np.float64)
    h = _lib.handle(device)
    for cnt, offA, sA, offB, sB, offC in _bcast_groups(tuple(lead), a.shape[:-2], b.shape[:-2], I * K, K * J):
        _lib.check(h.lib.nd4hip_dgemm_batched(
            h.ptr, cnt, I, K, J,
            ctypes.c_void_p(a.ctypes.data + 8 * offA), sA,
            ctypes.c_void_p(b.ctypes.data + 8 * offB), sB,
            ctypes.c_void_p(c.ctypes.data + 8 * offC * I * J)))
    return c


def _n_ops(sa, sb):
    I, K = sa[-2:]
    J = sb[-1]
    if sb[-2] != K:
        raise ValueError("Shape mismatch.")
    try:
        lead = np.broadcast_shapes(tuple(sa[:-2]), tuple(sb[:-2]))
    except ValueError:
        raise ValueError("Shapes are not broadcast-compatible.")
    shape = tuple(lead) + (I, J)
    return int(np.prod(shape, dtype=np.int64)) * K, shape


def matmul(*matrices, device=None):
    """Matrix-chain product in the FLOP-optimal order (matmul.js:150-236)."""
    ms = [_asarray(m, "matmul(...)") for m in matrices]
    if len(ms) == 1:
        return ms[0]
    if len(ms) == 2:
        return matmul2(ms[0], ms[1], device)
    n = len(ms)
    op = [[None] * n for _ in range(n)]
    for i in range(n):
        op[i][i] = (0, ms[i].shape)
    for length in range(2, n + 1):
        for i in range(0, n - length + 1):
            best = None
            for j in range(1, length):
                lf, ls = op[i][i + j - 1]
                rf, rs = op[i + j][i + length - 1]
                f, shp = _n_ops(ls, rs)
                f += lf + rf
                if best is None or f < best[0]:
                    best = (f, shp)
            op[i][i + length - 1] = best

    def product(lo, hi):
        if lo == hi:
            return ms[lo]
        best, idx = None, None
        for i in range(lo, hi):
            f = _n_ops(op[lo][i][1], op[i + 1][hi][1])[0] + op[lo][i][0] + op[i + 1][hi][0]
            if best is None or f < best:
                best, idx = f, i
        return matmul2(product(lo, idx), product(idx + 1, hi), device)
    return product(0, n - 1)


def qr_decomp(A, device=None):
    A = _asarray(A, "qr_decomp(A)")
    if A.ndim < 2:
        raise ValueError("qr_decomp(A): A.ndim must be at least 2.")
    M, N = A.shape[-2:]
    L = min(M, N)
    batch = int(np.prod(A.shape[:-2], dtype=np.int64))
    Q = np.empty(A.shape[:-2] + (M, L))
    R = np.empty(A.shape[:-2] + (L, N))
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgeqrf_q_batched(h.ptr, batch, M, N, _ptr(A), _ptr(Q), _ptr(R)))
    return Q, R


def lu_decomp(A, device=None):
    A = _asarray(A, "lu_decomp(A)")
    if A.ndim < 2 or A.shape[-1] != A.shape[-2]:
        raise ValueError("Last two dimensions must be quadratic.")
    N = A.shape[-1]
    batch = int(np.prod(A.shape[:-2], dtype=np.int64))
    LU = np.empty_like(A)
    P = np.empty(A.shape[:-1], dtype=np.int32)
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgetrf_batched(h.ptr, batch, N, _ptr(A), _ptr(LU), _ptr(P)))
    return LU, P


def svd_decomp(A, device=None, info=None):
    A = np.asarray(A)
    if np.iscomplexobj(A):
        raise TypeError("svd_dc(A): A.dtype must be float.")
    A = _asarray(A, "svd_decomp(A)")
    if A.ndim < 2:
        raise ValueError("svd_decomp(A): A.ndim must be at least 2.")
    M, N = A.shape[-2:]
    L = min(M, N)
    batch = int(np.prod(A.shape[:-2], dtype=np.int64))
    U = np.empty(A.shape[:-2] + (M, L))
    sv = np.empty(A.shape[:-2] + (L,))
    V = np.empty(A.shape[:-2] + (L, N))
    sweeps, off = ctypes.c_int(0), ctypes.c_double(0.0)
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgesvdj_batched(h.ptr, batch, M, N, _ptr(A), _ptr(U), _ptr(sv), _ptr(V),
                                            ctypes.byref(sweeps), ctypes.byref(off)))
    if info is not None:
        info["sweeps"], info["offnorm"] = sweeps.value, off.value
    return U, sv, V


svd_dc = svd_decomp
