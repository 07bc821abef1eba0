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


def matmul2(a, b, device=None, out=None):
    """`out`: optional preallocated C-contiguous float64 result (a fresh 128 MiB array costs 8-15 ms of page faults on its first
    write — the OS's price, tools/pcie_fresh.hip — which a caller that reuses buffers does not pay)."""
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
    c = np.empty(tuple(lead) + (I, J), dtype=np.float64) if out is None else out
    if c.shape != tuple(lead) + (I, J) or c.dtype != np.float64 or not c.flags.c_contiguous:
        raise ValueError("matmul2(a,b,out): out must be a C-contiguous float64 array of shape %r" % (tuple(lead) + (I, J),))
    h = _lib.handle(device)
    for cnt, offA, sA, offB, sB, offC in _bcast_groups(tuple(lead), a.shape[:-2], b.shape[:-2], I * K, K * J):
        _lib.check(h.lib.nd4hip_dgemm_batched(
            h.ptr, cnt, I, K, J,
            ctypes.c_void_p(a.ctypes.data + 8 * offA), sA,
            ctypes.c_void_p(b.ctypes.data + 8 * offB), sB,
            ctypes.c_void_p(c.ctypes.data + 8 * offC * I * J)))
    return c


def _product_shape(sa, sb):
    """Shape of matmul2 of operands shaped `sa`, `sb` (leading axes NumPy-broadcast) and its inner extent K."""
    if sb[-2] != sa[-1]:
        raise ValueError("Shape mismatch.")
    la_, lb_ = tuple(sa[:-2]), tuple(sb[:-2])
    rank = max(len(la_), len(lb_))
    la_, lb_ = (1,) * (rank - len(la_)) + la_, (1,) * (rank - len(lb_)) + lb_
    if any(x != y and x != 1 and y != 1 for x, y in zip(la_, lb_)):
        raise ValueError("Shapes are not broadcast-compatible.")
    return tuple(y if x == 1 else x for x, y in zip(la_, lb_)) + (sa[-2], sb[-1]), sa[-1]


def chain_plan(shapes):
    """Matrix-chain ordering (CLRS 15.2) generalised to broadcast leading axes: the cost of one product is
    numel(result)·K multiply-adds, as matmul.js:150-236 counts it. Returns `cut` with cut[lo][hi] = last operand
    of the left factor of the cheapest split of operands lo..hi (the first one among equals, in double
    arithmetic like the reference's Number). The shape of a sub-chain does not depend on how it is split, so one
    shape per span is enough."""
    n = len(shapes)
    shape = [[None] * n for _ in range(n)]
    cost = [[0.0] * n for _ in range(n)]
    cut = [[-1] * n for _ in range(n)]
    for k in range(n):
        shape[k][k] = tuple(shapes[k])
    for width in range(1, n):
        for lo in range(n - width):
            hi = lo + width
            best = float("inf")
            for mid in range(lo, hi):
                shp, K = _product_shape(shape[lo][mid], shape[mid + 1][hi])
                numel = 1.0
                for e in shp:
                    numel *= e
                c = numel * K + (cost[lo][mid] + cost[mid + 1][hi])
                if c < best:
                    best, cut[lo][hi], shape[lo][hi] = c, mid, shp
            if cut[lo][hi] < 0:
                raise ValueError("Integer overflow (too many FLOPs).")
            cost[lo][hi] = best
    return cut


def matmul(*matrices, device=None, _matmul2=None):
    """Product of a chain of matrices in the FLOP-optimal order (contract of matmul.js:150-236): one operand
    is returned as is, two go straight to matmul2, longer chains are parenthesised by `chain_plan`."""
    ms = [_asarray(m, "matmul(...)") for m in matrices]
    mm = _matmul2 or (lambda x, y: matmul2(x, y, device))
    if len(ms) == 1:
        return ms[0]
    if len(ms) == 2:
        return mm(ms[0], ms[1])
    cut = chain_plan([m.shape for m in ms])

    def evaluate(lo, hi):
        if lo == hi:
            return ms[lo]
        return mm(evaluate(lo, cut[lo][hi]), evaluate(cut[lo][hi] + 1, hi))
    return evaluate(0, len(ms) - 1)


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


def qr_decomp_full(A, device=None):
    """qr.js:27-77: Q [..., M, M], R [..., M, N]."""
    A = _asarray(A, "qr_decomp_full(A)")
    if A.ndim < 2:
        raise ValueError("A must be at least 2D.")
    M, N = A.shape[-2:]
    batch = int(np.prod(A.shape[:-2], dtype=np.int64))
    Q = np.empty(A.shape[:-2] + (M, M))
    R = np.empty(A.shape[:-2] + (M, N))
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgeqrf_full_batched(h.ptr, batch, M, N, _ptr(A), _ptr(Q), _ptr(R)))
    return Q, R


def qr_decomp_inplace(A, Y, device=None):
    """_qr_decomp_inplace (qr.js:146-183) on writable float64 arrays A [..., M, N] and Y [..., M, L] with equal leading
    dims: A <- R, Y <- Q^T Y, in place like the reference. Returns (A, Y)."""
    for a, n in ((A, "A"), (Y, "Y")):
        if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable):
            raise TypeError("qr_decomp_inplace: %s must be a writable C-contiguous float64 ndarray" % n)
    if A.ndim < 2 or Y.ndim < 2 or A.shape[:-1] != Y.shape[:-1]:
        raise ValueError("Assertion failed.")                      # the reference's only message (qr.js:148-166)
    M, N = A.shape[-2:]
    L = Y.shape[-1]
    batch = int(np.prod(A.shape[:-2], dtype=np.int64))
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgeqrf_qty_batched(h.ptr, batch, M, N, L, _ptr(A), _ptr(Y)))
    return A, Y


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
        info["rotations"] = h.svd_last_info()["rotations"]
    return U, sv, V


svd_dc = svd_decomp


# ---------------------------------------------------------------------------------------------------
# solve-side consumers of the path, device kernels in csrc/trsm.hip
# ---------------------------------------------------------------------------------------------------
def _bcast_groups_n(lead, shapes, units):
    """n-operand form of _bcast_groups: yields (count, [offset_k], [stride_k], out_index) with every stride in
    {0, units[k]} (what the odometers at lu.js:148-163 / tri.js:193-215 do one matrix at a time)."""
    nb = len(lead)
    total = int(np.prod(lead, dtype=np.int64)) if nb else 1
    offs = []
    for shp, unit in zip(shapes, units):
        shp = (1,) * (nb - len(shp)) + tuple(shp)
        offs.append(np.broadcast_to((np.arange(int(np.prod(shp, dtype=np.int64)), dtype=np.int64) * unit).reshape(shp), lead).reshape(-1))
    groups, b0 = [], 0
    while b0 < total:
        b1, strides = b0 + 1, [0] * len(offs)
        if b1 < total:
            cand = [int(o[b1] - o[b0]) for o in offs]
            if all(c in (0, u) for c, u in zip(cand, units)):
                strides = cand
                while b1 < total and all(int(o[b1] - o[b1 - 1]) == c for o, c in zip(offs, cand)):
                    b1 += 1
        groups.append((b1 - b0, [int(o[b0]) for o in offs], strides, b0))
        b0 = b1
    return groups


def _off(a, elems, itemsize=8):
    return ctypes.c_void_p(a.ctypes.data + itemsize * elems)


def _tri_solve(upper, T, Y, name, device=None):
    T = _asarray(T, name)
    Y = _asarray(Y, name)
    if T.ndim < 2:
        raise ValueError("%s: %s.ndim must be at least 2." % (name, "U" if upper else "L"))
    if Y.ndim < 2:
        raise ValueError("%s: Y.ndim must be at least 2." % name)
    M, J = Y.shape[-2:]
    if T.shape[-2] != M:
        raise ValueError("%s: %s and Y don't match." % (name, "U" if upper else "L"))
    if T.shape[-1] != M:
        raise ValueError("%s: Last two dimensions of %s must be quadratic." % (name, "U" if upper else "L"))
    try:
        lead = np.broadcast_shapes(T.shape[:-2], Y.shape[:-2])
    except ValueError:
        raise ValueError("%s: %s and Y not broadcast-compatible." % (name, "U" if upper else "L"))
    X = np.empty(tuple(lead) + (M, J))
    h = _lib.handle(device)
    for cnt, (oT, oY), (sT, sY), b0 in _bcast_groups_n(tuple(lead), [T.shape[:-2], Y.shape[:-2]], [M * M, M * J]):
        _lib.check(h.lib.nd4hip_dtrsm_batched(h.ptr, 1 if upper else 0, 0, cnt, M, J, _off(T, oT), sT, _off(Y, oY), sY, _off(X, b0 * M * J)))
    return X


def tril_solve(L, Y, device=None):
    """tri.js:155-221"""
    return _tri_solve(False, L, Y, "tril_solve(L,Y)", device)


def triu_solve(U, Y, device=None):
    """tri.js:224-290"""
    return _tri_solve(True, U, Y, "triu_solve(U,Y)", device)


def lu_solve(LU, P, y=None, device=None):
    """lu.js:84-177; accepts lu_solve((LU,P), y) like the reference (:86)."""
    if y is None:
        y = P
        LU, P = LU
    LU = _asarray(LU, "lu_solve")
    P = np.ascontiguousarray(np.asarray(P), dtype=np.int32)
    y = _asarray(y, "lu_solve")
    if LU.ndim < 2:
        raise ValueError("LU must be at least 2D.")
    if P.ndim < 1:
        raise ValueError("P must be at least 1D.")
    if y.ndim < 2:
        raise ValueError("y must be at least 2D.")
    N = LU.shape[-2]
    I, J = y.shape[-2:]
    if LU.shape[-1] != N:
        raise ValueError("Last two dimensions of LU must be quadratic.")
    if N != I:
        raise ValueError("LU and y don't match.")
    if P.shape[-1] != N:
        raise ValueError("LU and P don't match.")
    try:
        lead = np.broadcast_shapes(LU.shape[:-2], y.shape[:-2])
    except ValueError:
        raise ValueError("LU and y are not broadcast-compatible.")
    try:
        lead = np.broadcast_shapes(lead, P.shape[:-1])
    except ValueError:
        raise ValueError("P is not broadcast-compatible.")
    X = np.empty(tuple(lead) + (N, J))
    h = _lib.handle(device)
    for cnt, (oLU, oP, oY), (sLU, sP, sY), b0 in _bcast_groups_n(tuple(lead), [LU.shape[:-2], P.shape[:-1], y.shape[:-2]], [N * N, N, N * J]):
        _lib.check(h.lib.nd4hip_dgetrs_batched(h.ptr, cnt, N, J, _off(LU, oLU), sLU, _off(P, oP, 4), sP, _off(y, oY), sY, _off(X, b0 * N * J)))
    return X


def qr_lstsq(Q, R, y=None, device=None):
    """qr.js:186-273; accepts qr_lstsq((Q,R), y) like the reference (:188)."""
    if y is None:
        y = R
        Q, R = Q
    Q, R, y = np.asarray(Q), np.asarray(R), np.asarray(y)
    if Q.ndim < 2:
        raise ValueError("qr_lstsq(Q,R,y): Q.ndim must be at least 2.")
    if R.ndim < 2:
        raise ValueError("qr_lstsq(Q,R,y): R.ndim must be at least 2.")
    if y.ndim < 2:
        raise ValueError("qr_lstsq(Q,R,y): y.ndim must be at least 2.")
    Q, R, y = _asarray(Q, "qr_lstsq"), _asarray(R, "qr_lstsq"), _asarray(y, "qr_lstsq")
    N, M = Q.shape[-2:]
    I, J = R.shape[-1], y.shape[-1]
    if N != y.shape[-2]:
        raise ValueError("qr_lstsq(Q,R,y): Q and y don't match.")
    if M != R.shape[-2]:
        raise ValueError("qr_lstsq(Q,R,y): Q and R don't match.")
    if I > N:
        raise ValueError("qr_lstsq(Q,R,y): Under-determined systems not supported. Use rrqr instead.")
    try:
        lead = np.broadcast_shapes(Q.shape[:-2], R.shape[:-2], y.shape[:-2])
    except ValueError:
        raise ValueError("Q, R, y are not broadcast-compatible.")
    X = np.empty(tuple(lead) + (I, J))
    h = _lib.handle(device)
    for cnt, (oQ, oR, oY), (sQ, sR, sY), b0 in _bcast_groups_n(tuple(lead), [Q.shape[:-2], R.shape[:-2], y.shape[:-2]], [N * M, M * I, N * J]):
        _lib.check(h.lib.nd4hip_dqrls_batched(h.ptr, cnt, N, M, I, J, _off(Q, oQ), sQ, _off(R, oR), sR, _off(y, oY), sY, _off(X, b0 * I * J)))
    return X


def svd_rank(sv):
    """svd.js:31-63: per matrix, the first r with |sv_r| <= sqrt(eps) |sv_0| (host side: sv is tiny)."""
    sv = np.asarray(sv, dtype=np.float64)
    N = sv.shape[-1]
    with np.errstate(invalid="ignore"):
        small = np.abs(sv) <= np.sqrt(2.0 ** -52) * np.abs(sv[..., :1])
    rank = np.where(small.any(axis=-1), small.argmax(axis=-1), N).astype(np.int32)
    # the reference's loop only looks at the entries up to the cut (and at the cut itself): a NaN behind it does not raise
    seen = np.arange(N) <= rank[..., None]
    if not np.all(np.isfinite(sv[seen])):
        raise ValueError("svd_rank(): NaN or Infinity encountered.")
    return rank


def svd_lstsq(U, sv, V=None, y=None, device=None):
    """svd.js:100-228; accepts svd_lstsq((U,sv,V), y) like the reference (:102-108)."""
    if y is None:
        if V is not None:
            raise ValueError("svd_lstsq(Q,R,P, y): Either 2 ([Q,R,P], y) or 4 arguments (Q,R,P, y) expected.")
        y = sv
        U, sv, V = U
    U, sv, V, y = np.asarray(U), np.asarray(sv), np.asarray(V), np.asarray(y)
    if U.ndim < 2:
        raise ValueError("svd_lstsq(U,sv,V, y): U.ndim must be at least 2.")
    if sv.ndim < 1:
        raise ValueError("svd_lstsq(U,sv,V, y): sv.ndim must be at least 1.")
    if V.ndim < 2:
        raise ValueError("svd_lstsq(U,sv,V, y): V.ndim must be at least 2.")
    if y.ndim < 2:
        raise ValueError("svd_lstsq(U,sv,V, y): y.ndim must be at least 2.")
    U, sv, V, y = (_asarray(a, "svd_lstsq") for a in (U, sv, V, y))
    N, M = U.shape[-2:]
    I, J = V.shape[-1], y.shape[-1]
    if N != y.shape[-2]:
        raise ValueError("svd_lstsq(U,sv,V, y): U and y don't match.")
    if M != sv.shape[-1]:
        raise ValueError("svd_lstsq(U,sv,V, y): U and sv don't match.")
    if M != V.shape[-2]:
        raise ValueError("svd_lstsq(U,sv,V, y): V and sv don't match.")
    try:
        lead = np.broadcast_shapes(U.shape[:-2], V.shape[:-2], y.shape[:-2], sv.shape[:-1])
    except ValueError:
        raise ValueError("svd_lstsq(U,sv,V, y): U,sv,V,y not broadcast-compatible.")
    if not np.all(np.isfinite(sv)):
        raise ValueError("svd_solve(): NaN or Infinity encountered.")      # svd.js:171-172 (the reference's own wording)
    X = np.empty(tuple(lead) + (I, J))
    h = _lib.handle(device)
    for cnt, (oU, oS, oV, oY), (sU, sS, sV, sY), b0 in _bcast_groups_n(
            tuple(lead), [U.shape[:-2], sv.shape[:-1], V.shape[:-2], y.shape[:-2]], [N * M, M, M * I, N * J]):
        _lib.check(h.lib.nd4hip_dsvdls_batched(h.ptr, cnt, N, M, I, J, _off(U, oU), sU, _off(sv, oS), sS, _off(V, oV), sV,
                                               _off(y, oY), sY, _off(X, b0 * I * J)))
    return X


def svd_solve(U, sv, V=None, y=None, device=None):
    """svd.js:66-97. The reference's singularity loop (`for( let r; r < N; r++ )`, :85) never executes, so apart from
    the squareness check this IS svd_lstsq; mirrored as such."""
    if y is None:
        if V is not None:
            raise ValueError("svd_lstsq(Q,R,P, y): Either 2 ([Q,R,P], y) or 4 arguments (Q,R,P, y) expected.")
        y = sv
        U, sv, V = U
    if np.asarray(U).shape[-2] != np.asarray(V).shape[-1]:
        raise ValueError("rrqr_solve(Q,R,P, y): System not square.")
    return svd_lstsq(U, sv, V, y, device=device)


def cholesky_decomp(S, device=None):
    """cholesky.js:51-71: L with S = L L^T, strict upper part zero; only the lower triangle of S is read."""
    S = _asarray(S, "cholesky_decomp(S)")
    if S.ndim < 2 or S.shape[-1] != S.shape[-2]:
        raise ValueError("Last two dimensions must be quadratic.")
    N = S.shape[-1]
    L = np.empty_like(S)
    h = _lib.handle(device)
    try:
        _lib.check(h.lib.nd4hip_dpotrf_batched(h.ptr, int(np.prod(S.shape[:-2], dtype=np.int64)), N, _ptr(S), _ptr(L)))
    except _lib.Nd4HipError as e:
        if e.code == -5:
            # cholesky.js:43-44; a multi-device handle adds which of its devices reported it
            where = str(e)[str(e).find(" (device "):] if " (device " in str(e) else ""
            raise ValueError("Matrix contains NaNs or is (near) singular." + where)
        raise
    return L


def cholesky_solve(L, y, device=None):
    """cholesky.js:74-150"""
    L, y = np.asarray(L), np.asarray(y)
    if L.ndim < 2:
        raise ValueError("L must be at least 2D.")
    if y.ndim < 2:
        raise ValueError("y must be at least 2D.")
    L, y = _asarray(L, "cholesky_solve"), _asarray(y, "cholesky_solve")
    N, M = L.shape[-2:]
    I, J = y.shape[-2:]
    if N != M:
        raise ValueError("Last two dimensions of L must be quadratic.")
    if I != M:
        raise ValueError("L and y don't match.")
    try:
        lead = np.broadcast_shapes(L.shape[:-2], y.shape[:-2])
    except ValueError:
        raise ValueError("Shapes are not broadcast-compatible.")
    X = np.empty(tuple(lead) + (N, J))
    h = _lib.handle(device)
    for cnt, (oL, oY), (sL, sY), b0 in _bcast_groups_n(tuple(lead), [L.shape[:-2], y.shape[:-2]], [N * N, N * J]):
        _lib.check(h.lib.nd4hip_dpotrs_batched(h.ptr, cnt, N, J, _off(L, oL), sL, _off(y, oY), sY, _off(X, b0 * N * J)))
    return X


def ldl_decomp(S, device=None):
    """ldl.js:67-90: packed LD (unit-lower L below the diagonal, D on it, zeros above), S = L D L^T, no pivoting."""
    S = _asarray(S, "ldl_decomp(S)")
    if S.ndim < 2 or S.shape[-1] != S.shape[-2]:
        raise ValueError("Last two dimensions must be quadratic.")
    LD = np.empty_like(S)
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dldltrf_batched(h.ptr, int(np.prod(S.shape[:-2], dtype=np.int64)), S.shape[-1], _ptr(S), _ptr(LD)))
    return LD


def ldl_solve(LD, y, device=None):
    """ldl.js:133-201"""
    LD, y = np.asarray(LD), np.asarray(y)
    if LD.ndim < 2:
        raise ValueError("ldl_solve(LD,y): LD must be at least 2D.")
    if y.ndim < 2:
        raise ValueError("ldl_solve(LD,y): y must be at least 2D.")
    LD, y = _asarray(LD, "ldl_solve"), _asarray(y, "ldl_solve")
    N, M = LD.shape[-2:]
    I, J = y.shape[-2:]
    if N != M:
        raise ValueError("ldl_solve(LD,y): Last two dimensions of LD must be quadratic.")
    if I != M:
        raise ValueError("ldl_solve(LD,y): LD and y don't match.")
    try:
        lead = np.broadcast_shapes(LD.shape[:-2], y.shape[:-2])
    except ValueError:
        raise ValueError("Shapes are not broadcast-compatible.")
    X = np.empty(tuple(lead) + (N, J))
    h = _lib.handle(device)
    for cnt, (oL, oY), (sL, sY), b0 in _bcast_groups_n(tuple(lead), [LD.shape[:-2], y.shape[:-2]], [N * N, N * J]):
        _lib.check(h.lib.nd4hip_dldltrs_batched(h.ptr, cnt, N, J, _off(LD, oL), sL, _off(y, oY), sY, _off(X, b0 * N * J)))
    return X


def hessenberg_decomp(A, device=None):
    """hessenberg.js:89-115: (U, H) with A = U H U^T, H upper Hessenberg."""
    A = np.asarray(A)
    if A.ndim < 2:
        raise ValueError("hessenberg_decomp(A): A must at least be 2D.")
    A = _asarray(A, "hessenberg_decomp(A)")
    N = A.shape[-1]
    if A.shape[-2] != N:
        raise ValueError("hessenberg_decomp(A): A must be square.")
    U, H = np.empty_like(A), np.empty_like(A)
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgehrd_batched(h.ptr, int(np.prod(A.shape[:-2], dtype=np.int64)), N, _ptr(A), _ptr(U), _ptr(H)))
    return U, H


def bidiag_decomp(A, device=None):
    """bidiag.js:245-319: (U [..., M, I], B [..., I, J], V [..., J, N]) with A = U B V, B upper bidiagonal."""
    A = np.asarray(A)
    if A.ndim < 2:
        raise ValueError("bidiag_decomp(A): A must be at least 2D.")
    if np.iscomplexobj(A):
        raise ValueError("bidiag_decomp(A): complex A not yet supported.")
    A = _asarray(A, "bidiag_decomp(A)")
    M, N = A.shape[-2:]
    I = min(M, N)
    J = I if M >= N else I + 1
    lead = A.shape[:-2]
    U, B, V = np.empty(lead + (M, I)), np.empty(lead + (I, J)), np.empty(lead + (J, N))
    h = _lib.handle(device)
    _lib.check(h.lib.nd4hip_dgebrd_batched(h.ptr, int(np.prod(lead, dtype=np.int64)), M, N, _ptr(A), _ptr(U), _ptr(B), _ptr(V)))
    return U, B, V
