"""ctypes binding of oracle/libnd4_oracle.so — *** TEST INFRASTRUCTURE ***.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (nd4js_amd, libnd4hip.so) never does. See oracle/nd4_oracle.h for the
reference file:line each function restates; parity is pinned by tests/test_oracle_golden.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnd4_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.c_int64


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("nd4_oracle.c", "nd4_oracle_svd_dc.c", "nd4_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("ND4_ORACLE_SO")            # e.g. the sanitizer build (make -C oracle asan; tools/check_sanitize.sh)
        if not so:
            build()
            so = _SO
        L = ctypes.CDLL(so)
        L.nd4o_uniform.restype = ctypes.c_double
        L.nd4o_uniform.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        L.nd4o_fill_uniform.argtypes = [ctypes.c_uint32, ctypes.c_uint32, _i64, _dp]
        L.nd4o_matmul2.restype = ctypes.c_int
        L.nd4o_matmul2.argtypes = [ctypes.c_int, _ip, _dp, ctypes.c_int, _ip, _dp, _ip, _dp]
        L.nd4o_matmul_batched.argtypes = [_i64, _i64, _i64, _i64, _dp, _i64, _dp, _i64, _dp]
        L.nd4o_qr_decomp_full.argtypes = [_i64, _i64, _i64, _dp, _dp, _dp]
        L.nd4o_qr_decomp.argtypes = [_i64, _i64, _i64, _dp, _dp, _dp]
        L.nd4o_lu_decomp.argtypes = [_i64, _i64, _dp, _dp, _ip]
        L.nd4o_tril_solve.argtypes = [_i64, _i64, _i64, _dp, _i64, _dp]
        L.nd4o_triu_solve.argtypes = [_i64, _i64, _i64, _dp, _i64, _dp]
        L.nd4o_lu_solve.argtypes = [_i64, _i64, _i64, _dp, _i64, _ip, _i64, _dp, _i64, _dp]
        L.nd4o_cholesky_decomp.argtypes = [_i64, _i64, _dp, _dp]
        L.nd4o_cholesky_decomp.restype = ctypes.c_int
        L.nd4o_cholesky_solve.argtypes = [_i64, _i64, _i64, _dp, _i64, _dp, _i64, _dp]
        L.nd4o_ldl_decomp.argtypes = [_i64, _i64, _dp, _dp]
        L.nd4o_ldl_solve.argtypes = [_i64, _i64, _i64, _dp, _i64, _dp, _i64, _dp]
        L.nd4o_hessenberg_decomp.argtypes = [_i64, _dp, _dp]
        L.nd4o_bidiag_decomp.argtypes = [_i64, _i64, _dp, _dp, _dp, _dp, _dp]
        L.nd4o_qr_decomp_inplace.argtypes = [_i64, _i64, _i64, _dp, _dp]
        L.nd4o_qr_lstsq.argtypes = [_i64] * 5 + [_dp, _i64, _dp, _i64, _dp, _i64, _dp]
        L.nd4o_svd_lstsq.argtypes = [_i64] * 5 + [_dp, _i64, _dp, _i64, _dp, _i64, _dp, _i64, _dp, _dp]
        L.nd4o_svd_lstsq.restype = ctypes.c_int
        L.nd4o_svd_dc.restype = ctypes.c_int
        L.nd4o_svd_dc.argtypes = [_i64, _i64, _i64, _dp, _dp, _dp, _dp]
        L.nd4o_svd_jac_2sided.restype = ctypes.c_int
        L.nd4o_svd_jac_2sided.argtypes = [_i64, _i64, _dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def fill_uniform(seed, n, offset=0):
    out = np.empty(int(n), dtype=np.float64)
    lib().nd4o_fill_uniform(seed, offset, out.size, _d(out))
    return out


def matmul2(a, b):
    a, b = _f64(a), _f64(b)
    if a.ndim < 2:
        raise ValueError("A must be at least 2D.")
    if b.ndim < 2:
        raise ValueError("B must be at least 2D.")
    sa = np.asarray(a.shape, dtype=np.int32)
    sb = np.asarray(b.shape, dtype=np.int32)
    nd = max(a.ndim, b.ndim)
    lead = np.broadcast_shapes(a.shape[:-2], b.shape[:-2]) if a.shape[-1] == b.shape[-2] else ()
    sc = np.zeros(nd, dtype=np.int32)
    c = np.empty(tuple(lead) + (a.shape[-2], b.shape[-1]), dtype=np.float64)
    rc = lib().nd4o_matmul2(a.ndim, _i(sa), _d(a), b.ndim, _i(sb), _d(b), _i(sc), _d(c))
    if rc == -1:
        raise ValueError("The last dimension of A and the 2nd to last dimension of B do not match.")
    if rc == -2:
        raise ValueError("Shapes are not broadcast-compatible.")
    assert tuple(sc) == c.shape, (tuple(sc), c.shape)
    return c


def qr_decomp_full(a):
    a = _f64(a)
    M, N = a.shape[-2:]
    batch = int(np.prod(a.shape[:-2], dtype=np.int64))
    q = np.empty(a.shape[:-2] + (M, M))
    r = np.empty(a.shape[:-2] + (M, N))
    lib().nd4o_qr_decomp_full(batch, M, N, _d(a), _d(q), _d(r))
    return q, r


def qr_decomp(a):
    a = _f64(a)
    M, N = a.shape[-2:]
    L = min(M, N)
    batch = int(np.prod(a.shape[:-2], dtype=np.int64))
    q = np.empty(a.shape[:-2] + (M, L))
    r = np.empty(a.shape[:-2] + (L, N))
    lib().nd4o_qr_decomp(batch, M, N, _d(a), _d(q), _d(r))
    return q, r


def lu_decomp(a):
    a = _f64(a)
    N = a.shape[-1]
    if a.shape[-2] != N:
        raise ValueError("Last two dimensions must be quadratic.")
    batch = int(np.prod(a.shape[:-2], dtype=np.int64))
    lu = np.empty_like(a)
    p = np.empty(a.shape[:-1], dtype=np.int32)
    lib().nd4o_lu_decomp(batch, N, _d(a), _d(lu), _i(p))
    return lu, p


def svd_dc(a):
    """nd.la.svd_decomp = svd_dc (svd_dc.js:883-932): bidiagonalisation + divide & conquer, any M x N, leading batch dims."""
    a = _f64(a)
    M, N = a.shape[-2:]
    L = min(M, N)
    lead = a.shape[:-2]
    batch = int(np.prod(lead, dtype=np.int64))
    u = np.empty(lead + (M, L))
    sv = np.empty(lead + (L,))
    v = np.empty(lead + (L, N))
    rc = lib().nd4o_svd_dc(batch, M, N, _d(a), _d(u), _d(sv), _d(v))
    if rc != 0:
        raise RuntimeError("svd_dc: 'Assertion failed.' (svd_dc.js:%d)" % rc if rc > 0 else "svd_dc: out of memory")
    return u, sv, v


svd_decomp = svd_dc          # svd.js:25


def svd_jac_2sided(a):
    """Square input only (the reference's rectangular pre-reduction is a host composition)."""
    a = _f64(a)
    N = a.shape[-1]
    assert a.shape[-2] == N
    batch = int(np.prod(a.shape[:-2], dtype=np.int64))
    u = np.empty_like(a)
    v = np.empty_like(a)
    sv = np.empty(a.shape[:-1])
    sweeps = lib().nd4o_svd_jac_2sided(batch, N, _d(a), _d(u), _d(sv), _d(v))
    return u, sv, v, sweeps


def _bcast3(lead_shapes):
    return np.broadcast_shapes(*lead_shapes)


def _tri_solve(fn, T, Y):
    T, Y = _f64(T), _f64(Y)
    M, O = Y.shape[-2:]
    lead = _bcast3([T.shape[:-2], Y.shape[:-2]])
    Tb = np.ascontiguousarray(np.broadcast_to(T, lead + (M, M)))
    X = np.ascontiguousarray(np.broadcast_to(Y, lead + (M, O))).copy()
    batch = int(np.prod(lead, dtype=np.int64))
    fn(batch, M, O, _d(Tb), M * M, _d(X))
    return X


def tril_solve(L, Y):
    return _tri_solve(lib().nd4o_tril_solve, L, Y)


def triu_solve(U, Y):
    return _tri_solve(lib().nd4o_triu_solve, U, Y)


def lu_solve(LU, P, Y):
    LU, Y = _f64(LU), _f64(Y)
    P = np.ascontiguousarray(P, dtype=np.int32)
    N, J = Y.shape[-2:]
    lead = _bcast3([LU.shape[:-2], P.shape[:-1], Y.shape[:-2]])
    LUb = np.ascontiguousarray(np.broadcast_to(LU, lead + (N, N)))
    Pb = np.ascontiguousarray(np.broadcast_to(P, lead + (N,)))
    Yb = np.ascontiguousarray(np.broadcast_to(Y, lead + (N, J)))
    X = np.empty(lead + (N, J))
    batch = int(np.prod(lead, dtype=np.int64))
    lib().nd4o_lu_solve(batch, N, J, _d(LUb), N * N, _i(Pb), N, _d(Yb), N * J, _d(X))
    return X


def qr_lstsq(Q, R, Y):
    """qr.js:186-273"""
    Q, R, Y = _f64(Q), _f64(R), _f64(Y)
    N, M = Q.shape[-2:]
    I, J = R.shape[-1], Y.shape[-1]
    if N != Y.shape[-2]:
        raise ValueError("qr_lstsq(Q,R,y): Q and y don't match.")
    if M != R.shape[-2]:
        raise ValueError("qr_lstsq(Q,R,y): Q and R don't match.")
    if I > N:
        raise ValueError("qr_lstsq(Q,R,y): Under-determined systems not supported. Use rrqr instead.")
    lead = _bcast3([Q.shape[:-2], R.shape[:-2], Y.shape[:-2]])
    Qb = np.ascontiguousarray(np.broadcast_to(Q, lead + (N, M)))
    Rb = np.ascontiguousarray(np.broadcast_to(R, lead + (M, I)))
    Yb = np.ascontiguousarray(np.broadcast_to(Y, lead + (N, J)))
    X = np.empty(lead + (I, J))
    lib().nd4o_qr_lstsq(int(np.prod(lead, dtype=np.int64)), N, M, I, J, _d(Qb), N * M, _d(Rb), M * I, _d(Yb), N * J, _d(X))
    return X


def svd_lstsq(U, sv, V, Y):
    """svd.js:100-228"""
    U, sv, V, Y = _f64(U), _f64(sv), _f64(V), _f64(Y)
    N, M = U.shape[-2:]
    I, J = V.shape[-1], Y.shape[-1]
    if N != Y.shape[-2]:
        raise ValueError("svd_lstsq(U,sv,V, y): U and y don't match.")
    if M != sv.shape[-1]:
        raise ValueError("svd_lstsq(U,sv,V, y): U and sv don't match.")
    if M != V.shape[-2]:
        raise ValueError("svd_lstsq(U,sv,V, y): V and sv don't match.")
    lead = _bcast3([U.shape[:-2], sv.shape[:-1], V.shape[:-2], Y.shape[:-2]])
    Ub = np.ascontiguousarray(np.broadcast_to(U, lead + (N, M)))
    Sb = np.ascontiguousarray(np.broadcast_to(sv, lead + (M,)))
    Vb = np.ascontiguousarray(np.broadcast_to(V, lead + (M, I)))
    Yb = np.ascontiguousarray(np.broadcast_to(Y, lead + (N, J)))
    X = np.empty(lead + (I, J))
    tmp = np.empty(M * J)
    rc = lib().nd4o_svd_lstsq(int(np.prod(lead, dtype=np.int64)), N, M, I, J, _d(Ub), N * M, _d(Sb), M, _d(Vb), M * I, _d(Yb), N * J,
                              _d(X), _d(tmp))
    if rc:
        raise ValueError("svd_solve(): NaN or Infinity encountered.")
    return X


def qr_decomp_inplace(A, Y):
    """qr.js:146-183 on copies: returns (R [..., M, N], Q^T Y [..., M, L])"""
    A, Y = _f64(A).copy(), _f64(Y).copy()
    M, N = A.shape[-2:]
    L = Y.shape[-1]
    a2, y2 = A.reshape(-1, M, N), Y.reshape(-1, M, L)
    for b in range(a2.shape[0]):
        lib().nd4o_qr_decomp_inplace(M, N, L, _d(a2[b]), _d(y2[b]))
    return A, Y


def cholesky_decomp(S):
    """cholesky.js:51-71"""
    S = _f64(S)
    N = S.shape[-1]
    if S.ndim < 2 or S.shape[-2] != N:
        raise ValueError("Last two dimensions must be quadratic.")
    L = np.empty_like(S)
    if lib().nd4o_cholesky_decomp(int(np.prod(S.shape[:-2], dtype=np.int64)), N, _d(S), _d(L)):
        raise ValueError("Matrix contains NaNs or is (near) singular.")
    return L


def cholesky_solve(L, Y):
    """cholesky.js:74-150"""
    L, Y = _f64(L), _f64(Y)
    if L.ndim < 2:
        raise ValueError("L must be at least 2D.")
    if Y.ndim < 2:
        raise ValueError("y must be at least 2D.")
    N, J = Y.shape[-2:]
    if L.shape[-1] != L.shape[-2]:
        raise ValueError("Last two dimensions of L must be quadratic.")
    if L.shape[-1] != N:
        raise ValueError("L and y don't match.")
    try:
        lead = _bcast3([L.shape[:-2], Y.shape[:-2]])
    except ValueError:
        raise ValueError("Shapes are not broadcast-compatible.")
    Lb = np.ascontiguousarray(np.broadcast_to(L, lead + (N, N)))
    Yb = np.ascontiguousarray(np.broadcast_to(Y, lead + (N, J)))
    X = np.empty(lead + (N, J))
    lib().nd4o_cholesky_solve(int(np.prod(lead, dtype=np.int64)), N, J, _d(Lb), N * N, _d(Yb), N * J, _d(X))
    return X


def ldl_decomp(S):
    """ldl.js:67-90"""
    S = _f64(S)
    N = S.shape[-1]
    if S.ndim < 2 or S.shape[-2] != N:
        raise ValueError("Last two dimensions must be quadratic.")
    LD = np.empty_like(S)
    lib().nd4o_ldl_decomp(int(np.prod(S.shape[:-2], dtype=np.int64)), N, _d(S), _d(LD))
    return LD


def ldl_solve(LD, Y):
    """ldl.js:133-201"""
    LD, Y = _f64(LD), _f64(Y)
    if LD.ndim < 2:
        raise ValueError("ldl_solve(LD,y): LD must be at least 2D.")
    if Y.ndim < 2:
        raise ValueError("ldl_solve(LD,y): y must be at least 2D.")
    N, J = Y.shape[-2:]
    if LD.shape[-1] != LD.shape[-2]:
        raise ValueError("ldl_solve(LD,y): Last two dimensions of LD must be quadratic.")
    if LD.shape[-1] != N:
        raise ValueError("ldl_solve(LD,y): LD and y don't match.")
    try:
        lead = _bcast3([LD.shape[:-2], Y.shape[:-2]])
    except ValueError:
        raise ValueError("Shapes are not broadcast-compatible.")
    Lb = np.ascontiguousarray(np.broadcast_to(LD, lead + (N, N)))
    Yb = np.ascontiguousarray(np.broadcast_to(Y, lead + (N, J)))
    X = np.empty(lead + (N, J))
    lib().nd4o_ldl_solve(int(np.prod(lead, dtype=np.int64)), N, J, _d(Lb), N * N, _d(Yb), N * J, _d(X))
    return X


def hessenberg_decomp(A):
    """hessenberg.js:89-115: returns (U, H) with A = U H U^T"""
    A = _f64(A)
    if A.ndim < 2:
        raise ValueError("hessenberg_decomp(A): A must at least be 2D.")
    N = A.shape[-1]
    if A.shape[-2] != N:
        raise ValueError("hessenberg_decomp(A): A must be square.")
    H = A.copy()
    U = np.zeros_like(H)
    h2, u2 = H.reshape(-1, N, N), U.reshape(-1, N, N)
    for b in range(h2.shape[0] - 1, -1, -1):
        lib().nd4o_hessenberg_decomp(N, _d(u2[b]), _d(h2[b]))
    return U, H


def bidiag_decomp(A):
    """bidiag.js:245-319: (U [..., M, I], B [..., I, J], V [..., J, N]) with A = U B V"""
    A = _f64(A)
    if A.ndim < 2:
        raise ValueError("bidiag_decomp(A): A must be at least 2D.")
    M, N = A.shape[-2:]
    I = min(M, N)
    J = I if M >= N else I + 1
    lead = A.shape[:-2]
    U, B, V = np.empty(lead + (M, I)), np.empty(lead + (I, J)), np.empty(lead + (J, N))
    a2, u2, b2, v2 = A.reshape(-1, M, N), U.reshape(-1, M, I), B.reshape(-1, I, J), V.reshape(-1, J, N)
    tmp = np.empty(N)
    for k in range(a2.shape[0]):
        lib().nd4o_bidiag_decomp(M, N, _d(a2[k]), _d(u2[k]), _d(b2[k]), _d(v2[k]), _d(tmp))
    return U, B, V
