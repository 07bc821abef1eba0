"""Device-resident form of the hot path: the same C ABI (`*_dev` entry points) on torch CUDA tensors.

PyTorch is plumbing only (device memory + the current HIP stream + torch.distributed); every
kernel that runs is from libnd4hip.so. Tensors must be float64, contiguous, on a HIP device.
"""
import ctypes

import torch

from . import _lib


def _h(t):
    h = _lib.handle(t.device.index)
    h.set_stream(torch.cuda.current_stream(t.device).cuda_stream)
    return h


def _chk(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous float64 CUDA tensor" % name)
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _batch(shape):
    n = 1
    for s in shape:
        n *= int(s)
    return n


def fill_uniform(seed, shape, device="cuda", offset=0):
    out = torch.empty(shape, dtype=torch.float64, device=device)
    h = _h(out)
    _lib.check(h.lib.nd4hip_fill_uniform_dev(h.ptr, seed, offset, out.numel(), _p(out)))
    return out


def matmul2(a, b, out=None):
    """[..., I, K] x [..., K, J]; leading dims must be equal or one operand plain 2-D (broadcast)."""
    _chk(a, "a"), _chk(b, "b")
    I, K = a.shape[-2:]
    J = b.shape[-1]
    if b.shape[-2] != K:
        raise ValueError("The last dimension of A and the 2nd to last dimension of B do not match.")
    la, lb = tuple(a.shape[:-2]), tuple(b.shape[:-2])
    if la == lb:
        lead, sA, sB = la, I * K, K * J
    elif _batch(lb) == 1:
        lead, sA, sB = la, I * K, 0
    elif _batch(la) == 1:
        lead, sA, sB = lb, 0, K * J
    else:
        raise ValueError("Shapes are not broadcast-compatible.")   # general broadcasting: host wrapper (la.py)
    batch = _batch(lead)
    if out is None:
        out = torch.empty(lead + (I, J), dtype=torch.float64, device=a.device)
    h = _h(a)
    _lib.check(h.lib.nd4hip_dgemm_batched_dev(h.ptr, batch, I, K, J, _p(a), sA if batch > 1 else 0,
                                              _p(b), sB if batch > 1 else 0, _p(_chk(out, "out"))))
    return out


def gemm_ex(transA, transB, alpha, A, B, beta, C, M, N, K, lda, ldb, ldc):
    h = _h(C)
    _lib.check(h.lib.nd4hip_dgemm_ex_dev(h.ptr, int(transA), int(transB), M, N, K, alpha, _p(A), lda, _p(B), ldb,
                                         beta, _p(C), ldc))
    return C


def lu_decomp(A):
    _chk(A, "A")
    N = A.shape[-1]
    if A.dim() < 2 or A.shape[-2] != N:
        raise ValueError("Last two dimensions must be quadratic.")
    LU = torch.empty_like(A)
    P = torch.empty(A.shape[:-1], dtype=torch.int32, device=A.device)
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgetrf_batched_dev(h.ptr, _batch(A.shape[:-2]), N, _p(A), _p(LU), ctypes.c_void_p(P.data_ptr())))
    return LU, P


def qr_decomp(A):
    _chk(A, "A")
    if A.dim() < 2:
        raise ValueError("qr_decomp(A): A.ndim must be at least 2.")
    M, N = A.shape[-2:]
    L = min(M, N)
    Q = torch.empty(tuple(A.shape[:-2]) + (M, L), dtype=torch.float64, device=A.device)
    R = torch.empty(tuple(A.shape[:-2]) + (L, N), dtype=torch.float64, device=A.device)
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgeqrf_q_batched_dev(h.ptr, _batch(A.shape[:-2]), M, N, _p(A), _p(Q), _p(R)))
    return Q, R


def svd_decomp(A, info=None):
    _chk(A, "A")
    if A.dim() < 2:
        raise ValueError("svd_decomp(A): A.ndim must be at least 2.")
    M, N = A.shape[-2:]
    L = min(M, N)
    lead = tuple(A.shape[:-2])
    U = torch.empty(lead + (M, L), dtype=torch.float64, device=A.device)
    sv = torch.empty(lead + (L,), dtype=torch.float64, device=A.device)
    V = torch.empty(lead + (L, N), dtype=torch.float64, device=A.device)
    sweeps, off = ctypes.c_int(0), ctypes.c_double(0.0)
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgesvdj_batched_dev(h.ptr, _batch(lead), M, N, _p(A), _p(U), _p(sv), _p(V),
                                                ctypes.byref(sweeps), ctypes.byref(off)))
    if info is not None:
        info["sweeps"], info["offnorm"] = sweeps.value, off.value
        info["rotations"] = h.svd_last_info()["rotations"]
    return U, sv, V


def lu_solve(LU, P, Y):
    """device-resident lu_solve (lu.js:84-177): LU [..., N, N], P [..., N] int32, Y [..., N, J] with equal leading dims"""
    _chk(LU, "LU"), _chk(Y, "Y")
    N, J = Y.shape[-2:]
    if LU.shape[-1] != N or LU.shape[-2] != N:
        raise ValueError("LU and y don't match.")
    lead = tuple(Y.shape[:-2])
    if tuple(LU.shape[:-2]) != lead or tuple(P.shape[:-1]) != lead:
        raise ValueError("LU and y are not broadcast-compatible.")   # general broadcasting: host wrapper (la.py)
    X = torch.empty(lead + (N, J), dtype=torch.float64, device=Y.device)
    h = _h(Y)
    b = _batch(lead)
    _lib.check(h.lib.nd4hip_dgetrs_batched_dev(h.ptr, b, N, J, _p(LU), N * N if b > 1 else 0, ctypes.c_void_p(P.data_ptr()), N if b > 1 else 0,
                                               _p(Y), N * J if b > 1 else 0, _p(X)))
    return X


def tri_solve(T, Y, upper, unit_diag=False):
    _chk(T, "T"), _chk(Y, "Y")
    M, J = Y.shape[-2:]
    lead = tuple(Y.shape[:-2])
    if tuple(T.shape[:-2]) != lead or T.shape[-1] != M or T.shape[-2] != M:
        raise ValueError("T and Y don't match.")
    X = torch.empty_like(Y)
    h = _h(Y)
    b = _batch(lead)
    _lib.check(h.lib.nd4hip_dtrsm_batched_dev(h.ptr, int(bool(upper)), int(bool(unit_diag)), b, M, J, _p(T), M * M if b > 1 else 0,
                                              _p(Y), M * J if b > 1 else 0, _p(X)))
    return X


def qr_lstsq(Q, R, Y):
    """device-resident qr_lstsq (qr.js:186-273): Q [..., N, M], R [..., M, I], Y [..., N, J] with equal leading dims"""
    _chk(Q, "Q"), _chk(R, "R"), _chk(Y, "Y")
    N, M = Q.shape[-2:]
    I, J = R.shape[-1], Y.shape[-1]
    if N != Y.shape[-2]:
        raise ValueError("qr_lstsq(Q,R,y): Q and y don't match.")
    if M != R.shape[-2]:
        raise ValueError("qr_lstsq(Q,R,y): Q and R don't match.")
    if I > N:
        raise ValueError("qr_lstsq(Q,R,y): Under-determined systems not supported. Use rrqr instead.")
    lead = tuple(Y.shape[:-2])
    if tuple(Q.shape[:-2]) != lead or tuple(R.shape[:-2]) != lead:
        raise ValueError("Q, R, y are not broadcast-compatible.")      # general broadcasting: host wrapper (la.py)
    X = torch.empty(lead + (I, J), dtype=torch.float64, device=Y.device)
    h = _h(Y)
    b = _batch(lead)
    _lib.check(h.lib.nd4hip_dqrls_batched_dev(h.ptr, b, N, M, I, J, _p(Q), N * M if b > 1 else 0, _p(R), M * I if b > 1 else 0,
                                              _p(Y), N * J if b > 1 else 0, _p(X)))
    return X


def svd_lstsq(U, sv, V, Y):
    """device-resident svd_lstsq (svd.js:100-228); singular values are not checked for NaN/Inf here (no host read-back)"""
    _chk(U, "U"), _chk(sv, "sv"), _chk(V, "V"), _chk(Y, "Y")
    N, M = U.shape[-2:]
    I, J = V.shape[-1], Y.shape[-1]
    if N != Y.shape[-2]:
        raise ValueError("svd_lstsq(U,sv,V, y): U and y don't match.")
    if M != sv.shape[-1]:
        raise ValueError("svd_lstsq(U,sv,V, y): U and sv don't match.")
    if M != V.shape[-2]:
        raise ValueError("svd_lstsq(U,sv,V, y): V and sv don't match.")
    lead = tuple(Y.shape[:-2])
    if tuple(U.shape[:-2]) != lead or tuple(V.shape[:-2]) != lead or tuple(sv.shape[:-1]) != lead:
        raise ValueError("svd_lstsq(U,sv,V, y): U,sv,V,y not broadcast-compatible.")
    X = torch.empty(lead + (I, J), dtype=torch.float64, device=Y.device)
    h = _h(Y)
    b = _batch(lead)
    _lib.check(h.lib.nd4hip_dsvdls_batched_dev(h.ptr, b, N, M, I, J, _p(U), N * M if b > 1 else 0, _p(sv), M if b > 1 else 0,
                                               _p(V), M * I if b > 1 else 0, _p(Y), N * J if b > 1 else 0, _p(X)))
    return X


def qr_decomp_full(A):
    _chk(A, "A")
    M, N = A.shape[-2:]
    lead = tuple(A.shape[:-2])
    Q = torch.empty(lead + (M, M), dtype=torch.float64, device=A.device)
    R = torch.empty(lead + (M, N), dtype=torch.float64, device=A.device)
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgeqrf_full_batched_dev(h.ptr, _batch(lead), M, N, _p(A), _p(Q), _p(R)))
    return Q, R


def qr_decomp_inplace(A, Y):
    """_qr_decomp_inplace (qr.js:146-183) on device tensors, in place: A <- R, Y <- Q^T Y"""
    _chk(A, "A"), _chk(Y, "Y")
    if tuple(A.shape[:-1]) != tuple(Y.shape[:-1]):
        raise ValueError("Assertion failed.")
    M, N = A.shape[-2:]
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgeqrf_qty_batched_dev(h.ptr, _batch(A.shape[:-2]), M, N, Y.shape[-1], _p(A), _p(Y)))
    return A, Y


def cholesky_decomp(S):
    """device-resident cholesky_decomp (cholesky.js:51-71); one flag read-back decides the reference's singularity error"""
    _chk(S, "S")
    N = S.shape[-1]
    if S.dim() < 2 or S.shape[-2] != N:
        raise ValueError("Last two dimensions must be quadratic.")
    L = torch.empty_like(S)
    h = _h(S)
    try:
        _lib.check(h.lib.nd4hip_dpotrf_batched_dev(h.ptr, _batch(S.shape[:-2]), N, _p(S), _p(L)))
    except _lib.Nd4HipError as e:
        if e.code == -5:
            raise ValueError("Matrix contains NaNs or is (near) singular.")
        raise
    return L


def cholesky_solve(L, Y):
    _chk(L, "L"), _chk(Y, "Y")
    N, J = Y.shape[-2:]
    if L.shape[-1] != L.shape[-2]:
        raise ValueError("Last two dimensions of L must be quadratic.")
    if L.shape[-1] != N:
        raise ValueError("L and y don't match.")
    lead = tuple(Y.shape[:-2])
    if tuple(L.shape[:-2]) != lead:
        raise ValueError("Shapes are not broadcast-compatible.")       # general broadcasting: host wrapper (la.py)
    X = torch.empty_like(Y)
    h = _h(Y)
    b = _batch(lead)
    _lib.check(h.lib.nd4hip_dpotrs_batched_dev(h.ptr, b, N, J, _p(L), N * N if b > 1 else 0, _p(Y), N * J if b > 1 else 0, _p(X)))
    return X


def ldl_decomp(S):
    _chk(S, "S")
    N = S.shape[-1]
    if S.dim() < 2 or S.shape[-2] != N:
        raise ValueError("Last two dimensions must be quadratic.")
    LD = torch.empty_like(S)
    h = _h(S)
    _lib.check(h.lib.nd4hip_dldltrf_batched_dev(h.ptr, _batch(S.shape[:-2]), N, _p(S), _p(LD)))
    return LD


def ldl_solve(LD, Y):
    _chk(LD, "LD"), _chk(Y, "Y")
    N, J = Y.shape[-2:]
    if LD.shape[-1] != LD.shape[-2]:
        raise ValueError("ldl_solve(LD,y): Last two dimensions of LD must be quadratic.")
    if LD.shape[-1] != N:
        raise ValueError("ldl_solve(LD,y): LD and y don't match.")
    lead = tuple(Y.shape[:-2])
    if tuple(LD.shape[:-2]) != lead:
        raise ValueError("Shapes are not broadcast-compatible.")       # general broadcasting: host wrapper (la.py)
    X = torch.empty_like(Y)
    h = _h(Y)
    b = _batch(lead)
    _lib.check(h.lib.nd4hip_dldltrs_batched_dev(h.ptr, b, N, J, _p(LD), N * N if b > 1 else 0, _p(Y), N * J if b > 1 else 0, _p(X)))
    return X


def hessenberg_decomp(A):
    _chk(A, "A")
    N = A.shape[-1]
    if A.dim() < 2 or A.shape[-2] != N:
        raise ValueError("hessenberg_decomp(A): A must be square.")
    U, H = torch.empty_like(A), torch.empty_like(A)
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgehrd_batched_dev(h.ptr, _batch(A.shape[:-2]), N, _p(A), _p(U), _p(H)))
    return U, H


def bidiag_decomp(A):
    _chk(A, "A")
    if A.dim() < 2:
        raise ValueError("bidiag_decomp(A): A must be at least 2D.")
    M, N = A.shape[-2:]
    I = min(M, N)
    J = I if M >= N else I + 1
    lead = tuple(A.shape[:-2])
    U = torch.empty(lead + (M, I), dtype=torch.float64, device=A.device)
    B = torch.empty(lead + (I, J), dtype=torch.float64, device=A.device)
    V = torch.empty(lead + (J, N), dtype=torch.float64, device=A.device)
    h = _h(A)
    _lib.check(h.lib.nd4hip_dgebrd_batched_dev(h.ptr, _batch(lead), M, N, _p(A), _p(U), _p(B), _p(V)))
    return U, B, V
