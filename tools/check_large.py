"""one pass over the factorisations at a large size with residual checks on the device: python tools/check_large.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A = dev.fill_uniform(7, (N, N))
eye = torch.eye(N, dtype=torch.float64, device="cuda")
def T(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print("%-10s N=%d  %8.1f ms" % (name, N, (time.perf_counter() - t) * 1e3), flush=True); return out
C = T("matmul2", lambda: dev.matmul2(A, A))
LU, P = T("lu", lambda: dev.lu_decomp(A))
L = torch.tril(LU, -1) + eye; U = torch.triu(LU)
print("  |LU - A[P]| / |A| =", float((dev.matmul2(L, U) - A[P.long()]).norm() / A.norm()), " max|L| =", float(torch.tril(LU, -1).abs().max()))
Q, R = T("qr", lambda: dev.qr_decomp(A))
print("  |QR - A| / |A| =", float((dev.matmul2(Q, R) - A).norm() / A.norm()), " |Q^T Q - I|max =", float((dev.matmul2(Q.T.contiguous(), Q) - eye).abs().max()))
S = dev.matmul2(A, A.T.contiguous()); S.diagonal().add_(float(N))
Lc = T("cholesky", lambda: dev.cholesky_decomp(S))
print("  |LL^T - S| / |S| =", float((dev.matmul2(Lc, Lc.T.contiguous()) - S).norm() / S.norm()))
