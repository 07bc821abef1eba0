"""batches of small problems through every device entry point: python tools/time_small_battery.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev

def T(name, fn, b):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    fn(); fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 2
    print("%-28s ms %8.3f  us/problem %8.2f" % (name, dt * 1e3, dt / b * 1e6), flush=True)

for (b, n) in ((4096, 32), (1024, 64), (256, 200)):
    A = dev.fill_uniform(7, (b, n, n))
    S = torch.matmul(A, A.transpose(-1, -2)) + n * torch.eye(n, dtype=torch.float64, device="cuda")
    Y = dev.fill_uniform(8, (b, n, 3))
    tag = " b=%d n=%d" % (b, n)
    LU, P = dev.lu_decomp(A); Q, R = dev.qr_decomp(A); L = dev.cholesky_decomp(S); usv = dev.svd_decomp(A)
    T("lu_decomp" + tag, lambda: dev.lu_decomp(A), b)
    T("qr_decomp" + tag, lambda: dev.qr_decomp(A), b)
    T("cholesky_decomp" + tag, lambda: dev.cholesky_decomp(S), b)
    T("ldl_decomp" + tag, lambda: dev.ldl_decomp(S), b)
    T("svd_decomp" + tag, lambda: dev.svd_decomp(A), b)
    T("hessenberg_decomp" + tag, lambda: dev.hessenberg_decomp(A), b)
    T("bidiag_decomp" + tag, lambda: dev.bidiag_decomp(A), b)
    T("lu_solve" + tag, lambda: dev.lu_solve(LU, P, Y), b)
    T("cholesky_solve" + tag, lambda: dev.cholesky_solve(L, Y), b)
    T("qr_lstsq" + tag, lambda: dev.qr_lstsq(Q, R, Y), b)
    T("svd_lstsq" + tag, lambda: dev.svd_lstsq(*usv, Y), b)
    T("tri_solve" + tag, lambda: dev.tri_solve(R, Y, True), b)
