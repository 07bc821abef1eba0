"""NaN / Inf / zero inputs through every op: nothing may hang or fault (results are whatever IEEE gives): python tools/fuzz_special.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nd4js_amd import la, rng, _lib
for n in (5, 40, 130, 300):
    base = rng.matrix(1, n, n)
    for kind in ("nan", "inf", "zero", "nanrow", "huge", "tiny"):
        a = base.copy()
        if kind == "nan": a[n // 2, n // 3] = np.nan
        if kind == "inf": a[n // 3, n // 2] = np.inf
        if kind == "zero": a[:] = 0.0
        if kind == "nanrow": a[1, :] = np.nan
        if kind == "huge": a *= 1e300
        if kind == "tiny": a *= 1e-310
        y = rng.matrix(2, n, 3)
        for name, fn in (("matmul2", lambda: la.matmul2(a, a)), ("lu", lambda: la.lu_solve(la.lu_decomp(a), y)), ("qr", lambda: la.qr_lstsq(la.qr_decomp(a), y)),
                         ("svd", lambda: la.svd_decomp(a)), ("chol", lambda: la.cholesky_decomp(a @ a.T + np.eye(n))), ("ldl", lambda: la.ldl_decomp(a + a.T)),
                         ("hess", lambda: la.hessenberg_decomp(a)), ("bidiag", lambda: la.bidiag_decomp(a))):
            try:
                fn(); status = "ok"
            except (_lib.Nd4HipError, ValueError) as e:
                status = "raised: " + str(e)[:60]
            print(n, kind, name, status, flush=True)
print("fuzz done")
