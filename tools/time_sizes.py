"""Timing sweep over sizes (performance cliffs of the fallback paths): python tools/time_sizes.py [ops...] --n 1024 3072 4096"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nd4js_amd import dev  # noqa: E402

args = sys.argv[1:]
ns = [int(x) for x in args[args.index("--n") + 1:]] if "--n" in args else [1024, 3072, 4096]
ops = [a for a in (args[:args.index("--n")] if "--n" in args else args)] or ["lu", "qr", "chol"]
for n in ns:
    A = dev.fill_uniform(7, (n, n))
    S = dev.gemm_ex(False, True, 1.0, A, A, 0.0, torch.empty_like(A), n, n, n, n, n, n)
    S.diagonal().add_(float(n))
    for op in ops:
        fn = {"lu": lambda: dev.lu_decomp(A), "qr": lambda: dev.qr_decomp(A), "svd": lambda: dev.svd_decomp(A),
              "chol": lambda: dev.cholesky_decomp(S), "hess": lambda: dev.hessenberg_decomp(A), "bidiag": lambda: dev.bidiag_decomp(A), "ldl": lambda: dev.ldl_decomp(S), "matmul": lambda: dev.matmul2(A, A)}[op]
        fn(); torch.cuda.synchronize()
        reps = 1 if op in ("svd", "hess", "bidiag") else 3
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        print(op, n, "ms", round((time.perf_counter() - t) / reps * 1e3, 3), flush=True)
