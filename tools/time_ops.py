import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd4js_amd import dev
ops = sys.argv[1:] or ["lu", "qr"]
for n in (2048,):
    A = dev.fill_uniform(7, (n, n))
    S = dev.gemm_ex(False, True, 1.0, A, A, 0.0, torch.empty_like(A), n, n, n, n, n, n)
    S.diagonal().add_(float(n))
    for op in ops:
        fn = {"lu": dev.lu_decomp, "qr": dev.qr_decomp, "svd": dev.svd_decomp, "matmul": lambda x: dev.matmul2(x, x),
              "chol": lambda x: dev.cholesky_decomp(S)}[op]
        fn(A); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            fn(A)
        torch.cuda.synchronize()
        print(op, n, "ms", round((time.perf_counter() - t) / 5 * 1e3, 3), flush=True)
