"""batched factorisation timing: python tools/time_batched.py [batch] [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
A = dev.fill_uniform(7, (b, n, n))
S = torch.matmul(A, A.transpose(-1, -2)) + n * torch.eye(n, dtype=torch.float64, device="cuda")
for name, fn, fl in (("lu", lambda: dev.lu_decomp(A), 2 / 3 * n ** 3), ("qr", lambda: dev.qr_decomp(A), 8 / 3 * n ** 3),
                     ("chol", lambda: dev.cholesky_decomp(S), n ** 3 / 3), ("matmul", lambda: dev.matmul2(A, A), 2.0 * n ** 3)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(name, "batch", b, "n", n, "ms", round(dt * 1e3, 3), "TFLOP/s", round(fl * b / dt / 1e12, 2), flush=True)
