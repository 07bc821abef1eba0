"""End-to-end timing of the host-pointer C ABI (H2D + kernel + D2H): python tools/time_host_path.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nd4js_amd import la, rng
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a, b = rng.matrix(5, N, N), rng.matrix(6, N, N)
la.matmul2(a, b)
for name, fn in (("matmul2", lambda: la.matmul2(a, b)), ("lu_decomp", lambda: la.lu_decomp(a[:2048, :2048].copy())),
                 ("qr_decomp", lambda: la.qr_decomp(a[:2048, :2048].copy()))):
    fn()
    t = time.perf_counter()
    for _ in range(3):
        fn()
    print(name, "host path ms", round((time.perf_counter() - t) / 3 * 1e3, 2), flush=True)
