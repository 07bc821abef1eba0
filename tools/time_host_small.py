"""per-call latency of the host-pointer path on small inputs: python tools/time_host_small.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nd4js_amd import la, rng
for n in (16, 64, 256, 1024):
    a, b = rng.matrix(5, n, n), rng.matrix(6, n, n)
    for name, fn in (("matmul2", lambda: la.matmul2(a, b)), ("lu_decomp", lambda: la.lu_decomp(a)), ("qr_decomp", lambda: la.qr_decomp(a)),
                     ("svd_decomp", lambda: la.svd_decomp(a))):
        fn()
        reps = 20 if n <= 256 else 5
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        print("%-10s n=%-5d us/call %9.1f" % (name, n, (time.perf_counter() - t) / reps * 1e6), flush=True)
