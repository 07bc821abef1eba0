"""tall-skinny QR / SVD timing: python tools/time_tall.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
for (M, N) in ((4096, 64), (16384, 64), (65536, 32), (8192, 256), (32768, 256), (3000, 1000)):
    A = dev.fill_uniform(3, (M, N))
    for name, fn in (("qr", lambda: dev.qr_decomp(A)), ("svd", lambda: dev.svd_decomp(A))):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        fn(); fn()
        torch.cuda.synchronize()
        print(name, (M, N), "ms", round((time.perf_counter() - t) / 2 * 1e3, 3), flush=True)
