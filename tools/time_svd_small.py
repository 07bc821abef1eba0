"""batched small SVDs: python tools/time_svd_small.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
for (b, n) in ((8192, 16), (8192, 32), (2048, 64), (1024, 96), (512, 128), (256, 256)):
    A = dev.fill_uniform(7, (b, n, n))
    dev.svd_decomp(A); torch.cuda.synchronize()
    info = {}
    t = time.perf_counter()
    dev.svd_decomp(A, info=info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("svd batch", b, "n", n, "ms", round(dt * 1e3, 2), "us/matrix", round(dt / b * 1e6, 2), "sweeps", info.get("sweeps"), flush=True)
