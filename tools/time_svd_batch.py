#!/usr/bin/env python3
"""batched 512^2 SVD timing: python tools/time_svd_batch.py [batch]   (ND4HIP_SVD_CHUNK selects the chunk size)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
X = torch.empty((b, 512, 512), dtype=torch.float64, device="cuda")
for k in range(b):
    X[k] = dev.fill_uniform(1000 + k, (512, 512))
dev.svd_decomp(X[:8])
torch.cuda.synchronize()
info = {}
t = time.perf_counter()
U, sv, V = dev.svd_decomp(X, info=info)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("chunk", os.environ.get("ND4HIP_SVD_CHUNK", "off"), "batch", b, "s", round(dt, 4), "sweeps", info.get("sweeps"), "sv0", float(sv[0, 0]), float(sv[-1, -1]), flush=True)
