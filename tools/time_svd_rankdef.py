"""SVD of rank-deficient input (the completion of the null-space vectors): python tools/time_svd_rankdef.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
for (n, r) in ((512, 500), (512, 64), (1024, 64), (2048, 64), (2048, 1)):
    B = dev.fill_uniform(3, (n, r)); C = dev.fill_uniform(4, (r, n))
    A = dev.matmul2(B, C)
    dev.svd_decomp(A); torch.cuda.synchronize()
    info = {}
    t = time.perf_counter()
    U, sv, V = dev.svd_decomp(A, info=info)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    ortho = float((V @ V.transpose(-1, -2) - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max())
    print("n", n, "rank", r, "ms", round(dt * 1e3, 2), "sweeps", info.get("sweeps"), "max|VV^T-I|", ortho, flush=True)
