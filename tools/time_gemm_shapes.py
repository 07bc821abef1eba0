"""matmul2 timing over awkward shapes: python tools/time_gemm_shapes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
for (I, K, J) in ((4096, 4096, 4096), (4095, 4095, 4095), (4097, 4097, 4097), (4096, 4097, 4096), (4096, 64, 4096), (4096, 4096, 64),
                  (64, 4096, 4096), (8192, 512, 512), (1000, 1000, 1000), (1001, 1001, 1001)):
    a = dev.fill_uniform(1, (I, K)); b = dev.fill_uniform(2, (K, J))
    dev.matmul2(a, b); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        dev.matmul2(a, b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print((I, K, J), "ms", round(dt * 1e3, 3), "TFLOP/s", round(2.0 * I * K * J / dt / 1e12, 1), flush=True)
