"""Wall time of the batched 512 x 512 factorisations (device-resident): python tools/time_batch.py [batch] [lu] [qr] [svd]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ops = sys.argv[2:] or ["lu", "qr"]
X = dev.fill_uniform(1000, (b, 512, 512))
for op in ops:
    fn = {"svd": dev.svd_decomp, "lu": dev.lu_decomp, "qr": dev.qr_decomp}[op]
    fn(X); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        fn(X)
    torch.cuda.synchronize()
    print(op, b, "x 512^2 ms", round((time.perf_counter() - t) / 3 * 1e3, 2), flush=True)
