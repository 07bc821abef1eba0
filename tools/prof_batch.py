#!/usr/bin/env python3
"""Profiling driver for the batched configs: rocprofv3 --kernel-trace --stats -- python3 tools/prof_batch.py <batch> [svd|lu|qr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ops = sys.argv[2:] or ["svd"]
X = dev.fill_uniform(1000, (b, 512, 512))
for op in ops:
    {"svd": dev.svd_decomp, "lu": dev.lu_decomp, "qr": dev.qr_decomp}[op](X)
torch.cuda.synchronize()
print("done", ops)
