#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats -- python3 tools/prof_hess.py [N] [hess] [bidiag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
what = sys.argv[2:] or ["hess"]
A = dev.fill_uniform(7, (N, N))
for _ in range(2):
    if "hess" in what:
        dev.hessenberg_decomp(A)
    if "bidiag" in what:
        dev.bidiag_decomp(A)
torch.cuda.synchronize()
print("done")
