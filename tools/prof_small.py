#!/usr/bin/env python3
"""Profiling driver for the one-launch small SVD (svd.hip: jac_small): 8192 x 32^2 and one 32^2 matrix, three calls each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nd4js_amd import dev  # noqa: E402
SB = dev.fill_uniform(32, (8192, 32, 32))
S1 = dev.fill_uniform(31, (32, 32))
for _ in range(3):
    dev.svd_decomp(SB)
    dev.svd_decomp(S1)
torch.cuda.synchronize()
print("done")
