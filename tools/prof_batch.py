#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import dev
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
X = dev.fill_uniform(1000, (b, 512, 512))
dev.svd_decomp(X)
torch.cuda.synchronize()
print("done")
