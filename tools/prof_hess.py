import sys
sys.path.insert(0, "/root/repo")
import torch
from nd4js_amd import dev
A = dev.fill_uniform(7, (2048, 2048))
dev.hessenberg_decomp(A); torch.cuda.synchronize()
