"""Calibration: what a plain device-to-device copy achieves on this GPU (bytes read + written per second), the ceiling for any
kernel that reads its input once and writes as much: python tools/copy_bw.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
for mb in (67, 134, 268, 1074):
    n = mb * 1000 * 1000 // 8
    srcs = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(4)]
    dst = torch.empty(n, dtype=torch.float64, device="cuda")
    for s in srcs:
        dst.copy_(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for s in srcs:
            dst.copy_(s)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / len(srcs))
    print("copy of %4d MB: %.1f us -> %.2f TB/s read+written" % (mb, best * 1e3, 2 * n * 8 / best / 1e9), flush=True)
