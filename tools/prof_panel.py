#!/usr/bin/env python3
"""Profiling driver for the QR panel entry point (nd4hip_dgeqr2_panel_batched_dev): a few calls on distinct inputs per shape, so
that `rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE|WRITE_SIZE` see the batched panel kernels alone.
usage: prof_panel.py [panels rows]...   (default: 256 2048  2048 512)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from nd4js_amd import _lib, dev  # noqa: E402
h = _lib.handle(0)
args = [int(x) for x in sys.argv[1:]] or [256, 2048, 2048, 512]
for nb, rows in zip(args[0::2], args[1::2]):
    A = dev.fill_uniform(21, (nb, rows, 16))
    Ws = [A.clone() for _ in range(5)]
    V = torch.empty_like(A)
    T = torch.empty((nb, 16, 16), dtype=torch.float64, device="cuda")
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    for W in Ws:
        _lib.check(h.lib.nd4hip_dgeqr2_panel_batched_dev(h.ptr, nb, rows, 16, ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(V.data_ptr()),
                                                         ctypes.c_void_p(T.data_ptr())))
    torch.cuda.synchronize()
print("done", args)
