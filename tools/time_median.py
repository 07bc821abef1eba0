"""Median-of-10 HIP-event timing of single ops (same protocol as bench_ops.py): python tools/time_median.py lu qr svd --n 2048"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_ops  # noqa: E402
from nd4js_amd import _lib, dev  # noqa: E402

args = sys.argv[1:]
ns = [int(x) for x in args[args.index("--n") + 1:]] if "--n" in args else [2048]
ops = [a for a in (args[:args.index("--n")] if "--n" in args else args)] or ["lu", "qr"]
h = _lib.handle()
for n in ns:
    A = dev.fill_uniform(7, (n, n))
    for op in ops:
        fn = {"lu": lambda: dev.lu_decomp(A), "qr": lambda: dev.qr_decomp(A), "svd": lambda: dev.svd_decomp(A),
              "hess": lambda: dev.hessenberg_decomp(A), "bidiag": lambda: dev.bidiag_decomp(A), "matmul": lambda: dev.matmul2(A, A)}[op]
        med, lo, hi = bench_ops._median_ms(fn, h, reps=10 if op not in ("hess", "bidiag") else 3)
        print(op, n, "median ms", round(med, 3), "min", round(lo, 3), "max", round(hi, 3), flush=True)
