#!/usr/bin/env python3
"""Profiling driver: runs selected hot-path ops a few times so that
`rocprofv3 --kernel-trace --stats -- python3 tools/prof_ops.py lu qr` gives per-kernel times.
usage: prof_ops.py [matmul] [lu] [qr] [chol] [svd] [svdbatch] [lusolve] [--n N] [--reps R]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from nd4js_amd import dev  # noqa: E402


def main():
    args = sys.argv[1:]
    n = int(args[args.index("--n") + 1]) if "--n" in args else 2048
    reps = int(args[args.index("--reps") + 1]) if "--reps" in args else 3
    A = dev.fill_uniform(7, (n, n))
    if "chol" in args:
        S = dev.gemm_ex(False, True, 1.0, A, A, 0.0, torch.empty_like(A), n, n, n, n, n, n)
        S.diagonal().add_(float(n))
    for _ in range(reps):
        if "chol" in args:
            dev.cholesky_decomp(S)
        if "matmul" in args:
            dev.matmul2(A, A)
        if "lu" in args:
            dev.lu_decomp(A)
        if "qr" in args:
            dev.qr_decomp(A)
        if "svd" in args:
            dev.svd_decomp(A)
        if "lusolve" in args:
            if _ == 0:
                lu_p = dev.lu_decomp(A)
            dev.lu_solve(lu_p[0], lu_p[1], A)
        if "svdbatch" in args:
            X = dev.fill_uniform(1000, (32, 512, 512))
            dev.svd_decomp(X)
    torch.cuda.synchronize()
    print("done", args)


if __name__ == "__main__":
    main()
