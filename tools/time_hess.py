"""Time hessenberg_decomp / bidiag_decomp of one N x N matrix on the device (median of 5, HIP events on the handle's stream).
usage: python tools/time_hess.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_ops import _median_ms


def main():
    from nd4js_amd import _lib, dev
    h = _lib.handle(0)
    sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048]
    for N in sizes:
        A = dev.fill_uniform(7, (N, N))
        for name in ("hessenberg_decomp", "bidiag_decomp"):
            fn = getattr(dev, name)
            ms, lo, hi = _median_ms(lambda: fn(A), h, reps=5, warm=1)
            print("%s N=%d median %.3f ms min %.3f max %.3f" % (name, N, ms, lo, hi), flush=True)


if __name__ == "__main__":
    main()
