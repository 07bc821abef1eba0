"""Time lu_decomp / qr_decomp of one N x N matrix on the device (median of 10, HIP events). usage: python tools/time_lu.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import _median_ms


def main():
    from nd4js_amd import _lib, dev
    h = _lib.handle(0)
    for N in [int(x) for x in sys.argv[1:]] or [2048]:
        A = dev.fill_uniform(7, (N, N))
        for name in ("lu_decomp", "qr_decomp"):
            fn = getattr(dev, name)
            ms, lo, hi = _median_ms(lambda: fn(A), h, reps=10, warm=3)
            print("%s N=%d median %.3f ms min %.3f max %.3f" % (name, N, ms, lo, hi), flush=True)


if __name__ == "__main__":
    main()
