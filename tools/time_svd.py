"""Time svd_decomp of one N x N matrix on the device (median of 5, HIP events). usage: python tools/time_svd.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_ops import _median_ms


def main():
    from nd4js_amd import _lib, dev
    h = _lib.handle(0)
    for N in [int(x) for x in sys.argv[1:]] or [2048]:
        A = dev.fill_uniform(7, (N, N))
        info = {}
        ms, lo, hi = _median_ms(lambda: dev.svd_decomp(A, info), h, reps=5, warm=1)
        print("svd_decomp N=%d median %.2f ms min %.2f max %.2f sweeps %s" % (N, ms, lo, hi, info.get("sweeps")), flush=True)


if __name__ == "__main__":
    main()
