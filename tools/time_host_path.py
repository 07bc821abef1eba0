"""End-to-end timing of the host-pointer C ABI (H2D + kernel + D2H): python tools/time_host_path.py [N]
Two figures per op where it matters: with a FRESH result array per call (what the JS API hands out: the first write to new
pages costs 8-15 ms per 128 MiB in page faults, tools/pcie_fresh.hip) and with a reused, already faulted-in result array."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nd4js_amd import la, rng
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a, b = rng.matrix(5, N, N), rng.matrix(6, N, N)
c = np.empty((N, N))
la.matmul2(a, b, out=c)
s2 = a[:2048, :2048].copy()
for name, fn in (("matmul2 (fresh result array)", lambda: la.matmul2(a, b)), ("matmul2 (reused result array)", lambda: la.matmul2(a, b, out=c)),
                 ("lu_decomp 2048 (fresh results)", lambda: la.lu_decomp(s2)), ("qr_decomp 2048 (fresh results)", lambda: la.qr_decomp(s2))):
    fn()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    print(name, "host path ms: median", round(sorted(ts)[2] * 1e3, 2), "min", round(min(ts) * 1e3, 2), flush=True)
