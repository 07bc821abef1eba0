import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nd4js_amd import dev
cases = [(2048, 1), (512, 64), (512, 1024)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for n, b in cases:
    X = dev.fill_uniform(9, (b, n, n)) if b > 1 else dev.fill_uniform(9, (n, n))
    info = {}
    dev.svd_decomp(X, info=info); torch.cuda.synchronize()
    t = time.perf_counter(); dev.svd_decomp(X, info=info); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(n, b, "ms", round(dt * 1e3, 2), info, "nominal GF/s", round(21.0 * n ** 3 * b / dt / 1e9, 1), flush=True)
