import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from nd4js_amd import la, rng
a = rng.matrix(1000, 512, 512)
info = {}
u, sv, v = la.svd_decomp(a, info=info)
d = np.diag(u.T @ u) - 1
print(os.environ.get("ND4HIP_SVD_NOBLOCK"), info, "signed diag err: mean %.3e min %.3e max %.3e" % (d.mean(), d.min(), d.max()))
# MFMA rounding bias probe: positive operands, long-double reference
x = np.abs(rng.matrix(1, 128, 4096)); y = np.abs(rng.matrix(2, 4096, 128))
c = la.matmul2(x, y)
ref = (x.astype(np.longdouble) @ y.astype(np.longdouble))
rel = ((c.astype(np.longdouble) - ref) / ref).astype(np.float64)
cn = x @ y
reln = ((cn.astype(np.longdouble) - ref) / ref).astype(np.float64)
print("gpu gemm signed rel err mean %.3e std %.3e | numpy mean %.3e std %.3e" % (rel.mean(), rel.std(), reln.mean(), reln.std()))
