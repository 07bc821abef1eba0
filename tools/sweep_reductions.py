"""One-off sweep of the one-launch Hessenberg / bidiagonalisation over sizes around the tile edges: reference properties only.
usage: python tools/sweep_reductions.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nd4js_amd import la, rng

EPS = 2.0 ** -52
bad = 0
for N in [128, 129, 130, 191, 255, 256, 257, 384, 511, 512, 513, 700, 1023, 1024, 1025, 1536, 2046, 2047, 2048]:
    a = rng.matrix(9100 + N, N, N)
    u, h = la.hessenberg_decomp(a)
    r1 = np.abs(u @ h @ u.T - a).max() / (np.abs(a).max() * N)
    r2 = np.abs(u @ u.T - np.eye(N)).max()
    z = np.abs(np.tril(h, -2)).max()
    ok = r1 <= 64 * EPS and r2 <= 16 * EPS * N and z == 0.0 and np.isfinite(h).all()
    bad += not ok
    print("hess %5d  res %.2e  orth %.2e  below %.1e  %s" % (N, r1, r2, z, "ok" if ok else "BAD"), flush=True)
for M, N in [(128, 128), (128, 2048), (2048, 128), (129, 255), (255, 129), (512, 513), (513, 512), (1024, 640), (640, 1024), (1025, 1025),
             (1536, 2048), (2048, 1536), (2047, 2048), (2048, 2047), (2048, 2048), (300, 1999), (1999, 300)]:
    a = rng.matrix(9300 + M + 3 * N, M, N)
    u, b, v = la.bidiag_decomp(a)
    K = min(M, N)
    r1 = np.abs(u @ b @ v - a).max() / (np.abs(a).max() * max(M, N))
    r2 = max(np.abs(u.T @ u - np.eye(u.shape[1])).max(), np.abs(v @ v.T - np.eye(v.shape[0])).max())
    z = max(np.abs(np.tril(b, -1)).max(), np.abs(np.triu(b, 2)).max())
    ok = r1 <= 64 * EPS and r2 <= 16 * EPS * max(M, N) and z == 0.0 and np.isfinite(b).all()
    bad += not ok
    print("bidiag %5d x %5d  res %.2e  orth %.2e  off %.1e  %s" % (M, N, r1, r2, z, "ok" if ok else "BAD"), flush=True)
print("BAD:", bad)
sys.exit(1 if bad else 0)
