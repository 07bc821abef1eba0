"""Random-shape sweep of the round-2 kernel paths against numpy / scipy properties (not a test: run once on the GPU box after a change):
python tools/fuzz_paths.py [seed] [cases]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.linalg as sl  # noqa: E402

from nd4js_amd import la  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rs = np.random.RandomState(seed)
eps = 2.0 ** -52
bad = 0


def check(name, ok, detail):
    global bad
    if not ok:
        bad += 1
        print("FAIL", name, detail, flush=True)


def dims(lo, hi, mult=None):
    n = int(rs.randint(lo, hi + 1))
    if mult and rs.rand() < 0.5:
        n = max(mult, (n // mult) * mult)
    return n


for it in range(cases):
    # ---- QR (look-ahead, Q^T accumulator, wide / tall / batched)
    b = int(rs.choice([1, 1, 1, 2, 3, 5]))
    M, N = dims(60, int(os.environ.get("FUZZ_MAX", 700)), 16), dims(1, int(os.environ.get("FUZZ_MAX", 700)), 16)
    a = rs.standard_normal((b, M, N)) if b > 1 else rs.standard_normal((M, N))
    q, r = la.qr_decomp(a)
    L = min(M, N)
    check("qr recon", np.abs(q @ r - a).max() <= 64 * eps * max(M, N) * np.abs(a).max(), (b, M, N))
    check("qr orth", np.abs(np.swapaxes(q, -1, -2) @ q - np.eye(L)).max() <= 64 * eps * max(M, 8), (b, M, N))
    check("qr triu", np.array_equal(r, np.triu(r)), (b, M, N))
    if rs.rand() < 0.4:
        qf, rf = la.qr_decomp_full(a)
        check("qr_full recon", np.abs(qf @ rf - a).max() <= 64 * eps * max(M, N) * np.abs(a).max(), (b, M, N))
        check("qr_full orth", np.abs(qf @ np.swapaxes(qf, -1, -2) - np.eye(M)).max() <= 64 * eps * M, (b, M, N))
    # ---- LU (look-ahead) + solve (one-launch triangular solves)
    n = dims(60, 900, 32)
    a = rs.standard_normal((b, n, n)) if b > 1 else rs.standard_normal((n, n))
    lu, p = la.lu_decomp(a)
    for k in range(b):
        ak, luk, pk = (a[k], lu[k], p[k]) if b > 1 else (a, lu, p)
        lo = np.tril(luk, -1) + np.eye(n)
        up = np.triu(luk)
        check("lu recon", np.abs(lo @ up - ak[pk]).max() <= 256 * eps * n * np.abs(ak).max(), (b, n))
        check("lu |l|<=1", np.abs(lo).max() <= 1.0 + 1e-15, (b, n))
        check("lu perm", sorted(pk.tolist()) == list(range(n)), (b, n))
    j = dims(1, 300)
    y = rs.standard_normal((n, j))
    x = la.lu_solve(lu, p, y)
    cond = np.linalg.cond(a).max()
    check("lu_solve", np.abs(a @ x - y).max() <= 1e-13 * max(cond, 10) * n, (b, n, j, cond))
    # ---- Cholesky (look-ahead, fused trsm) + solve
    n = dims(60, 800, 32)
    g = rs.standard_normal((n, n))
    s = g @ g.T + n * np.eye(n)
    lc = la.cholesky_decomp(s)
    check("chol", np.abs(lc @ lc.T - s).max() <= 64 * eps * n * np.abs(s).max() and np.array_equal(lc, np.tril(lc)), n)
    y = rs.standard_normal((n, j))
    x = la.cholesky_solve(lc, y)
    check("chol_solve", np.abs(s @ x - y).max() <= 1e-12 * n, (n, j))
    # ---- triangular solves
    n = dims(32, 1300, 32)
    t = np.tril(rs.standard_normal((n, n))) * 0.25 + np.diag(2.0 + rs.rand(n))
    y = rs.standard_normal((n, j))
    check("tril_solve", np.abs(t @ la.tril_solve(t, y) - y).max() <= 1e-12 * n, (n, j))
    check("triu_solve", np.abs(t.T @ la.triu_solve(t.T.copy(), y) - y).max() <= 1e-12 * n, (n, j))
    # ---- bidiagonalisation / Hessenberg (fused / blocked for one large matrix)
    if it % 4 == 0:
        M, N = dims(100, 600), dims(100, 600)
        a = rs.standard_normal((M, N))
        u, bmat, v = la.bidiag_decomp(a)
        check("bidiag recon", np.abs(u @ bmat @ v - a).max() <= 256 * eps * max(M, N) * np.abs(a).max(), (M, N))
        n = dims(100, 700, 2)
        a = rs.standard_normal((n, n))
        uh, hh = la.hessenberg_decomp(a)
        check("hess recon", np.abs(uh @ hh @ uh.T - a).max() <= 256 * eps * n * np.abs(a).max() and np.abs(np.tril(hh, -2)).max() == 0.0, n)
    print("case", it, "ok" if bad == 0 else f"{bad} failures so far", flush=True)
print("FAILURES" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
