"""Input families of the golden fixtures — numpy twin of oracle/gen_golden.js `applyFamily`.

They mirror the reference tests' own input families (qr_test.js:67-146, lu_test.js:82-94,
_generic_test_svd_decomp.js:180-336): dense, ~10 % zeros, zero row / column, rank deficient,
diagonal, upper triangular."""
import numpy as np

from nd4js_amd.rng import fill_uniform, hash_idx


def apply_family(fam, a, seed):
    M, N = a.shape
    flat = a.reshape(-1)
    if fam == "dense":
        pass
    elif fam == "sparse10":
        z = hash_idx(seed + 77, np.arange(flat.size), 10) == 0
        flat[z] = 0.0
    elif fam == "zerorow":
        a[int(hash_idx(seed + 78, 0, M)), :] = 0.0
    elif fam == "zerocol":
        a[:, int(hash_idx(seed + 79, 0, N))] = 0.0
    elif fam == "rankdef":
        rank = max(1, min(M, N) >> 1)
        for i in range(rank, M):
            a[i, :] = 0.5 * a[(i - rank) % rank, :] - 0.25 * a[(i + 1) % rank, :]
    elif fam == "diag":
        a[...] = np.where(np.eye(M, N, dtype=bool), a, 0.0)
    elif fam == "triu":
        a[...] = np.triu(a)
    else:
        raise ValueError(fam)
    return a


def make_input(seed, shape, family="dense"):
    shape = tuple(shape)
    a = fill_uniform(seed, int(np.prod(shape))).reshape((-1,) + shape[-2:])
    for b in range(a.shape[0]):
        apply_family(family, a[b], seed + b)
    return a.reshape(shape)


def triangle(seed, shape, upper):
    """numpy twin of gen_golden.js `triangle`: off-diagonal / 4, |diag| in [2, 3) -> well conditioned."""
    shape = tuple(shape)
    M = shape[-1]
    a = fill_uniform(seed, int(np.prod(shape))).reshape((-1, M, M))
    for b in range(a.shape[0]):
        d = np.diag(a[b]).copy()
        d = d + np.where(d >= 0, 2.0, -2.0)
        t = (np.triu(a[b], 1) if upper else np.tril(a[b], -1)) * 0.25
        a[b] = t + np.diag(d)
    return a.reshape(shape)


def spd(seed, shape):
    """S = B B^T + N I from the seeded B (oracle/gen_golden.js `chol`): symmetric positive definite, cond ~ 2."""
    import oracle
    from nd4js_amd import rng
    B = rng.matrix(seed, *shape)
    N = shape[-1]
    S = oracle.matmul2(B, np.swapaxes(B, -1, -2).copy())
    S[..., np.arange(N), np.arange(N)] += N
    return S


def sym_indefinite(seed, shape):
    """S = L D L^T from a seeded unit-lower L (entries / 4) and D_i = +-(1 + |u|) (oracle/gen_golden.js `ldl`)."""
    import oracle
    from nd4js_amd import rng
    r = rng.matrix(seed, *shape)
    N = shape[-1]
    idx = np.arange(N)
    L = np.tril(r * 0.25, -1)
    L[..., idx, idx] = 1.0
    D = np.zeros_like(r)
    d = r[..., idx, idx]
    D[..., idx, idx] = np.where(d >= 0, 1 + d, -1 + d)
    return oracle.matmul2(oracle.matmul2(L, D), np.swapaxes(L, -1, -2).copy())


def hess_input(seed, shape, family):
    """inputs of oracle/gen_golden.js `hess`"""
    from nd4js_amd import rng
    a = rng.matrix(seed, *shape)
    N = shape[-1]
    if family == "sparse":
        a[rng.matrix(seed + 1000, *shape) > 0.6] = 0.0
    if family == "hess":
        a = a * (np.arange(N)[None, :] + 1 >= np.arange(N)[:, None])       # zero below the sub-diagonal
    return a
