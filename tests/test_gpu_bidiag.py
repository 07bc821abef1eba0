"""GPU parity of bidiag_decomp (bidiag.js:32-319, all three shape branches) through the C ABI, against
reference-generated goldens and the oracle; properties as in bidiag_test.js (A = U B V, orthogonality, bidiagonal form)."""
import numpy as np
import pytest

import oracle
from conftest import golden_cases
from nd4js_amd import rng

pytestmark = pytest.mark.gpu
EPS = 2.0 ** -52


@pytest.fixture(scope="module")
def la():
    from nd4js_amd import la as _la
    return _la


def bidiag_input(g):
    a = rng.matrix(g.seed, *g.shape)
    if g.sparse:
        a[rng.matrix(g.seed + 1000, *g.shape) > 0.6] = 0.0
    return a


def check_props(a, u, b, v):
    M, N = a.shape[-2:]
    I = min(M, N)
    J = I if M >= N else I + 1
    assert u.shape[-2:] == (M, I) and b.shape[-2:] == (I, J) and v.shape[-2:] == (J, N)
    mask = np.zeros((I, J), dtype=bool)
    idx = np.arange(I)
    mask[idx, idx] = True
    mask[idx[idx + 1 < J], idx[idx + 1 < J] + 1] = True
    assert np.all(b[..., ~mask] == 0)                                                    # exact zeros off the two diagonals
    n = max(M, N, 4)
    assert np.abs(np.swapaxes(u, -1, -2) @ u - np.eye(I)).max() <= 8 * EPS * n
    assert np.abs(v @ np.swapaxes(v, -1, -2) - np.eye(J)).max() <= 8 * EPS * n
    scale = max(np.linalg.norm(a), 1e-300)
    assert np.linalg.norm(u @ b @ v - a) <= 32 * EPS * n * scale


@pytest.mark.parametrize("name", golden_cases(op="bidiag_decomp"))
def test_bidiag_golden(la, golden, name):
    g = golden(name)
    a = bidiag_input(g)
    u, b, v = la.bidiag_decomp(a)
    check_props(a, u, b, v)
    n = max(a.shape[-2:])
    tol = 256 * EPS * max(n, 4)
    amax = max(np.abs(a).max(), 1e-300)
    if not g.sparse:                                 # exact zeros in the input make some signs a coin toss of the rounding
        assert np.abs(b - g["B"]).max() <= tol * amax * max(n, 4) ** 0.5
        assert np.abs(u - g["U"]).max() <= tol * 8 and np.abs(v - g["V"]).max() <= tol * 8
    else:
        assert np.abs(np.abs(b) - np.abs(g["B"])).max() <= tol * amax * max(n, 4) ** 0.5


@pytest.mark.parametrize("M,N", [(1, 1), (2, 2), (3, 3), (33, 33), (64, 64), (200, 200), (512, 512),
                                 (5, 1), (9, 2), (70, 33), (300, 120), (1, 4), (2, 9), (33, 70), (120, 300), (63, 64), (64, 63),
                                 (128, 128), (129, 257), (257, 129), (300, 299), (299, 300), (400, 200), (200, 400)])
def test_bidiag_shapes_vs_oracle(la, M, N):
    a = rng.matrix(6400 + 7 * M + N, M, N)
    u, b, v = la.bidiag_decomp(a)
    check_props(a, u, b, v)
    if M * N <= 300 * 300:
        uo, bo, vo = oracle.bidiag_decomp(a)
        n = max(M, N)
        assert np.abs(b - bo).max() <= 1e-11 * n and np.abs(u - uo).max() <= 1e-10 and np.abs(v - vo).max() <= 1e-10
    sv = np.linalg.svd(b, compute_uv=False)
    assert np.abs(sv - np.linalg.svd(a, compute_uv=False)[: len(sv)]).max() <= 1e-11 * max(np.abs(a).max() * max(M, N), 1)


def test_bidiag_batch_device_and_errors(la):
    import torch
    from nd4js_amd import dev
    a = rng.matrix(6500, 2, 3, 20, 31)
    u, b, v = la.bidiag_decomp(a)
    check_props(a, u, b, v)
    ud, bd, vd = dev.bidiag_decomp(torch.from_numpy(a).cuda())
    assert np.array_equal(ud.cpu().numpy(), u) and np.array_equal(bd.cpu().numpy(), b) and np.array_equal(vd.cpu().numpy(), v)
    with pytest.raises(ValueError, match="at least 2D"):
        la.bidiag_decomp(np.ones(3))
    with pytest.raises(ValueError, match="complex A not yet supported"):
        la.bidiag_decomp(np.ones((2, 2), dtype=np.complex128))


@pytest.mark.parametrize("M,N", [(1024, 1024), (2048, 2048), (3000, 2100), (4096, 3072), (2100, 3000)])
def test_fused_path_large(la, M, N):
    """VERDICT r2 #7 / ADVICE r2: the fused two-launch path (bidiag.hip: bd2_colpass / bd2_rowpass) is enabled up to 4096 x 3072 but was
    only tested to 512^2 (2048^2 was benchmarked, never checked): its reference properties (bidiag_test.js) at the benchmark size and at
    the caps — where bd2_colpass runs 1024 threads with its largest dynamic LDS — and the values against the oracle at 1024^2."""
    a = rng.matrix(7400 + M + N, M, N)
    u, b, v = la.bidiag_decomp(a)
    check_props(a, u, b, v)
    if max(M, N) <= 1024:
        uo, bo, vo = oracle.bidiag_decomp(a)
        n = max(M, N)
        tol = 256 * EPS * n
        assert np.abs(b - bo).max() <= tol * np.abs(a).max() * n ** 0.5
        assert np.abs(u - uo).max() <= tol * 8 and np.abs(v - vo).max() <= tol * 8


@pytest.mark.parametrize("M,N", [(2048, 130), (130, 2048), (1025, 1023), (1023, 1025), (2047, 2047), (513, 512)])
def test_one_launch_reduction_shapes(la, M, N):
    """one matrix with 128 <= M, N <= 2048 is reduced by ONE launch (bidiag.hip: bdp) — 16 x 16 workgroups hold the
    matrix as tiles in registers, four rounds of tagged words per step. Shapes far from square (most tiles empty), odd extents, both
    M > N and M < N: the reference's properties (bidiag_test.js), and the singular values against LAPACK's."""
    a = rng.matrix(7600 + M + N, M, N)
    u, b, v = la.bidiag_decomp(a)
    check_props(a, u, b, v)
    sv = np.linalg.svd(b, compute_uv=False)
    assert np.abs(sv - np.linalg.svd(a, compute_uv=False)[: len(sv)]).max() <= 1e-11 * np.abs(a).max() * max(M, N)


@pytest.mark.parametrize("M,N", [(512, 512), (300, 299), (257, 129)])
def test_fused_path_below_the_one_launch_cap(la, M, N, monkeypatch):
    """The fused two-launch path (what shapes beyond 2048 take) on shapes the one-launch reduction would take: same checks."""
    monkeypatch.setenv("ND4HIP_BIDIAG_NO_PERSIST", "1")
    a = rng.matrix(7700 + M + N, M, N)
    u, b, v = la.bidiag_decomp(a)
    check_props(a, u, b, v)
    if M * N <= 300 * 300:
        uo, bo, vo = oracle.bidiag_decomp(a)
        assert np.abs(b - bo).max() <= 1e-11 * max(M, N) and np.abs(u - uo).max() <= 1e-10 and np.abs(v - vo).max() <= 1e-10


def test_one_launch_reduction_special_inputs(la):
    """As tests/test_gpu_hess.py::test_one_launch_reduction_special_inputs, for bdp: zeros, identity, 1e150 / 1e-150 / 1e-290 scales,
    rank 8, zero rows and columns, an already bidiagonal input, a graded matrix."""
    N = 256
    base = rng.matrix(7800, N, N)
    zc = base.copy()
    zc[:, 5] = 0.0
    zc[17, :] = 0.0
    zc[:, 200:210] = 0.0
    cases = {"zeros": np.zeros((N, N)), "identity": np.eye(N), "1e150": base * 1e150, "1e-150": base * 1e-150, "1e-290": base * 1e-290,
             "lowrank": rng.matrix(7801, N, 8) @ rng.matrix(7802, 8, N), "zero rows and columns": zc,
             "upper bidiagonal": np.triu(np.tril(base, 1)), "graded": base * np.logspace(0, -12, N)[:, None]}
    for name, a in cases.items():
        sc = max(np.abs(a).max(), 1e-300)
        u, b, v = la.bidiag_decomp(a)
        assert np.isfinite(b).all() and np.isfinite(u).all() and np.isfinite(v).all(), name
        assert np.abs(u @ b @ v - a).max() <= 256 * EPS * N * sc, name
        assert np.abs(u.T @ u - np.eye(N)).max() <= 16 * EPS * N and np.abs(v @ v.T - np.eye(N)).max() <= 16 * EPS * N, name
        assert np.abs(np.tril(b, -1)).max() == 0.0 and np.abs(np.triu(b, 2)).max() == 0.0, name
