"""GPU parity of the solve-side consumers: lu_solve (lu.js:84-177), tril_solve / triu_solve
(tri.js:155-290) through the C ABI, against the reference-generated golden vectors and the oracle."""
import numpy as np
import pytest

import oracle
from conftest import golden_cases
from families import triangle
from nd4js_amd import rng

pytestmark = pytest.mark.gpu


def relerr(x, ref):
    return np.linalg.norm((x - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1e-300)


@pytest.fixture(scope="module")
def la():
    from nd4js_amd import la as _la
    return _la


@pytest.mark.parametrize("name", golden_cases(op="lu_solve"))
def test_lu_solve_golden(la, golden, name):
    g = golden(name)
    a = rng.matrix(g.seedA, *g.shapeA)
    y = rng.matrix(g.seedY, *g.shapeY)
    x = la.lu_solve(la.lu_decomp(a), y)                 # the full decomp -> solve chain on the GPU
    ref = g["X"]
    assert x.shape == ref.shape
    cond = np.linalg.cond(a).max()
    assert relerr(x, ref) <= 1e-13 * max(cond, 10)
    assert np.abs(a @ x - np.broadcast_to(y, x.shape)).max() <= 1e-12 * max(cond, 10)


@pytest.mark.parametrize("name", golden_cases(op="triu_solve") + golden_cases(op="tril_solve"))
def test_tri_solve_golden(la, golden, name):
    g = golden(name)
    upper = g.op == "triu_solve"
    t = triangle(g.seedT, g.shapeT, upper)
    y = rng.matrix(g.seedY, *g.shapeY)
    x = (la.triu_solve if upper else la.tril_solve)(t, y)
    assert x.shape == g["X"].shape and relerr(x, g["X"]) <= 1e-13


@pytest.mark.parametrize("N,J", [(1, 1), (5, 3), (31, 1), (32, 32), (33, 7), (100, 257), (500, 64), (1024, 1024), (1100, 3)])
def test_lu_solve_vs_oracle(la, N, J):
    a = rng.matrix(3000 + N, N, N)
    y = rng.matrix(3100 + J, N, J)
    lu, p = oracle.lu_decomp(a)
    x = la.lu_solve(lu, p, y)
    ref = oracle.lu_solve(lu, p, y)
    cond = np.linalg.cond(a)
    assert relerr(x, ref) <= 1e-14 * max(cond, 10)


def test_tri_only_reads_its_triangle(la):
    t = triangle(3200, (64, 64), True)
    y = rng.matrix(3201, 64, 5)
    junk = t + np.tril(rng.matrix(3202, 64, 64), -1) * 100.0     # garbage below the diagonal must be ignored
    assert np.array_equal(la.triu_solve(junk, y), la.triu_solve(t, y))


def test_solve_errors_match_reference_text(la):
    with pytest.raises(ValueError, match="LU and y don't match"):
        la.lu_solve(np.eye(3), np.arange(3), np.ones((4, 1)))
    with pytest.raises(ValueError, match="must be quadratic"):
        la.triu_solve(np.ones((3, 4)), np.ones((3, 1)))
    with pytest.raises(ValueError, match="don't match"):
        la.tril_solve(np.eye(3), np.ones((4, 1)))


def test_lu_solve_2048(la):
    """C3 size end to end: decomp + solve, residual against A."""
    N = 2048
    a = rng.matrix(7, N, N)
    y = rng.matrix(3300, N, 8)
    x = la.lu_solve(la.lu_decomp(a), y)
    assert np.abs(a @ x - y).max() <= 1e-9


# ---- least squares from a factorisation: qr_lstsq (qr.js:186-273), svd_lstsq / svd_solve (svd.js:66-228) ----
@pytest.mark.parametrize("name", golden_cases(op="qr_lstsq"))
def test_qr_lstsq_golden(la, golden, name):
    """Same (Q,R) as the reference (the oracle's Givens QR is bit-identical) -> only the solve differs."""
    g = golden(name)
    a = rng.matrix(g.seedA, *g.shapeA)
    y = rng.matrix(g.seedY, *g.shapeY)
    q, r = (oracle.qr_decomp_full if g.full else oracle.qr_decomp)(a)
    x = la.qr_lstsq(q, r, y)
    ref = g["X"]
    assert x.shape == ref.shape
    assert relerr(x, ref) <= 1e-14 * max(np.linalg.cond(a).max(), 10)


@pytest.mark.parametrize("name", [c for c in golden_cases(op="qr_lstsq") if "full" not in c])
def test_qr_lstsq_chain_on_gpu(la, golden, name):
    """qr_decomp -> qr_lstsq both on the GPU (Householder Q,R differ from Givens by rounding only)."""
    g = golden(name)
    a = rng.matrix(g.seedA, *g.shapeA)
    x = la.qr_lstsq(la.qr_decomp(a), rng.matrix(g.seedY, *g.shapeY))
    assert relerr(x, g["X"]) <= 1e-13 * max(np.linalg.cond(a).max(), 10)


@pytest.mark.parametrize("name", golden_cases(op="svd_lstsq") + golden_cases(op="svd_solve"))
def test_svd_lstsq_golden(la, golden, name):
    g = golden(name)
    y = rng.matrix(g.seedY, *g.shapeY)
    fn = la.svd_solve if g.op == "svd_solve" else la.svd_lstsq
    x = fn(g["U"], g["sv"], g["V"], y)
    ref = g["X"]
    assert x.shape == ref.shape
    sv = g["sv"]
    rank = la.svd_rank(sv)
    cond = (sv[..., 0] / np.take_along_axis(sv, np.maximum(rank - 1, 0)[..., None], -1)[..., 0]).max()
    assert relerr(x, ref) <= 1e-14 * max(cond, 10)
    if g.rankDef:
        assert np.all(rank == sv.shape[-1] - 1)


@pytest.mark.parametrize("N,I,J", [(1, 1, 1), (40, 40, 1), (97, 33, 5), (300, 200, 130), (1024, 512, 64), (2048, 2048, 8)])
def test_qr_lstsq_vs_oracle_and_normal_equations(la, N, I, J):
    a = rng.matrix(3400 + N, N, I)
    y = rng.matrix(3500 + J, N, J)
    q, r = la.qr_decomp(a)
    x = la.qr_lstsq(q, r, y)
    cond = np.linalg.cond(a)
    if N <= 300:
        assert relerr(x, oracle.qr_lstsq(q, r, y)) <= 1e-14 * max(cond, 10)
    # least-squares optimality: A^T (A x - y) = 0
    assert np.abs(a.T @ (a @ x - y)).max() <= 1e-11 * max(cond, 10) * np.abs(y).max() * np.sqrt(N)


@pytest.mark.parametrize("N,I,J", [(64, 64, 3), (200, 120, 7), (120, 200, 7), (512, 512, 33)])
def test_svd_lstsq_chain_on_gpu(la, N, I, J):
    """svd_decomp -> svd_lstsq both on the GPU: minimum-norm least-squares solution == pinv(A) y."""
    a = rng.matrix(3600 + N, N, I)
    y = rng.matrix(3700 + J, N, J)
    x = la.svd_lstsq(la.svd_decomp(a), y)
    ref = np.linalg.pinv(a) @ y
    assert relerr(x, ref) <= 1e-13 * max(np.linalg.cond(a), 10)


def test_svd_lstsq_rank_cut_on_gpu(la):
    """Exactly rank-deficient system: components below sqrt(eps)*sv_0 are dropped (svd.js:165-177), not divided by."""
    b = rng.matrix(3800, 96, 40)
    a = b @ rng.matrix(3801, 40, 64)                       # 96 x 64, rank 40
    y = rng.matrix(3802, 96, 3)
    u, sv, v = la.svd_decomp(a)
    assert int(la.svd_rank(sv)) == 40
    x = la.svd_lstsq(u, sv, v, y)
    assert relerr(x, oracle.svd_lstsq(u, sv, v, y)) <= 1e-13
    assert relerr(x, np.linalg.pinv(a, rcond=1e-8) @ y) <= 1e-9


def test_lstsq_device_resident(la):
    import torch
    from nd4js_amd import dev
    a = rng.matrix(3900, 3, 80, 48)
    y = rng.matrix(3901, 3, 80, 6)
    ad, yd = torch.from_numpy(a).cuda(), torch.from_numpy(y).cuda()
    q, r = dev.qr_decomp(ad)
    x1 = dev.qr_lstsq(q, r, yd).cpu().numpy()
    u, sv, v = dev.svd_decomp(ad)
    x2 = dev.svd_lstsq(u, sv, v, yd).cpu().numpy()
    ref = np.linalg.pinv(a) @ y
    assert relerr(x1, ref) <= 1e-12 and relerr(x2, ref) <= 1e-12


def test_lstsq_errors_match_reference_text(la):
    with pytest.raises(ValueError, match="Q and y don't match"):
        la.qr_lstsq(np.ones((4, 3)), np.ones((3, 3)), np.ones((5, 1)))
    with pytest.raises(ValueError, match="Q and R don't match"):
        la.qr_lstsq(np.ones((4, 3)), np.ones((2, 3)), np.ones((4, 1)))
    with pytest.raises(ValueError, match="Under-determined"):
        la.qr_lstsq(np.ones((3, 3)), np.ones((3, 5)), np.ones((3, 1)))
    with pytest.raises(ValueError, match="not broadcast-compatible"):
        la.qr_lstsq(np.ones((2, 4, 3)), np.ones((3, 3, 3)), np.ones((4, 1)))
    with pytest.raises(ValueError, match="U and sv don't match"):
        la.svd_lstsq(np.ones((4, 3)), np.ones(2), np.ones((3, 3)), np.ones((4, 1)))
    with pytest.raises(ValueError, match="NaN or Infinity"):
        la.svd_lstsq(np.eye(2), np.array([1.0, np.inf]), np.eye(2), np.ones((2, 1)))
    with pytest.raises(ValueError, match="System not square"):
        la.svd_solve(np.ones((4, 3)), np.ones(3), np.ones((3, 3)), np.ones((4, 1)))


@pytest.mark.parametrize("M,J", [(256, 32), (512, 40), (1056, 100), (2080, 33)])
def test_one_launch_triangular_solve(la, M, J):
    """M a multiple of 32 with >= 32 right-hand sides takes trsm.hip's column-local kernel (trsm_cols: X in MFMA accumulators, inverted
    32 x 32 diagonal blocks, panels of 1024 rows with a GEMM in between): both triangles against the oracle's substitution."""
    y = rng.matrix(3400 + J, M, J)
    for upper in (False, True):
        t = triangle(3410 + M, (M, M), upper)
        x = (la.triu_solve if upper else la.tril_solve)(t, y)
        ref = (oracle.triu_solve if upper else oracle.tril_solve)(t, y)
        assert relerr(x, ref) <= 1e-13
        assert np.abs(t @ x - y).max() <= 1e-12 * M


def test_one_launch_triangular_solve_batched_broadcast(la):
    """leading batch dimensions broadcast (tri.js:155-290): one triangle against a stack of right-hand sides and vice versa"""
    t = triangle(3420, (512, 512), False)
    y = rng.matrix(3421, 3, 512, 48)
    x = la.tril_solve(t, y)
    for b in range(3):
        assert relerr(x[b], oracle.tril_solve(t, y[b])) <= 1e-13
    ts = triangle(3422, (2, 256, 256), True)
    y1 = rng.matrix(3423, 256, 64)
    x = la.triu_solve(ts, y1)
    for b in range(2):
        assert relerr(x[b], oracle.triu_solve(ts[b], y1)) <= 1e-13
