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
