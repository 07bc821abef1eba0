"""GPU parity of hessenberg_decomp (hessenberg.js:27-115) through the C ABI, against reference-generated
goldens and the oracle; properties as in hessenberg_test.js (A = U H U^T, U orthogonal, H upper Hessenberg)."""
import numpy as np
import pytest

import oracle
from conftest import golden_cases
from families import hess_input
from nd4js_amd import rng

pytestmark = pytest.mark.gpu
EPS = 2.0 ** -52


@pytest.fixture(scope="module")
def la():
    from nd4js_amd import la as _la
    return _la


def check_props(a, u, h):
    N = a.shape[-1]
    assert np.array_equal(np.tril(h, -2), np.zeros_like(h))                             # exact zeros below the sub-diagonal
    assert np.abs(u @ np.swapaxes(u, -1, -2) - np.eye(N)).max() <= 8 * EPS * max(N, 4)
    scale = max(np.linalg.norm(a), 1e-300)
    assert np.linalg.norm(u @ h @ np.swapaxes(u, -1, -2) - a) <= 32 * EPS * max(N, 4) * scale
    if N > 1:                                                                            # hessenberg.js:88-89
        assert np.all(u[..., -1, :-1] == 0) and np.all(u[..., :-1, -1] == 0) and np.all(u[..., -1, -1] == 1)


@pytest.mark.parametrize("name", golden_cases(op="hessenberg_decomp"))
def test_hessenberg_golden(la, golden, name):
    g = golden(name)
    a = hess_input(g.seed, tuple(g.shape), g.family)
    u, h = la.hessenberg_decomp(a)
    assert u.shape == g["U"].shape and h.shape == g["H"].shape
    check_props(a, u, h)
    N = a.shape[-1]
    tol = 64 * EPS * max(N, 4)
    assert np.abs(h - g["H"]).max() <= tol * max(np.abs(a).max(), 1e-300) * max(N, 4) ** 0.5
    assert np.abs(u - g["U"]).max() <= tol * 4
    if g.family == "hess":                                                               # every step skipped: nothing changes
        assert np.array_equal(h, a) and np.array_equal(u, np.eye(N))


@pytest.mark.parametrize("N", [1, 2, 3, 4, 31, 32, 33, 64, 200, 257, 512, 1024])
def test_hessenberg_sizes(la, N):
    a = rng.matrix(6200 + N, N, N)
    u, h = la.hessenberg_decomp(a)
    check_props(a, u, h)
    if N <= 257:
        uo, ho = oracle.hessenberg_decomp(a)
        assert np.abs(h - ho).max() <= 1e-11 and np.abs(u - uo).max() <= 1e-11
    ev, evr = np.sort_complex(np.linalg.eigvals(h)), np.sort_complex(np.linalg.eigvals(a))   # similarity keeps the spectrum
    if N <= 257:
        assert np.abs(ev - evr).max() <= 1e-8 * max(np.abs(evr).max(), 1)


def test_hessenberg_batch_and_device(la):
    import torch
    from nd4js_amd import dev
    a = rng.matrix(6300, 2, 3, 48, 48)
    u, h = la.hessenberg_decomp(a)
    check_props(a, u, h)
    ud, hd = dev.hessenberg_decomp(torch.from_numpy(a).cuda())
    assert np.array_equal(ud.cpu().numpy(), u) and np.array_equal(hd.cpu().numpy(), h)
    with pytest.raises(ValueError, match="must be square"):
        la.hessenberg_decomp(np.ones((2, 3)))
    with pytest.raises(ValueError, match="at least be 2D"):
        la.hessenberg_decomp(np.ones(3))


@pytest.mark.parametrize("N", [512, 600])
def test_blocked_path_vs_oracle(la, N, monkeypatch):
    """One matrix with even N >= 512 beyond the one-launch reduction (N > 2048, or ND4HIP_HESS_NO_PERSIST as here) takes the blocked
    two-launch step (hess.hip: hessb_pass / hessb_reduce): values against the oracle, which is bit-identical to the reference on the
    goldens."""
    monkeypatch.setenv("ND4HIP_HESS_NO_PERSIST", "1")
    a = rng.matrix(6400 + N, N, N)
    u, h = la.hessenberg_decomp(a)
    check_props(a, u, h)
    uo, ho = oracle.hessenberg_decomp(a)
    eps = 2.0 ** -52
    assert np.abs(h - ho).max() <= 64 * eps * N * np.abs(a).max() * N ** 0.5
    assert np.abs(u - uo).max() <= 64 * eps * N


@pytest.mark.parametrize("persist", [True, False])
def test_blocked_path_skipped_steps(la, persist, monkeypatch):
    """Rows that are already in Hessenberg form are skipped (hessenberg.js:46): all of them (the input comes back bit-identical, U = I),
    and a mix of skipped and reflected steps inside one block of 32 — in the one-launch reduction (a skipped step still runs its two
    exchange rounds) and in the blocked path."""
    if not persist:
        monkeypatch.setenv("ND4HIP_HESS_NO_PERSIST", "1")
    N = 512
    hess = np.triu(rng.matrix(6500, N, N), -1)
    u, h = la.hessenberg_decomp(hess)
    assert np.array_equal(h, hess) and np.array_equal(u, np.eye(N))
    mixed = hess.copy()
    rows = [N - 3, N - 4, N - 20, N - 33, N - 34, 300, 40]
    for r in rows:
        mixed[r, : r - 1] = rng.matrix(6501 + r, 1, N)[0, : r - 1]
    u, h = la.hessenberg_decomp(mixed)
    check_props(mixed, u, h)
    # values: only where the comparison is well conditioned. Rounding differences grow from row to row on this input (a Hessenberg
    # form is not forward stable; the unblocked kernels drift from the oracle in the same way): the rows finished in the first two
    # blocks (where skipped and reflected steps alternate) agree to rounding, the factorisation as a whole by its properties above.
    uo, ho = oracle.hessenberg_decomp(mixed)
    lo = N - 48
    assert np.abs(h[lo:] - ho[lo:]).max() <= 1e-11 * np.abs(mixed).max()
    assert np.abs(u[:, lo:] - uo[:, lo:]).max() <= 1e-11


@pytest.mark.parametrize("N", [1024, 2048])
def test_blocked_path_large(la, N):
    """the blocked path at the benchmark size (2048^2: properties of hessenberg_test.js) and against the oracle at 1024^2
    (the largest tested size used to be 600)."""
    a = rng.matrix(6600 + N, N, N)
    u, h = la.hessenberg_decomp(a)
    check_props(a, u, h)
    if N <= 1024:
        uo, ho = oracle.hessenberg_decomp(a)
        assert np.abs(h - ho).max() <= 64 * EPS * N * np.abs(a).max() * N ** 0.5
        assert np.abs(u - uo).max() <= 64 * EPS * N


@pytest.mark.parametrize("N", [128, 129, 255, 513, 1025, 2047])
def test_one_launch_reduction_sizes(la, N):
    """128 <= N <= 2048 (one matrix) is reduced by ONE launch (hess.hip: hessp) — 16 x 16 workgroups hold H as tiles
    in registers, two rounds of tagged words per step. Sizes that do not fill the tiles (zero padding), odd N, the three tile edges
    (32, 64, 128): properties of hessenberg_test.js, and the values against the oracle where it is quick."""
    a = rng.matrix(6700 + N, N, N)
    u, h = la.hessenberg_decomp(a)
    check_props(a, u, h)
    if N <= 513:
        uo, ho = oracle.hessenberg_decomp(a)
        assert np.abs(h - ho).max() <= 64 * EPS * N * np.abs(a).max() * N ** 0.5
        assert np.abs(u - uo).max() <= 64 * EPS * N


def test_blocked_path_beyond_the_one_launch_reduction(la):
    """N > 2048 no longer fits the registers of 256 workgroups: the blocked path, by its properties."""
    N = 2304
    a = rng.matrix(6800, N, N)
    u, h = la.hessenberg_decomp(a)
    check_props(a, u, h)


def _special_inputs(N, seed):
    base = rng.matrix(seed, N, N)
    zc = base.copy()
    zc[:, 5] = 0.0
    zc[17, :] = 0.0
    zc[:, 200:210] = 0.0
    return {"zeros": np.zeros((N, N)), "identity": np.eye(N), "1e150": base * 1e150, "1e-150": base * 1e-150, "1e-290": base * 1e-290,
            "lowrank": rng.matrix(seed + 1, N, 8) @ rng.matrix(seed + 2, 8, N), "zero rows and columns": zc,
            "upper hessenberg": np.triu(base, -1), "graded": base * np.logspace(0, -12, N)[:, None]}


def test_one_launch_reduction_special_inputs(la):
    """The one-launch reduction scales every norm by the largest entry it sees (per 8-column group, then across the workgroups) and
    uses few-ulp reciprocals in the reflector's scalars: no overflow at 1e150, no underflow at 1e-290, zero rows are skipped
    (hessenberg.js:46), rank-deficient and graded inputs keep the reference's properties relative to the input's scale."""
    N = 256
    for name, a in _special_inputs(N, 6900).items():
        sc = max(np.abs(a).max(), 1e-300)
        u, h = la.hessenberg_decomp(a)
        assert np.isfinite(h).all() and np.isfinite(u).all(), name
        assert np.abs(u @ h @ u.T - a).max() <= 256 * EPS * N * sc, name
        assert np.abs(u @ u.T - np.eye(N)).max() <= 16 * EPS * N, name
        assert np.abs(np.tril(h, -2)).max() == 0.0, name
