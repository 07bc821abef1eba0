"""The host-pointer boundary (what the JS host binds): chunked / pipelined staging, the error path, several devices behind
one handle (nd4js_amd/csrc/nd4hip_host.hip)."""
import numpy as np
import pytest

import oracle
from nd4js_amd import _lib, la, rng

pytestmark = pytest.mark.gpu


def relerr(x, ref):
    return np.linalg.norm((x - ref).ravel()) / max(np.linalg.norm(ref.ravel()), 1e-300)


def test_pipelined_chunks_are_bit_identical_to_one_chunk():
    """a batch large enough to be cut into several chunks (64 MiB each) gives exactly what its members give on their own"""
    a = rng.matrix(6100, 96, 384, 384)                     # 96 x 1.1 MiB inputs, 3 outputs: several chunks
    lu, p = la.lu_decomp(a)
    q, r = la.qr_decomp(a)
    for k in (0, 31, 32, 63, 95):                          # chunk edges included (a lone matrix may take other kernels than a
        lu1, p1 = la.lu_decomp(a[k])                       # batch member: rounding-level differences only)
        q1, r1 = la.qr_decomp(a[k])
        assert relerr(lu[k], lu1) <= 1e-13 and np.array_equal(p[k], p1)
        assert relerr(q[k], q1) <= 1e-13 and relerr(r[k], r1) <= 1e-13
    b = rng.matrix(6101, 384, 200)
    c = la.matmul2(a, b)                                   # broadcast operand: uploaded once, used by every chunk
    for k in (0, 50, 95):
        assert relerr(c[k], a[k] @ b) <= 1e-13
    # the same batch in one chunk (a handle-level property: 24 of the 96 matrices stay below the chunk size) is bit-identical
    lu24, p24 = la.lu_decomp(a[:24])
    assert np.array_equal(p24, p[:24]) and relerr(lu24, lu[:24]) <= 1e-13


def test_single_large_matmul_is_pipelined_over_rows():
    a, b = rng.matrix(6110, 3000, 1500), rng.matrix(6111, 1500, 2100)
    c = la.matmul2(a, b)
    assert relerr(c, a @ b) <= 1e-13
    out = np.empty_like(c)
    assert la.matmul2(a, b, out=out) is out and np.array_equal(out, c)


def test_failure_in_a_later_chunk_leaves_the_handle_usable():
    """VERDICT r1 #8 / ADVICE: a mid-call failure (a matrix that is not positive definite in the third chunk) must not hand
    staging that is still in flight to the next call"""
    N, B = 512, 80
    x = rng.matrix(6120, B, N, 64)
    s = x @ np.swapaxes(x, -1, -2) + N * np.eye(N)
    bad = s.copy()
    bad[60, 100, 100] = -1.0
    for _ in range(3):
        with pytest.raises(ValueError, match="near\\) singular"):
            la.cholesky_decomp(bad)
        L = la.cholesky_decomp(s)                          # the very next call reuses the cached staging blocks
        assert relerr(L[7], oracle.cholesky_decomp(s[7])) <= 1e-13 and relerr(L[79], oracle.cholesky_decomp(s[79])) <= 1e-13
        a = rng.matrix(6121, 700, 300)
        assert relerr(la.matmul2(a, a.T.copy()), a @ a.T) <= 1e-13


def test_multi_device_handle():
    """nd4hip_create_multi: with one device it is the plain path; with more (when the box has them) the batch axis is sharded
    over the devices by host threads and the results are bit-identical to the single-device run."""
    n = _lib.load().nd4hip_device_count()
    h1 = _lib.Handle([0])
    assert h1.devices() == [0]
    a = rng.matrix(6130, 13, 96, 96)
    ref = la.svd_decomp(a)
    got = la.svd_decomp(a, device=h1)
    assert all(np.array_equal(x, y) for x, y in zip(got, ref))
    with pytest.raises(_lib.Nd4HipError):
        _lib.Handle([0, 0])
    if n >= 2:
        hn = _lib.Handle(list(range(n)))
        assert hn.devices() == list(range(n))
        info = {}
        got = la.svd_decomp(a, device=hn, info=info)
        assert all(np.array_equal(x, y) for x, y in zip(got, ref)) and info["sweeps"] >= 1 and info["rotations"] > 0
        lu, p = la.lu_decomp(a, device=hn)
        lu1, p1 = la.lu_decomp(a)
        assert np.array_equal(lu, lu1) and np.array_equal(p, p1)
        hn.close()
    h1.close()


def test_multi_device_code_runs_on_one_gpu(monkeypatch):
    """ND4HIP_TEST_ALLOW_DUP_DEVICES=1 lets nd4hip_create_multi take [0, 0, 0], so that the per-device host threads, the
    block partition (uneven: 13 = 5 + 4 + 4), the merged SVD audit (max sweeps / off-norm, rotation sum) and the first error with its
    device execute on the one-GPU box: results bit-identical to n_dev = 1 (units: svd_dc.js:918-925, lu.js:34-40), and a block that
    fails reports the device it ran on and leaves the handle usable."""
    monkeypatch.setenv("ND4HIP_TEST_ALLOW_DUP_DEVICES", "1")
    h3 = _lib.Handle([0, 0, 0])
    assert h3.devices() == [0, 0, 0]
    a = rng.matrix(6130, 13, 96, 96)
    info1, info3 = {}, {}
    ref = la.svd_decomp(a, info=info1)
    got = la.svd_decomp(a, device=h3, info=info3)
    assert all(np.array_equal(x, y) for x, y in zip(got, ref))
    assert info3["sweeps"] == info1["sweeps"] and info3["rotations"] == info1["rotations"] and info3["offnorm"] == info1["offnorm"]
    lu, p = la.lu_decomp(a, device=h3)
    lu1, p1 = la.lu_decomp(a)
    assert np.array_equal(lu, lu1) and np.array_equal(p, p1)
    q, r = la.qr_decomp(a, device=h3)
    q1, r1 = la.qr_decomp(a)                               # (blocks of <= 8 matrices take the multi-workgroup panels, the batch of 13
    assert relerr(q, q1) <= 1e-13 and relerr(r, r1) <= 1e-13   #  the thread-per-row ones: rounding-level differences, as between chunks)
    # fewer members than devices: the spare devices get nothing
    got2 = la.svd_decomp(a[:2], device=h3)
    assert all(np.array_equal(x, y[:2]) for x, y in zip(got2, ref))
    # a failing block: member 9 lies in the third block (device index 2 of the handle)
    x = rng.matrix(6131, 13, 128, 32)
    s = x @ np.swapaxes(x, -1, -2) + 128 * np.eye(128)
    bad = s.copy()
    bad[9, 50, 50] = -1.0
    for _ in range(2):
        with pytest.raises(ValueError, match="device 2 of the handle"):
            la.cholesky_decomp(bad, device=h3)
        L = la.cholesky_decomp(s, device=h3)               # the handle and all three blocks' staging stay usable
        assert np.array_equal(L, la.cholesky_decomp(s))
    h3.close()
    monkeypatch.delenv("ND4HIP_TEST_ALLOW_DUP_DEVICES")
    with pytest.raises(_lib.Nd4HipError):
        _lib.Handle([0, 0])
