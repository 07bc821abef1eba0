"""Experiment: a batch of SVDs as k sub-batches on k handles (streams) of ONE device, driven by k host threads, against the whole batch
on one handle. usage: python tools/svd_streams.py [batch] [n] [k ...]"""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nd4js_amd import _lib, dev


def run(handles, A, U, sv, V):
    k = len(handles)
    B, M, N = A.shape
    per = (B + k - 1) // k
    errs = []

    def work(i):
        lo, hi = i * per, min(B, (i + 1) * per)
        if lo >= hi:
            return
        h = handles[i]
        sw, off = ctypes.c_int(0), ctypes.c_double(0.0)
        rc = h.lib.nd4hip_dgesvdj_batched_dev(h.ptr, hi - lo, M, N, ctypes.c_void_p(A[lo:hi].data_ptr()), ctypes.c_void_p(U[lo:hi].data_ptr()),
                                              ctypes.c_void_p(sv[lo:hi].data_ptr()), ctypes.c_void_p(V[lo:hi].data_ptr()), ctypes.byref(sw), ctypes.byref(off))
        if rc == 0:
            rc = h.lib.nd4hip_synchronize(h.ptr)
        errs.append(rc)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert all(e == 0 for e in errs), errs
    return time.perf_counter() - t0


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    ks = [int(x) for x in sys.argv[3:]] or [1, 2, 4]
    A = dev.fill_uniform(5, (B, n, n))
    U = torch.empty_like(A); V = torch.empty_like(A); sv = torch.empty((B, n), dtype=torch.float64, device="cuda")
    ref = None
    for k in ks:
        hs = [_lib.Handle(0) for _ in range(k)]
        run(hs, A, U, sv, V)
        t = min(run(hs, A, U, sv, V) for _ in range(3))
        s = sv.clone()
        if ref is None:
            ref = s
        print("k=%d  %.1f ms  (%.0f matrices/s)  sv identical to k=%d: %s" % (k, t * 1e3, B / t, ks[0], bool(torch.equal(s, ref))), flush=True)
        for h in hs:
            h.close()


if __name__ == "__main__":
    main()
