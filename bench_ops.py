"""Side measurements that ride along in bench.py's JSON line ("ops"): the other BASELINE configs.

  lu2048 / qr2048 / svd2048   one 2048^2 matrix on ONE GPU (replicas only: single matrices do not shard)
  svd_batch                   BASELINE configs[4]: batch x (512x512) SVDs, batch axis sharded over the
                              ranks (contiguous blocks), no data-path collective; afterwards a health
                              all-reduce (max sweeps / max off-norm) and the all-gather of sv over RCCL.
Timing protocol: device-resident operands, warm-ups, MEDIAN of >= 10 runs, each run bracketed by HIP events on
the stream the kernels are launched on. Flop conventions: SURVEY.md §8(d) (LU 2/3 N^3, QR with explicit Q 8/3 N^3, SVD nominal
21 N^3; executed Jacobi work in both accountings of that table).
"""
import os
import statistics
import time

import torch

PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
EPS = 2.220446049250313e-16


def _median_ms(fn, h, reps=10, warm=2, prep=None):
    """prep (optional) runs before every call, outside the timed region (e.g. restoring an operand that fn overwrites)."""
    h.set_stream(torch.cuda.current_stream().cuda_stream)   # (before the warm-ups: a change of stream re-orders the workspace arena once)
    for _ in range(warm):                  # workspace allocation, code load, clocks back up after an idle phase
        if prep is not None:
            prep()
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        if prep is not None:
            prep()
        h.timer_start()
        fn()
        ms.append(h.timer_stop())
    _median_ms.last = [round(x, 3) for x in ms]      # every iteration of the last measurement (a lazy first iteration shows here)
    return statistics.median(ms), min(ms), max(ms)


def _entry(ms3, flops, **extra):
    ms, lo, hi = ms3
    d = {"ms": round(ms, 3), "ms_min": round(lo, 3), "ms_max": round(hi, 3), "timing": "median of 10 (HIP events)",
         "gflops": round(flops / ms / 1e6, 1), "frac_mfma_peak": round(flops / ms / 1e9 / PEAK_FP64_TFLOPS, 4), "algorithmic_flops": flops}
    d.update(extra)
    return d


from nd4js_amd.dist import shard  # noqa: E402  contiguous block of the batch axis owned by a rank


def svd_accounting(N, ms, info, blocked=True):
    """Both executed-work accountings of SURVEY.md §8(d) for a one-sided Jacobi SVD that took `ms`."""
    sweeps, rot = info.get("sweeps", 0), info.get("rotations", 0)
    f_nom = 21.0 * N ** 3
    out = {"ms": round(ms, 2), "gflops_nominal": round(f_nom / ms / 1e6, 1), "nominal_flops": f_nom,
           "sweeps": sweeps, "rotations_applied": rot, "rotations_per_sweep_full": N * (N - 1) // 2,
           "offnorm": info.get("offnorm"), "offnorm_gate": N * EPS,
           "algorithm": "block one-sided Jacobi (32-row blocks; Gram / rotation rounds / apply, Gram and apply on fp64 MFMA)" if blocked
           else "row-pair one-sided Jacobi"}
    # (1) §8(d) "one-sided Jacobi executed work": 7 N^2 (N-1) flop and 32 N^2 (N-1) bytes per sweep of an UNBLOCKED sweep;
    #     counted per rotation actually applied (14 N flop, 64 N bytes each) so that skipped pairs do not inflate it
    f_useful = 14.0 * N * rot
    b_unblocked = 64.0 * N * rot
    out["useful_jacobi"] = {"flops": f_useful, "tflops": round(f_useful / ms / 1e9, 3), "frac_mfma_peak": round(f_useful / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                            "unblocked_bytes": b_unblocked, "unblocked_GBps": round(b_unblocked / ms / 1e6, 1),
                            "unblocked_frac_hbm_peak": round(b_unblocked / ms / 1e6 / PEAK_HBM_GBS, 4),
                            "note": "14 N flop / 64 N bytes per applied rotation = SURVEY 8(d)'s 7 N^2 (N-1) flop, 32 N^2 (N-1) B per full sweep; "
                                    "the blocked kernels keep row blocks in registers/LDS, so the byte figure can exceed the HBM peak"}
    if blocked:
        # (2) what the block kernels execute: per sweep 12 N^3 flop on MFMA (Gram 2/3 of 2 N^3 ... + two applies) and
        #     5 passes over a 64-row pair per step = 40 N^2 (N/32 - 1) bytes
        nblk = N // 32
        f_ex = 12.0 * N ** 3 * sweeps
        b_ex = 40.0 * N * N * (nblk - 1) * sweeps
        out["executed_block"] = {"flops": f_ex, "tflops": round(f_ex / ms / 1e9, 2), "frac_mfma_peak": round(f_ex / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                                 "bytes": b_ex, "GBps": round(b_ex / ms / 1e6, 1), "frac_hbm_peak": round(b_ex / ms / 1e6 / PEAK_HBM_GBS, 4)}
    return out


def svd_checks(dev, A, N):
    """The result of svd_decomp(A) against what the reference pins: its singular values (the committed C4 fixture, N = 2048, seed 9:
    max |d sigma| / sigma_max) and its own acceptance bounds (_generic_test_svd_decomp.js:142-154): ||A - U diag(sv) V||_F <=
    48 eps N ||A||_F, max |U^T U - I| <= 4 eps N, max |V V^T - I| <= 4 eps N. Products on the device's own GEMM."""
    import json
    import numpy as np
    U, sv, V = dev.svd_decomp(A)
    eye = torch.eye(N, dtype=torch.float64, device="cuda")
    USV = dev.matmul2(U * sv.unsqueeze(0), V)
    res = {"residual_fro": float(torch.linalg.norm(A - USV)), "residual_limit": 48.0 * EPS * N * float(torch.linalg.norm(A)),
           "orth_U_max": float((dev.gemm_ex(True, False, 1.0, U, U, 0.0, torch.empty_like(U), N, N, N, N, N, N) - eye).abs().max()),
           "orth_V_max": float((dev.gemm_ex(False, True, 1.0, V, V, 0.0, torch.empty_like(V), N, N, N, N, N, N) - eye).abs().max()),
           "orth_limit": 4.0 * EPS * N,
           "sv_sorted_nonnegative": bool((sv[:-1] >= sv[1:]).all().item() and (sv >= 0).all().item())}
    root = os.path.dirname(os.path.abspath(__file__))
    try:
        with open(os.path.join(root, "tests", "golden", "manifest.json")) as fh:
            g = json.load(fh)["cases"].get("c4_svd%d" % N)
        if g and g.get("seed") == 9:
            ref = np.load(os.path.join(root, "tests", "golden", g["files"]["sv"]))
            res["sv_vs_reference_max_rel"] = float(np.abs(sv.cpu().numpy() - ref).max() / ref.max())
    except Exception as ex:  # pragma: no cover
        res["golden_error"] = repr(ex)
    return res


def run(world, rank, local, dist, svd_batch=None, n_single=2048, end_to_end=True):
    from nd4js_amd import _lib, dev
    h = _lib.handle(local)
    out = {}
    if world == 1:
        N = n_single
        A = dev.fill_uniform(7, (N, N))
        # the CPU-baseline leg before this leaves the GPU idle for seconds: spin the clocks back up (~0.2 s of GEMMs)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.2:
            dev.matmul2(A, A)
            torch.cuda.synchronize()
        out["lu%d" % N] = _entry(_median_ms(lambda: dev.lu_decomp(A), h, warm=3), 2.0 / 3.0 * N ** 3, ms_each=list(_median_ms.last))
        out["qr%d" % N] = _entry(_median_ms(lambda: dev.qr_decomp(A), h), 8.0 / 3.0 * N ** 3)
        # beyond the register-resident panels (N > 2048: two-level blocking; the reference author's own benchmark range ends at
        # N ~ 3100, benchmarks/bench_la_decomps.html:285-288): same flop conventions, median of 5
        if os.environ.get("ND4_BENCH_LARGE", "1") != "0":
            for NL in (4096, 8192):
                AL = dev.fill_uniform(17, (NL, NL))
                out["lu%d" % NL] = _entry(_median_ms(lambda: dev.lu_decomp(AL), h, reps=5, warm=1), 2.0 / 3.0 * NL ** 3)
                out["qr%d" % NL] = _entry(_median_ms(lambda: dev.qr_decomp(AL), h, reps=5, warm=1), 8.0 / 3.0 * NL ** 3)
                out["lu%d" % NL]["timing"] = out["qr%d" % NL]["timing"] = "median of 5 (HIP events)"
                del AL
                torch.cuda.empty_cache()
        # N1: lu_solve with N right-hand sides on the device-resident factors: 2 N^2 J flop
        LUd, Pd = dev.lu_decomp(A)
        Y = dev.fill_uniform(11, (N, N))
        out["lu_solve%d" % N] = _entry(_median_ms(lambda: dev.lu_solve(LUd, Pd, Y), h), 2.0 * N ** 3, rhs_columns=N)
        # N4: Cholesky of S = B B^T + N I (formed on the device), N^3/3 flop
        S = dev.gemm_ex(False, True, 1.0, A, A, 0.0, torch.empty_like(A), N, N, N, N, N, N)
        S.diagonal().add_(float(N))
        out["cholesky%d" % N] = _entry(_median_ms(lambda: dev.cholesky_decomp(S), h), N ** 3 / 3.0)
        # the remaining built rows under the same clock: LDL^T (N^3/3), Hessenberg reduction with U
        # (10/3 N^3 + 4/3 N^3), bidiagonalisation with U and V (8/3 N^3 + 2 x 4/3 N^3), least squares on device-resident factors with
        # N right-hand sides (qr_lstsq: Q^T Y + back substitution = 3 N^3; svd_lstsq: two products = 4 N^3); median of 5
        Sl = S.clone()
        Sl.diagonal().sub_(2.0 * float(N))                  # indefinite (LDL^T does not need definiteness), still diagonally dominant
        out["ldl%d" % N] = _entry(_median_ms(lambda: dev.ldl_decomp(Sl), h, reps=5, warm=1), N ** 3 / 3.0)
        out["hess%d" % N] = _entry(_median_ms(lambda: dev.hessenberg_decomp(A), h, reps=5, warm=1), 14.0 / 3.0 * N ** 3)
        out["bidiag%d" % N] = _entry(_median_ms(lambda: dev.bidiag_decomp(A), h, reps=5, warm=1), 16.0 / 3.0 * N ** 3)
        Qd, Rd = dev.qr_decomp(A)
        out["qr_lstsq%d" % N] = _entry(_median_ms(lambda: dev.qr_lstsq(Qd, Rd, Y), h, reps=5, warm=1), 3.0 * N ** 3, rhs_columns=N)
        for k in ("ldl%d" % N, "hess%d" % N, "bidiag%d" % N, "qr_lstsq%d" % N):
            out[k]["timing"] = "median of 5 (HIP events)"
        del Sl, Qd, Rd
        A9 = dev.fill_uniform(9, (N, N))
        info = {}
        ms3 = _median_ms(lambda: dev.svd_decomp(A9, info=info), h, reps=10, warm=1)
        e = svd_accounting(N, ms3[0], info, blocked=N >= 16)
        e.update({"ms_min": round(ms3[1], 2), "ms_max": round(ms3[2], 2), "timing": "median of 10 (HIP events)"})
        e["checks"] = svd_checks(dev, A9, N)
        out["svd%d" % N] = e
        U9, s9, V9 = dev.svd_decomp(A9)
        out["svd_lstsq%d" % N] = _entry(_median_ms(lambda: dev.svd_lstsq(U9, s9, V9, Y), h, reps=5, warm=1), 4.0 * N ** 3, rhs_columns=N)
        out["svd_lstsq%d" % N]["timing"] = "median of 5 (HIP events)"
        del U9, s9, V9
        # small problems (VERDICT r2 #7a): all sweeps of an N <= 64 matrix in ONE launch (jac_small), device-resident
        try:
            S1 = dev.fill_uniform(31, (32, 32))
            SB = dev.fill_uniform(32, (8192, 32, 32))
            m1 = _median_ms(lambda: dev.svd_decomp(S1), h, reps=20, warm=3)
            mb = _median_ms(lambda: dev.svd_decomp(SB), h, reps=5, warm=1)
            out["svd_small"] = {"single_32x32_us": round(m1[0] * 1e3, 1), "batch_8192x32x32_ms": round(mb[0], 3),
                                "us_per_matrix_batched": round(mb[0] * 1e3 / 8192, 3), "timing": "median (HIP events), device-resident",
                                "kernel": "jac_small<32> (one workgroup per matrix, all sweeps) + the common epilogue launches"}
            del S1, SB
        except Exception as ex:  # pragma: no cover
            out["svd_small"] = {"error": repr(ex)}
        # QR panel (north_star: >= 50 % of HBM peak "on the QR panel"): geqr2 + larft of 16 columns, one workgroup per matrix;
        # algorithmic bytes = 16 m b (each panel element read once and written once)
        try:
            out["qr_panel"] = qr_panel(h, dev)
        except Exception as ex:  # pragma: no cover
            out["qr_panel"] = {"error": repr(ex)}
    # ---- batched LU / QR (1024 x 512^2, the shapes of BASELINE configs[4]), batch axis sharded over ranks like the SVD below:
    #      the regime in which the chip is filled (qr.js:43-49, lu.js:34-40 loop over independent matrices) ----
    B = int(os.environ.get("ND4_BENCH_SVD_BATCH", svd_batch or 1024))
    n = 512
    lo, hi = shard(B, world, rank)
    mine = hi - lo
    for name, fn, fl, by in (("lu_batch", dev.lu_decomp, 2.0 / 3.0 * n ** 3, 16.0 * n * n), ("qr_batch", dev.qr_decomp, 8.0 / 3.0 * n ** 3, 24.0 * n * n)):
        try:
            out[name] = batch_decomp(h, dev, dist, world, rank, name, fn, B, n, lo, hi, fl, by)
        except Exception as ex:  # pragma: no cover
            out[name] = {"error": repr(ex)}
    X = torch.empty((mine, n, n), dtype=torch.float64, device="cuda")
    for k in range(mine):
        _lib.check(h.lib.nd4hip_fill_uniform_dev(h.ptr, 1000 + lo + k, 0, n * n, X[k].data_ptr()))
    info = {}
    dev.svd_decomp(X[: min(mine, 8)], info=info)        # warm-up on a slice
    nccl = dist is not None and dist.get_backend() == "nccl"

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if dist is None:
            return v
        tt = torch.tensor([v], dtype=torch.float64, device="cuda" if nccl else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.item()

    from nd4js_amd.dist import svd_decomp_sharded
    sync()
    t0 = time.perf_counter()
    U, sv_all, V, hl = svd_decomp_sharded(X, B)         # local SVDs + R1 health all-reduce + R2 sv all-gather
    sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    f = 21.0 * n ** 3 * B
    res = {"batch": B, "n": n, "n_gpus": world, "matrices_per_rank": mine,
           "kernel_only": {"seconds": round(dt, 4), "gflops_nominal": round(f / dt / 1e9, 1), "matrices_per_s": round(B / dt, 1),
                           "what": "device-resident shard -> local SVDs -> R1 all-reduce(max) -> R2 all-gather(sv); barrier + sync on both sides, max over ranks"},
           "seconds": round(dt, 4), "gflops_nominal": round(f / dt / 1e9, 1), "matrices_per_s": round(B / dt, 1),
           "max_sweeps": hl["max_sweeps"], "max_offnorm": hl["max_offnorm"], "offnorm_gate": n * EPS,
           "scaling": "strong (fixed batch sharded over ranks)"}
    if end_to_end:
        # end-to-end through the host-pointer C ABI (what the JS host sees): this rank's block of the host batch in pinned
        # memory -> H2D -> kernels -> D2H of U, sv, V. Expected to be PCIe-bound: 6 GiB cross the bus for 1024 matrices.
        # Only the LOCAL work sits in try/except: every rank joins every collective (barriers, the max over ranks, the error flag),
        # whatever happened to it, so a failure on one rank cannot leave the others waiting.
        import ctypes
        err, hS, same = None, None, None
        try:
            hX = torch.empty((mine, n, n), dtype=torch.float64).pin_memory()
            hX.copy_(X)
            hU, hV = torch.empty_like(hX).pin_memory(), torch.empty_like(hX).pin_memory()
            hS = torch.empty((mine, n), dtype=torch.float64).pin_memory()
        except Exception as ex:  # pragma: no cover
            err = repr(ex)
        sync()
        t0 = time.perf_counter()
        if err is None:
            try:
                _lib.check(h.lib.nd4hip_dgesvdj_batched(h.ptr, mine, n, n, ctypes.c_void_p(hX.data_ptr()), ctypes.c_void_p(hU.data_ptr()),
                                                        ctypes.c_void_p(hS.data_ptr()), ctypes.c_void_p(hV.data_ptr()), None, None))
            except Exception as ex:  # pragma: no cover
                err = repr(ex)
        sync()
        de = max_over_ranks(time.perf_counter() - t0)
        if err is None:
            same = bool(torch.equal(hS.cuda(), sv_all[lo:hi]))
        any_err = max_over_ranks(0.0 if err is None else 1.0)
        if any_err:
            res["end_to_end"] = {"error": err or "failed on another rank"}
        else:
            res["end_to_end"] = {"seconds": round(de, 4), "gflops_nominal": round(f / de / 1e9, 1), "matrices_per_s": round(B / de, 1),
                                 "bytes_over_pcie_per_rank": int(mine * (3 * n * n + n) * 8), "bound": "PCIe (H2D of A, D2H of U, sv, V)",
                                 "sv_bit_identical_to_device_resident": same}
        hX = hU = hV = hS = None
    if rank == 0:
        # parity gate: members with golden sv (every 16th) against the reference
        try:
            import json
            import numpy as np
            root = os.path.dirname(os.path.abspath(__file__))
            with open(os.path.join(root, "tests", "golden", "manifest.json")) as fh:
                g = json.load(fh)["cases"].get("c5_svd512")
            if g:
                members = np.load(os.path.join(root, "tests", "golden", g["files"]["members"]))
                ref = np.load(os.path.join(root, "tests", "golden", g["files"]["sv"]))
                keep = members < B
                got = sv_all[torch.from_numpy(members[keep].astype("int64")).cuda()].cpu().numpy()
                res["sv_vs_reference_max_rel"] = float(np.abs(got - ref[keep]).max() / ref[keep].max())
                res["members_checked"] = int(keep.sum())
        except Exception as e:  # pragma: no cover
            res["parity_error"] = repr(e)
    out["svd_batch"] = res
    return out


def batch_decomp(h, dev, dist, world, rank, name, fn, B, n, lo, hi, flops_each, bytes_each, reps=5):
    """`B` independent n x n factorisations, this rank's contiguous block [lo, hi) device-resident, no collective on the data path;
    per repetition: barrier + synchronize on both sides, the MAX over ranks of the wall time; median of `reps`."""
    from nd4js_amd import _lib
    mine = hi - lo
    X = torch.empty((mine, n, n), dtype=torch.float64, device="cuda")
    for k in range(mine):
        _lib.check(h.lib.nd4hip_fill_uniform_dev(h.ptr, 3000 + lo + k, 0, n * n, X[k].data_ptr()))
    nccl = dist is not None and dist.get_backend() == "nccl"

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fn(X[: min(mine, 8)])
    fn(X)
    times = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn(X)
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if nccl else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        times.append(dt)
    dt = statistics.median(times)
    return {"batch": B, "n": n, "n_gpus": world, "matrices_per_rank": mine, "seconds": round(dt, 5), "matrices_per_s": round(B / dt, 1),
            "gflops": round(flops_each * B / dt / 1e9, 1), "frac_mfma_peak": round(flops_each * B / dt / 1e12 / PEAK_FP64_TFLOPS / world, 4),
            "algorithmic_bytes": bytes_each * B, "GBps": round(bytes_each * B / dt / 1e9, 1),
            "frac_hbm_peak": round(bytes_each * B / dt / 1e9 / PEAK_HBM_GBS / world, 4),
            "timing": "median of %d (wall clock, barrier + synchronize on both sides, max over ranks), device-resident" % reps,
            "scaling": "strong (fixed batch sharded over ranks)"}


def qr_panel(h, dev, M=2048, b=16, batch=256, nbuf=8):
    """One panel factorisation (16 columns: R, the reflector block V and its factor T) on its own, through its entry point
    nd4hip_dgeqr2_panel_batched_dev: a single 2048-row panel (a latency chain), 256 of them in one launch (one workgroup of 8 waves
    per panel and per CU), and shorter panels, where several workgroups share a CU: 2048 panels of 512 rows (2 waves each), 1024 of
    1024 rows, 4096 of 256 rows. Algorithmic bytes 16 m b per panel (read once, V written once). Timing: `nbuf` calls back to back
    on `nbuf` different inputs between two HIP events (so that neither the launch latency of a 50 us kernel nor a cache-resident
    input counts), median of 10 such groups; the inputs are restored outside the timed region."""
    import ctypes
    from nd4js_amd import _lib
    res = {"cols": b, "algorithmic_bytes_per_panel_row": 16 * b, "timing": "median of 10 groups of %d back-to-back calls on distinct inputs (HIP events)" % nbuf}
    for name, rows, nb in (("single", M, 1), ("batched", M, batch), ("batched_512rows", 512, 2048), ("batched_1024rows", 1024, 1024),
                           ("batched_256rows", 256, 4096), ("batched_1024x2048rows", M, 1024)):
        nbf = nbuf if nb * rows <= 2048 * 512 else 4
        A = dev.fill_uniform(21, (nb, rows, b))
        V = torch.empty_like(A)
        T = torch.empty((nb, b, b), dtype=torch.float64, device="cuda")
        Ws = [A.clone() for _ in range(nbf)]

        def go():
            for W in Ws:
                _lib.check(h.lib.nd4hip_dgeqr2_panel_batched_dev(h.ptr, nb, rows, b, ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(V.data_ptr()),
                                                                 ctypes.c_void_p(T.data_ptr())))

        def restore():                     # outside the timed region: a call overwrites (at least) the top block of its input with R
            for W in Ws:
                W.copy_(A)
        us = _median_ms(go, h, reps=10, warm=2, prep=restore)[0] * 1e3 / nbf
        byts = 16.0 * rows * b * nb
        res[name] = {"panels": nb, "rows": rows, "us": round(us, 2), "bytes": byts, "GBps": round(byts / us / 1e3, 1),
                     "frac_hbm_peak": round(byts / us / 1e3 / PEAK_HBM_GBS, 4)}
        del Ws, A, V, T
    return res
