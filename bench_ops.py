"""Side measurements that ride along in bench.py's JSON line ("ops"): the other BASELINE configs.

  lu2048 / qr2048 / svd2048   one 2048^2 matrix on ONE GPU (replicas only: single matrices do not shard)
  svd_batch                   BASELINE configs[4]: batch x (512x512) SVDs, batch axis sharded over the
                              ranks (contiguous blocks), no data-path collective; afterwards a health
                              all-reduce (max sweeps / max off-norm) and the all-gather of sv over RCCL.
Flop conventions: SURVEY.md §8(d) (LU 2/3 N^3, QR with explicit Q 8/3 N^3, SVD nominal 21 N^3).
"""
import os
import time

import torch

PEAK_FP64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def _time(fn, h, reps):
    fn()                                   # warm-up (workspace allocation, code load)
    fn()                                   # and once more: after an idle phase (the CPU baseline leg) the clocks ramp up again
    torch.cuda.synchronize()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.timer_start()
    for _ in range(reps):
        fn()
    return h.timer_stop() / reps


from nd4js_amd.dist import shard  # noqa: E402  contiguous block of the batch axis owned by a rank


def run(world, rank, local, dist, svd_batch=None, n_single=2048):
    from nd4js_amd import _lib, dev
    h = _lib.handle(local)
    out = {}
    if world == 1:
        N = n_single
        A = dev.fill_uniform(7, (N, N))
        # the CPU-baseline leg before this leaves the GPU idle for ~10 s: spin the clocks back up (~0.2 s of GEMMs), otherwise
        # the first, latency-bound measurement (LU) reads 30-40 % slow
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.2:
            dev.matmul2(A, A)
            torch.cuda.synchronize()
        ms = _time(lambda: dev.lu_decomp(A), h, 5)
        f = 2.0 / 3.0 * N ** 3
        out["lu%d" % N] = {"ms": round(ms, 3), "gflops": round(f / ms / 1e6, 1), "frac_mfma_peak": round(f / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                           "algorithmic_flops": f}
        ms = _time(lambda: dev.qr_decomp(A), h, 5)
        f = 8.0 / 3.0 * N ** 3
        out["qr%d" % N] = {"ms": round(ms, 3), "gflops": round(f / ms / 1e6, 1), "frac_mfma_peak": round(f / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                           "algorithmic_flops": f}
        # N1: lu_solve with N right-hand sides on the device-resident factors: 2 N^2 J flop
        LUd, Pd = dev.lu_decomp(A)
        Y = dev.fill_uniform(11, (N, N))
        ms = _time(lambda: dev.lu_solve(LUd, Pd, Y), h, 5)
        f = 2.0 * N * N * N
        out["lu_solve%d" % N] = {"ms": round(ms, 3), "gflops": round(f / ms / 1e6, 1), "frac_mfma_peak": round(f / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                                 "rhs_columns": N, "algorithmic_flops": f}
        # N4: Cholesky of S = B B^T + N I (formed on the device), N^3/3 flop
        S = dev.gemm_ex(False, True, 1.0, A, A, 0.0, torch.empty_like(A), N, N, N, N, N, N)
        S.diagonal().add_(float(N))
        ms = _time(lambda: dev.cholesky_decomp(S), h, 5)
        f = N ** 3 / 3.0
        out["cholesky%d" % N] = {"ms": round(ms, 3), "gflops": round(f / ms / 1e6, 1), "frac_mfma_peak": round(f / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                                 "algorithmic_flops": f}
        A9 = dev.fill_uniform(9, (N, N))
        info = {}
        ms = _time(lambda: dev.svd_decomp(A9, info=info), h, 1)
        f = 21.0 * N ** 3
        sweeps = info.get("sweeps", 0)
        blocked = (N % 64 == 0 and N >= 128)
        nblk = N // 32
        if blocked:      # block Jacobi (svd_block.hip): per sweep 12 N^3 flop on MFMA, 5 passes over W/Ut per step
            ex_flops = 12.0 * N ** 3 * sweeps
            ex_bytes = 40.0 * N * N * (nblk - 1) * sweeps
        else:            # row-pair Jacobi (svd.hip): SURVEY.md §8d
            ex_flops = 7.0 * N * N * (N - 1) * sweeps
            ex_bytes = 32.0 * N * N * (N - 1) * sweeps
        out["svd%d" % N] = {"ms": round(ms, 2), "gflops_nominal": round(f / ms / 1e6, 1), "sweeps": sweeps,
                            "algorithm": "block one-sided Jacobi (32-row blocks, Gram/eigen/apply on MFMA)" if blocked else "row-pair one-sided Jacobi",
                            "executed_gflops": round(ex_flops / ms / 1e6, 1), "executed_frac_mfma_peak": round(ex_flops / ms / 1e9 / PEAK_FP64_TFLOPS, 4),
                            "executed_traffic_GBps": round(ex_bytes / ms / 1e6, 1), "executed_frac_hbm_peak": round(ex_bytes / ms / 1e6 / PEAK_HBM_GBS, 4),
                            "offnorm": info.get("offnorm")}
    # ---- batched SVD, batch axis sharded over ranks ----
    B = int(os.environ.get("ND4_BENCH_SVD_BATCH", svd_batch or 1024))
    n = 512
    lo, hi = shard(B, world, rank)
    mine = hi - lo
    X = torch.empty((mine, n, n), dtype=torch.float64, device="cuda")
    for k in range(mine):
        _lib.check(h.lib.nd4hip_fill_uniform_dev(h.ptr, 1000 + lo + k, 0, n * n, X[k].data_ptr()))
    info = {}
    dev.svd_decomp(X[: min(mine, 8)], info=info)        # warm-up on a slice
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from nd4js_amd.dist import svd_decomp_sharded
    U, sv_all, V, hl = svd_decomp_sharded(X, B)         # local SVDs + R1 health all-reduce + R2 sv all-gather
    health = torch.tensor([float(hl["max_sweeps"]), hl["max_offnorm"]], dtype=torch.float64)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    f = 21.0 * n ** 3 * B
    res = {"batch": B, "n": n, "n_gpus": world, "seconds": round(dt, 4), "gflops_nominal": round(f / dt / 1e9, 1),
           "matrices_per_s": round(B / dt, 1), "max_sweeps": int(health[0].item()), "max_offnorm": health[1].item(),
           "scaling": "strong (fixed batch sharded over ranks)"}
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
        except Exception as e:  # pragma: no cover
            res["parity_error"] = repr(e)
    out["svd_batch"] = res
    return out
