#!/usr/bin/env python3
"""bench.py — headline benchmark of the nd.la hot path on MI355X (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Step = one 4096x4096x4096 fp64 matmul (BASELINE configs[1]) through the C ABI on device-resident
synthetic inputs. With N ranks every rank multiplies its own 4096^2 pair (the reference's batch axis
sharded one matrix per rank, no data-path collective): weak scaling, value = aggregate GFLOP/s.
Rank 0 prints ONE JSON line; extra per-op measurements (QR / LU / SVD configs) ride along in "ops".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6      # MI355X fp64 MFMA = vector peak
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--n", type=int, default=4096, help="matrix size (default = BASELINE configs[1])")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-ops", action="store_true", help="skip the QR/LU/SVD side measurements")
    return p.parse_args()


# The true reference (nd4js dist/nd.js under Node 12) timed in the survey container (8-vCPU Xeon 2.1 GHz, 1 thread):
# fixed reference points printed beside the port's numbers; the reference cannot travel to the GPU box.
REFERENCE_JS_SURVEY_S = {"matmul4096": 126.1, "qr2048": 11.43, "lu2048": 6.26, "svd2048": 70.5, "svd512": 1.407}


def cpu_baseline_matmul(n, rows):
    """Oracle (CPU port of matmul.js:49-53) on a bounded sample: the first `rows` rows of C, i.e.
    the same i-k-j loop streaming all of B, 1 thread (the reference is single-threaded)."""
    import oracle
    from nd4js_amd import rng
    a = rng.matrix(5, n, n)[:rows].copy()
    b = rng.matrix(6, n, n)
    t = time.perf_counter()
    c = oracle.matmul2(a, b)
    dt = time.perf_counter() - t
    return {"value": round(2.0 * rows * n * n / dt / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": "rows 0..%d of the %dx%d product (oracle/nd4_oracle.c i-k-j loop, %.1f s)" % (rows - 1, n, n, dt),
            "host_cpus": os.cpu_count(), "host_cpus_available": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
            "cpu_model": __import__("oracle.cpu_batch", fromlist=["cpu_model"]).cpu_model(),
            "reference_js_survey_s": REFERENCE_JS_SURVEY_S,
            "reference_js_survey_gflops": round(2.0 * 4096 ** 3 / REFERENCE_JS_SURVEY_S["matmul4096"] / 1e9, 3)}, c


def cpu_baseline_ops(n=2048):
    """The oracle (C port, 1 core, same flop conventions) beside every side op: LU, QR and svd_decomp of the full 2048^2 config once
    each, svd_decomp at 512^2."""
    import oracle
    from nd4js_amd import rng
    out = {}
    a = rng.matrix(7, n, n)
    t = time.perf_counter()
    lu, p = oracle.lu_decomp(a)
    dt = time.perf_counter() - t
    out["lu%d" % n] = {"seconds": round(dt, 3), "gflops": round(2.0 / 3.0 * n ** 3 / dt / 1e9, 3), "cores": 1, "kind": "port",
                       "reference_js_survey_s": REFERENCE_JS_SURVEY_S.get("lu%d" % n)}
    t = time.perf_counter()
    q, r = oracle.qr_decomp(a)
    dt = time.perf_counter() - t
    out["qr%d" % n] = {"seconds": round(dt, 3), "gflops": round(8.0 / 3.0 * n ** 3 / dt / 1e9, 3), "cores": 1, "kind": "port",
                       "reference_js_survey_s": REFERENCE_JS_SURVEY_S.get("qr%d" % n)}
    # svd_decomp: the reference's OWN algorithm (svd_dc: bidiagonalisation + divide & conquer, svd_dc.js:883-932, restated in
    # oracle/nd4_oracle_svd_dc.c and pinned to the reference's goldens) on 1 core: member 0 of the 1024 x 512^2 batch, and the
    # 2048^2 config itself (~30 s; ND4_BENCH_CPU_SVD2048=0 skips it)
    m = 512
    x = rng.matrix(1000, m, m)
    t = time.perf_counter()
    _, sv, _ = oracle.svd_dc(x)
    dt = time.perf_counter() - t
    out["svd%d" % m] = {"seconds": round(dt, 3), "gflops_nominal": round(21.0 * m ** 3 / dt / 1e9, 3), "cores": 1, "kind": "port",
                        "algorithm": "svd_dc (bidiagonalisation + divide & conquer), the reference's svd_decomp",
                        "reference_js_survey_s": REFERENCE_JS_SURVEY_S["svd512"],
                        "note": "sample for the 1024 x 512^2 config: matrix 0 of the batch"}
    if os.environ.get("ND4_BENCH_CPU_SVD2048", "1") != "0":
        x2 = rng.matrix(9, n, n)
        t = time.perf_counter()
        _, sv2, _ = oracle.svd_dc(x2)
        dt = time.perf_counter() - t
        out["svd%d" % n] = {"seconds": round(dt, 3), "gflops_nominal": round(21.0 * n ** 3 / dt / 1e9, 3), "cores": 1, "kind": "port",
                            "algorithm": "svd_dc (bidiagonalisation + divide & conquer), the reference's svd_decomp",
                            "reference_js_survey_s": REFERENCE_JS_SURVEY_S.get("svd%d" % n), "sv_max": float(sv2[0])}
    return out, (p, r, sv)


def cpu_baseline_svd_batch(count=64):
    """BASELINE configs[4] on ALL host cores: `count` members of the 1024 x 512^2 batch spread over the cores this
    process may use (at most 16: the GPU box's CPU share per GPU), in a child process (no fork after the GPU is initialised);
    the reference's own algorithm (svd_dc) as restated in oracle/nd4_oracle_svd_dc.c."""
    import subprocess
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(avail, 16, count))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_batch.py"), "--count", str(count), "--workers", str(workers), "--n", "512",
                        "--algo", "dc"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-400:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d.update({"kind": "port", "algorithm": "svd_dc (the reference's svd_decomp)", "cores": d["workers"],
              "sample": "%d members (seeds 1000..) of the 1024 x 512^2 batch, spread over %d processes" % (d["count"], d["workers"]),
              "batch1024_extrapolated_s": round(1024.0 / d["matrices_per_s"], 1),
              "reference_js_survey_batch1024_1core_s": round(1024 * REFERENCE_JS_SURVEY_S["svd512"], 0)})
    return d


def node_leg(n=4096, svd_batch=1024):
    """The production boundary (the Node.js host over the N-API addon, matmul.js:91-147 reached through
    nd4js_amd/js): 4096^2 matmul2 and a batch of 512^2 SVDs from host Float64Arrays and from DeviceNDArrays, timed by
    tools/node_bench.js in a child process. Skipped when node or the addon is absent."""
    import shutil
    import subprocess
    node = shutil.which("node")
    addon = os.path.join(ROOT, "nd4js_amd", "js", "nd4hip_napi.node")
    if not node or not os.path.exists(addon):
        return {"skipped": "node or the N-API addon is not available on this box"}
    try:
        r = subprocess.run([node, os.path.join(ROOT, "tools", "node_bench.js"), str(n), str(svd_batch)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=300)
    except Exception as ex:  # pragma: no cover
        return {"error": repr(ex)}
    if r.returncode != 0:
        return {"error": (r.stderr or r.stdout)[-400:]}
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["what"] = ("through nd4js_amd/js (N-API): host_* = Float64Array operands and results (PCIe both ways per call), device_* = DeviceNDArray "
                 "operands and results; best of 2-5 wall-clock timings inside node")
    return d


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (torch.distributed.run, one per GPU,
    rendezvous on 127.0.0.1) BEFORE anything in this process touches the GPU, relay rank 0's JSON line and exit with the
    children's code. ND4_BENCH_BACKEND=gloo lets the ranks share fewer GPUs (single-GPU rehearsal of the N>1 path)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one rank per GPU; ND4_BENCH_BACKEND=gloo + fewer GPUs than ranks is the single-GPU rehearsal of the N>1 path
    backend = os.environ.get("ND4_BENCH_BACKEND", "nccl")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from nd4js_amd import _lib, dev

    n = args.n
    A = dev.fill_uniform(5 + 10 * rank, (n, n))
    B = dev.fill_uniform(6 + 10 * rank, (n, n))
    C = torch.empty((n, n), dtype=torch.float64, device="cuda")
    h = _lib.handle(local)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dev.matmul2(A, B, out=C)
    barrier()
    t0 = time.perf_counter()
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.timer_start()                                   # HIP events on the stream the kernel runs on
    for _ in range(args.steps):
        dev.matmul2(A, B, out=C)
    kernel_ms = h.timer_stop() / args.steps
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    flops = 2.0 * n ** 3
    value = world * flops * args.steps / elapsed / 1e9

    # side measurements on EVERY rank (svd_batch is sharded): a failure on one rank must not leave the others in a collective
    ops, ops_error = None, None
    if not args.no_ops:
        import bench_ops
        try:
            ops = bench_ops.run(world, rank, local, dist)
        except Exception as e:  # reported, and the run is marked invalid below
            ops_error = repr(e)
    failed = 1.0 if ops_error else 0.0
    if dist is not None:
        tt = torch.tensor([failed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        failed = tt.item()

    out = None
    if rank == 0:
        import numpy as np
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        gates = {}                                        # name -> (value, limit): every one is ENFORCED below
        out = {
            "metric": "fp64 GFLOP/s: matmul N=4096 & SVD N=2048; % of MFMA/HBM peak",
            "value": round(value, 1), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "matmul %dx%dx%d fp64, one matrix pair per GPU (BASELINE configs[1])" % (n, n, n),
                       "parallelism": "batch-sharded x%d, no collective" % world, "inputs": "uniform(-1,1) seeds 5/6, HBM-resident"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP64_TFLOPS, 4), "traffic": None,
                         "kernel": "dgemm_kernel<NN,vec,full>", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": 24.0 * n * n},
        }
        # HBM-side bytes per launch from the separate rocprofv3 --pmc passes of this same command
        # (tools/pmc_summary.py -> profiles/; a profiler cannot wrap itself inside the timed run)
        pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.startswith("r")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
        for rnd in reversed(pmc):
            f = os.path.join(ROOT, "profiles", rnd, "gemm4096_pmc.json")
            if os.path.exists(f) and n == 4096:
                with open(f) as fh:
                    pj = json.load(fh)
                out["roofline"]["traffic"] = pj["traffic_bytes_per_launch"]
                out["roofline"]["traffic_source"] = "profiles/%s/gemm4096_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, KB->B)" % rnd
                for k in ("mfma_busy_frac", "clock_ghz"):
                    if k in pj:
                        out["roofline"][k] = pj[k]
                break
        # counter figures beside the time-derived fractions of the side ops (same passes, other kernels)
        out["pmc"] = {}
        for rnd in reversed(pmc):
            for name in ("qr_panel_batched_pmc.json", "qr_lu_2048_pmc.json", "svd_small_pmc.json", "svd2048_pmc.json", "hess_bidiag_2048_pmc.json"):
                f = os.path.join(ROOT, "profiles", rnd, name)
                if name not in out["pmc"] and os.path.exists(f):
                    try:
                        with open(f) as fh:
                            pj = json.load(fh)
                        ks = pj.get("kernels", [pj])
                        out["pmc"][name] = {"source": "profiles/%s/%s" % (rnd, name),
                                            "traffic_bytes_per_launch": {k["kernel"]: k.get("traffic_bytes_per_launch") for k in ks if "kernel" in k}}
                    except Exception as ex:  # pragma: no cover
                        out["pmc"][name] = {"error": repr(ex)}
        # parity gates printed with the number and enforced
        parity = {"gate": 1e-10}
        with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
            cases = json.load(f)["cases"]
        g = cases.get("c2_matmul4096")
        if g and n == 4096:
            idx = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", g["files"]["idx"])).astype("int64")).cuda()
            val = np.load(os.path.join(ROOT, "tests", "golden", g["files"]["val"]))
            got = C.reshape(-1)[idx].cpu().numpy()
            parity["matmul_vs_reference_samples_relerr"] = float(np.linalg.norm(got - val) / np.linalg.norm(val))
            gates["matmul_vs_reference_samples_relerr"] = (parity["matmul_vs_reference_samples_relerr"], 1e-10)
        if world == 1 and not args.no_cpu_baseline:
            cb, c_cpu = cpu_baseline_matmul(n, min(1024, n))
            out["cpu_baseline"] = cb
            parity["matmul_vs_oracle_rows_relerr"] = float(
                np.linalg.norm(C[: c_cpu.shape[0]].cpu().numpy() - c_cpu) / np.linalg.norm(c_cpu))
            gates["matmul_vs_oracle_rows_relerr"] = (parity["matmul_vs_oracle_rows_relerr"], 1e-10)
        out["parity"] = parity
        if ops is not None:
            out["ops"] = ops
            sb = ops.get("svd_batch", {})
            if world > 1 and sb:
                # the one config that SHARDS (BASELINE configs[4]): with N > 1 the headline value is N independent matmuls, a curve
                # that cannot bend; this block is the batched-SVD scaling point of this run (strong scaling: fixed batch)
                out["scale"] = {"workload": "batch of %d x (%dx%d) fp64 SVDs sharded over %d GPUs (BASELINE configs[4])" % (sb["batch"], sb["n"], sb["n"], world),
                                "scaling": "strong", "n_gpus": world, "kernel_only": sb.get("kernel_only"), "end_to_end": sb.get("end_to_end"),
                                "max_sweeps": sb.get("max_sweeps"),
                                "lu_batch": ops.get("lu_batch"), "qr_batch": ops.get("qr_batch")}     # the same batch shape through lu_decomp / qr_decomp
            if "sv_vs_reference_max_rel" in sb:
                gates["svd_batch_sv_vs_reference_max_rel"] = (sb["sv_vs_reference_max_rel"], 1e-10)
            # (the off-norm of a converged Jacobi run is <= N eps by construction: reported, not a gate)
            ee = sb.get("end_to_end") or {}
            if "error" in ee:
                gates["svd_batch_end_to_end_failed"] = (1.0, 0.0)
            elif ee.get("sv_bit_identical_to_device_resident") is not None:
                gates["svd_batch_end_to_end_sv_mismatch"] = (0.0 if ee["sv_bit_identical_to_device_resident"] else 1.0, 0.0)
            sv2 = ops.get("svd2048")
            if sv2:
                # the headline's second half as a roofline block of its own: both accountings of SURVEY.md §8(d)
                out["roofline_svd"] = {"workload": "svd_decomp 2048x2048 fp64 (BASELINE configs[3])", "ms": sv2["ms"], "gflops_nominal_21N3": sv2["gflops_nominal"],
                                       "sweeps": sv2["sweeps"], "rotations_applied": sv2["rotations_applied"], "offnorm": sv2["offnorm"],
                                       "useful_jacobi": sv2["useful_jacobi"], "executed_block": sv2.get("executed_block"),
                                       "bound": "mfma (Gram + apply) / LDS-latency chain (rotation rounds)"}
                ck = sv2.get("checks") or {}
                out["roofline_svd"]["checks"] = ck
                # the reference's own acceptance bounds and the committed C4 singular values: these CAN fail
                if "sv_vs_reference_max_rel" in ck:
                    gates["svd2048_sv_vs_reference_max_rel"] = (ck["sv_vs_reference_max_rel"], 1e-10)
                if ck:
                    gates["svd2048_residual_fro"] = (ck["residual_fro"], ck["residual_limit"])
                    gates["svd2048_orth_U_max"] = (ck["orth_U_max"], ck["orth_limit"])
                    gates["svd2048_orth_V_max"] = (ck["orth_V_max"], ck["orth_limit"])
                    gates["svd2048_sv_unsorted_or_negative"] = (0.0 if ck["sv_sorted_nonnegative"] else 1.0, 0.0)
            if world == 1 and os.environ.get("ND4_BENCH_NODE", "1") != "0":
                ops["node"] = node_leg(4096, int(os.environ.get("ND4_BENCH_NODE_SVD_BATCH", "1024")))
            if world == 1 and not args.no_cpu_baseline:
                # the CPU port beside every side op, and its results as one more parity check of the device results
                cpu_ops, (p_cpu, r_cpu, sv_cpu) = cpu_baseline_ops(2048)
                out["cpu_baseline"]["ops"] = cpu_ops
                try:
                    cpu_ops["svd_batch_all_cores"] = cpu_baseline_svd_batch(64)
                except Exception as ex:  # pragma: no cover
                    cpu_ops["svd_batch_all_cores"] = {"error": repr(ex)}
                from nd4js_amd import dev
                A7 = dev.fill_uniform(7, (2048, 2048))
                LUd, Pd = dev.lu_decomp(A7)
                parity["lu2048_P_identical_to_oracle"] = bool(np.array_equal(Pd.cpu().numpy(), p_cpu))
                gates["lu2048_P_mismatches"] = (0.0 if parity["lu2048_P_identical_to_oracle"] else 1.0, 0.0)
                Qd, Rd = dev.qr_decomp(A7)
                parity["qr2048_R_vs_oracle_relerr"] = float(np.linalg.norm(Rd.cpu().numpy() - r_cpu) / np.linalg.norm(r_cpu))
                gates["qr2048_R_vs_oracle_relerr"] = (parity["qr2048_R_vs_oracle_relerr"], 1e-10)
                X0 = dev.fill_uniform(1000, (512, 512))
                sv0 = dev.svd_decomp(X0)[1].cpu().numpy()
                parity["svd512_sv_vs_oracle_max_rel"] = float(np.abs(sv0 - sv_cpu).max() / sv_cpu.max())
                gates["svd512_sv_vs_oracle_max_rel"] = (parity["svd512_sv_vs_oracle_max_rel"], 1e-10)
        bad = {k: v for k, (v, lim) in gates.items() if not (v <= lim)}          # NaN fails too
        out["gates"] = {k: {"value": v, "limit": lim} for k, (v, lim) in gates.items()}
        out["valid"] = not bad and not failed
        if ops_error or failed:
            out["ops_error"] = ops_error or "a side measurement failed on another rank"
        if bad:
            out["failed_gates"] = bad
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
        if not out["valid"]:
            raise SystemExit(3)                            # a wrong result or a failed side measurement is not a benchmark


if __name__ == "__main__":
    main()
