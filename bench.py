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


def cpu_baseline_matmul(n, rows):
    """Oracle (CPU port of matmul.js:49-53) on a bounded sample: the first `rows` rows of C, i.e.
    the same i-k-j loop streaming all of B, 1 thread (the reference is single-threaded)."""
    import numpy as np
    import oracle
    from nd4js_amd import rng
    a = rng.matrix(5, n, n)[:rows].copy()
    b = rng.matrix(6, n, n)
    t = time.perf_counter()
    c = oracle.matmul2(a, b)
    dt = time.perf_counter() - t
    return {"value": round(2.0 * rows * n * n / dt / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": "rows 0..%d of the %dx%d product (oracle/nd4_oracle.c i-k-j loop, %.1f s)" % (rows - 1, n, n, dt),
            "host_cpus": os.cpu_count()}, c


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

    out = None
    if rank == 0:
        achieved = flops / (kernel_ms * 1e-3) / 1e12
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
        try:
            pmc = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.startswith("r"))
            for rnd in reversed(pmc):
                f = os.path.join(ROOT, "profiles", rnd, "gemm4096_pmc.json")
                if os.path.exists(f) and n == 4096:
                    with open(f) as fh:
                        out["roofline"]["traffic"] = json.load(fh)["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = "profiles/%s/gemm4096_pmc.json (FETCH_SIZE x2 + WRITE_SIZE, KB->B)" % rnd
                    break
        except Exception:  # pragma: no cover
            pass
        # parity gate printed with the number
        try:
            import numpy as np
            with open(os.path.join(ROOT, "tests", "golden", "manifest.json")) as f:
                g = json.load(f)["cases"].get("c2_matmul4096")
            if g and n == 4096:
                idx = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", g["files"]["idx"])).astype("int64")).cuda()
                val = np.load(os.path.join(ROOT, "tests", "golden", g["files"]["val"]))
                got = C.reshape(-1)[idx].cpu().numpy()
                out["parity"] = {"matmul_vs_reference_samples_relerr": float(np.linalg.norm(got - val) / np.linalg.norm(val)), "gate": 1e-10}
        except Exception as e:  # pragma: no cover
            out["parity"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                import numpy as np
                cb, c_cpu = cpu_baseline_matmul(n, min(2048, n))
                out["cpu_baseline"] = cb
                out.setdefault("parity", {})["matmul_vs_oracle_rows_relerr"] = float(
                    np.linalg.norm(C[: c_cpu.shape[0]].cpu().numpy() - c_cpu) / np.linalg.norm(c_cpu))
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(e)}
        if not args.no_ops:
            try:
                import bench_ops
                out["ops"] = bench_ops.run(world, rank, local, dist)
            except ImportError:
                pass
            except Exception as e:  # pragma: no cover
                out["ops"] = {"error": repr(e)}
    elif not args.no_ops:
        try:
            import bench_ops
            bench_ops.run(world, rank, local, dist)
        except ImportError:
            pass
        except Exception:
            pass
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
