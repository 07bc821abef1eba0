#!/usr/bin/env python3
"""*** TEST INFRASTRUCTURE *** — the CPU baseline of BASELINE configs[4] on ALL host cores.

  python oracle/cpu_batch.py --count 16 --workers 16 --n 512

Runs the oracle's restatement of the reference's `svd_decomp` (= `svd_dc`: bidiagonalisation + divide & conquer,
oracle/nd4_oracle_svd_dc.c restating svd_dc.js:37-932; `--algo jacobi`: the two-sided Jacobi of svd_jac_2sided.js:95-134) on
`count` members of the 1024 x 512^2 batch (seeds 1000 + i, the same synthetic inputs bench.py uses), spread over `workers`
processes, and prints one JSON line with the wall time. bench.py starts it as a
child process (it must not fork after the GPU has been initialised) and only as the reported CPU baseline, never as a
product path.
"""
import argparse
import json
import os
import sys
import time
from multiprocessing import Pool

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _one(args):
    member, n, algo = args
    import numpy as np
    import oracle
    a = np.asarray(oracle.fill_uniform(1000 + member, n * n), dtype=np.float64).reshape(n, n)
    t = time.perf_counter()
    if algo == "jacobi":
        _, sv, _, sweeps = oracle.svd_jac_2sided(a)
    else:
        _, sv, _ = oracle.svd_dc(a)
        sweeps = 0
    return time.perf_counter() - t, int(sweeps), float(sv[0])


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--count", type=int, default=16)
    p.add_argument("--workers", type=int, default=0)
    p.add_argument("--n", type=int, default=512)
    p.add_argument("--algo", default="dc", choices=["dc", "jacobi"])
    a = p.parse_args()
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = a.workers if a.workers > 0 else avail
    workers = max(1, min(workers, a.count))
    import oracle
    oracle.build()
    t = time.perf_counter()
    with Pool(workers) as pool:
        res = pool.map(_one, [(i, a.n, a.algo) for i in range(a.count)], chunksize=1)
    wall = time.perf_counter() - t
    per = [r[0] for r in res]
    print(json.dumps({"count": a.count, "workers": workers, "n": a.n, "wall_seconds": round(wall, 3),
                      "seconds_per_matrix_mean": round(sum(per) / len(per), 3), "seconds_per_matrix_max": round(max(per), 3),
                      "matrices_per_s": round(a.count / wall, 3), "sweeps_max": max(r[1] for r in res), "algo": a.algo,
                      "cpus_available": avail, "cpus_total": os.cpu_count(), "cpu_model": cpu_model()}))


if __name__ == "__main__":
    main()
