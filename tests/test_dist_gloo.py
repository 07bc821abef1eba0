"""world_size=2 gloo test of the batch-sharding path (nd4js_amd/dist.py) on CPU: the collective
plumbing (health all-reduce, uneven all-gather of sv) with the ORACLE standing in as the per-rank
compute (test infrastructure only; on GPUs the compute is the HIP path, covered by -m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from nd4js_amd import rng
from nd4js_amd.dist import shard, shard_sizes, svd_decomp_sharded


def test_shard_partition_is_contiguous_and_complete():
    for batch in (1, 2, 7, 8, 1024, 1023):
        for world in (1, 2, 3, 8):
            edges = [shard(batch, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = shard_sizes(batch, world)
            assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1


def _cpu_compute(A, info=None):
    u, sv, v, sweeps = oracle.svd_jac_2sided(A.numpy())
    if info is not None:
        info["sweeps"], info["offnorm"] = sweeps, 0.0
    return torch.from_numpy(u), torch.from_numpy(sv), torch.from_numpy(v)


def _worker(rank, world, port, batch, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard(batch, world, rank)
        A = torch.from_numpy(np.stack([rng.matrix(1000 + b, n, n) for b in range(lo, hi)]))
        U, sv_all, V, health = svd_decomp_sharded(A, batch, compute=_cpu_compute)
        assert sv_all.shape == (batch, n) and U.shape == (hi - lo, n, n)
        assert health["max_sweeps"] >= 1 and not health["failed"]
        np.save(os.path.join(out_dir, "sv_rank%d.npy" % rank), sv_all.numpy())
        # R3: U and V gathered whole on every rank (uneven blocks, padding trimmed)
        Ua, _, Va, _ = svd_decomp_sharded(A, batch, compute=_cpu_compute, gather_uv=True)
        assert Ua.shape == (batch, n, n) and Va.shape == (batch, n, n)
        assert torch.equal(Ua[lo:hi], U) and torch.equal(Va[lo:hi], V)
        np.save(os.path.join(out_dir, "u_rank%d.npy" % rank), Ua.numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_svd_world2_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    batch, n, world = 5, 12, 2                    # uneven split: 3 + 2
    mp.spawn(_worker, args=(world, port, batch, n, str(tmp_path)), nprocs=world, join=True)
    sv0 = np.load(tmp_path / "sv_rank0.npy")
    sv1 = np.load(tmp_path / "sv_rank1.npy")
    assert np.array_equal(sv0, sv1)               # every rank holds the whole gathered result
    ref = np.stack([oracle.svd_jac_2sided(rng.matrix(1000 + b, n, n))[1] for b in range(batch)])
    assert np.array_equal(sv0, ref)               # sharding changed nothing: bit-identical to the serial run
    u0, u1 = np.load(tmp_path / "u_rank0.npy"), np.load(tmp_path / "u_rank1.npy")
    assert np.array_equal(u0, u1)                 # R3: both ranks hold the same whole U
    assert np.array_equal(u0[4], oracle.svd_jac_2sided(rng.matrix(1000 + 4, n, n))[0])


def _failing_worker(rank, world, port, batch, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nd4js_amd.dist import PeerFailed

    def compute(A, info=None):
        if rank == 1:
            raise RuntimeError("nd4hip error -4: no convergence")      # what ND4HIP_ERR_NOCONV looks like on the host
        return _cpu_compute(A, info)
    try:
        lo, hi = shard(batch, world, rank)
        A = torch.from_numpy(np.stack([rng.matrix(1000 + b, n, n) for b in range(lo, hi)]))
        try:
            svd_decomp_sharded(A, batch, compute=compute)
            outcome = "returned"
        except PeerFailed:
            outcome = "peer_failed"
        except RuntimeError as e:
            outcome = "own:" + str(e)
        dist.barrier()                          # both ranks are still in step: nobody is stuck in the all-gather
        with open(os.path.join(out_dir, "outcome%d.txt" % rank), "w") as f:
            f.write(outcome)
    finally:
        dist.destroy_process_group()


def test_failure_on_one_rank_raises_on_all_ranks(tmp_path):
    """a rank whose local decomposition raises must not leave the healthy ranks blocked in R2."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_failing_worker, args=(2, port, 4, 8, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "outcome0.txt").read_text() == "peer_failed"
    assert (tmp_path / "outcome1.txt").read_text().startswith("own:nd4hip error -4")
