"""Multi-GPU form of the batched decompositions: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), the reference's leading batch axis (qr.js:43-49, lu.js:34-40,
svd_dc.js:918-925 loop over independent matrices) cut into contiguous blocks, one per rank.

There is NO data-path collective: matrices never interact. The only exchanges are
  R1  all-reduce(MAX) of [sweeps, off-norm, failed]      (convergence / health, 24 bytes)
  R2  all-gather of the singular values                   (batch x L doubles in total)
Single matrices do not shard ("replicas only").
"""
import torch


class PeerFailed(RuntimeError):
    """Raised on the healthy ranks when some other rank's local work failed (all ranks leave the collective sequence together)."""


def shard(batch, world, rank):
    """[lo, hi) of the batch axis owned by `rank`: contiguous blocks, remainder to the low ranks."""
    per, rem = divmod(int(batch), int(world))
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def shard_sizes(batch, world):
    return [shard(batch, world, r)[1] - shard(batch, world, r)[0] for r in range(world)]


def gather_blocks(x_local, batch_total, group=None):
    """R3: the [batch_total, ...] tensor assembled on EVERY rank from the ranks' contiguous blocks — for a
    device-resident result that is wanted whole (U, V: 2 GiB each for 1024 x 512^2, a 256 MiB shard per GPU of an 8-GPU node:
    with 7 direct xGMI links a fully connected all-gather moves each shard over its own link, ~1.8 ms per tensor at 153 GB/s).
    One padded all_gather_into_tensor (blocks differ by at most one member), padding trimmed."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return x_local
    cdev = x_local.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sizes = shard_sizes(batch_total, world)
    pad = max(sizes)
    tail = tuple(x_local.shape[1:])
    mine = torch.zeros((pad,) + tail, dtype=x_local.dtype, device=cdev)
    mine[: x_local.shape[0]] = x_local
    buf = torch.empty((world * pad,) + tail, dtype=x_local.dtype, device=cdev)
    dist.all_gather_into_tensor(buf, mine, group=group)
    return torch.cat([buf[r * pad: r * pad + sizes[r]] for r in range(world)]).to(x_local.device)


def svd_decomp_sharded(A_local, batch_total, group=None, compute=None, gather_uv=False):
    """A_local: this rank's block [b_local, M, N] (device tensor). Returns (U_local, sv_all, V_local, health)
    where sv_all is the [batch_total, L] result gathered on every rank and health = dict(max_sweeps,
    max_offnorm, failed) reduced over all ranks. `compute` defaults to the GPU path. gather_uv=True also gathers U and V
    (R3): every rank then returns the full [batch_total, ...] U and V instead of its block."""
    import torch.distributed as dist
    if compute is None:
        from . import dev
        compute = dev.svd_decomp
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard(batch_total, world, rank)
    assert A_local.shape[0] == hi - lo, "A_local must be this rank's contiguous block of the batch"
    info, error = {}, None
    try:
        U, sv, V = compute(A_local, info=info)
    except Exception as e:          # joins R1 first: the other ranks must learn of the failure before anyone enters R2
        error = e
    # RCCL moves device tensors directly; any other backend (gloo: CPU tests, single-GPU rehearsal) is staged
    # through host memory
    cdev = A_local.device if (world > 1 and dist.get_backend(group) == "nccl") else torch.device("cpu")
    health = torch.tensor([float(info.get("sweeps", 0)), float(info.get("offnorm") or 0.0), 0.0 if error is None else 1.0],
                          dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(health, op=dist.ReduceOp.MAX, group=group)               # R1
    if error is not None:
        raise error
    if health[2].item() != 0.0:
        # a peer failed (e.g. ND4HIP_ERR_NOCONV, out of memory): it never joins R2, so nobody may enter it
        raise PeerFailed("svd_decomp_sharded: another rank failed in its local decomposition; results of rank %d discarded" % rank)
    if world > 1:
        # equal-size exchange (blocks differ by at most one matrix): pad to the largest block, gather
        # into one flat buffer (one collective, fully-connected over xGMI), trim the padding
        L = sv.shape[-1]
        sizes = shard_sizes(batch_total, world)
        pad = max(sizes)
        mine = torch.zeros((pad, L), dtype=sv.dtype, device=cdev)
        mine[: sv.shape[0]] = sv
        buf = torch.empty((world * pad, L), dtype=sv.dtype, device=cdev)
        dist.all_gather_into_tensor(buf, mine, group=group)                           # R2
        sv_all = torch.cat([buf[r * pad: r * pad + sizes[r]] for r in range(world)]).to(sv.device)
    else:
        sv_all = sv
    if gather_uv:
        U, V = gather_blocks(U, batch_total, group), gather_blocks(V, batch_total, group)           # R3
    return U, sv_all, V, {"max_sweeps": int(health[0].item()), "max_offnorm": health[1].item(), "failed": bool(health[2].item())}
