"""Gaussian-sharded splat forward across the GPUs of one node.

The base splat is linear in the Gaussian set: ``logits = sum_g contrib_g``.  Each rank
splats a contiguous slice of the Gaussians into its own full ``[N,18]`` fp32 grid and the
partial grids are summed with ONE all-reduce (``torch.distributed`` backend ``"nccl"`` is
RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).  This is an addition over the reference,
which only runs data-parallel replicas (train.py:41-43, :86-91).

The prob variant's outputs are ratios / products and are not sharded here (it would need
the un-normalised numerator, ``prob_sum``, ``density`` and ``sum log(1-e)`` reduced before
normalisation).
"""
import torch
import torch.distributed as dist


def shard_bounds(P, rank, world_size):
    """Contiguous, balanced slice [lo, hi) of P Gaussians for ``rank`` (ascending-id order
    is preserved inside a shard)."""
    base, rem = divmod(P, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_splat_forward(local_splat, pts, means3D, opacities, semantics, scales, cov3D, group=None):
    """``local_splat(pts, means3D, opacities, semantics, scales, cov3D) -> logits [N,18]`` is
    the single-GPU op (``local_aggregate.LocalAggregator``); Gaussian arguments carry the
    leading batch dim of 1 like the reference's.  Returns the full logits on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    P = means3D.shape[1]
    lo, hi = shard_bounds(P, rank, world)
    logits = local_splat(pts, means3D[:, lo:hi], opacities[:, lo:hi], semantics[:, lo:hi], scales[:, lo:hi],
                         cov3D[:, lo:hi])
    if world > 1:
        logits = logits.contiguous()
        dist.all_reduce(logits, op=dist.ReduceOp.SUM, group=group)
    return logits
