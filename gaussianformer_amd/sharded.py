"""Gaussian-sharded splat forward across the GPUs of one node.

The base splat is linear in the Gaussian set: ``logits = sum_g contrib_g``.  Each rank
splats a contiguous slice of the Gaussians into its own full ``[N,18]`` fp32 grid and the
partial grids are summed with ONE all-reduce (``torch.distributed`` backend ``"nccl"`` is
RCCL over xGMI on ROCm; ``gloo`` in the CPU tests).  This is an addition over the reference,
which only runs data-parallel replicas (train.py:41-43, :86-91).

The prob variant's outputs are a ratio and a product, so its shards exchange the pieces and
normalise afterwards (``sharded_splat_forward_prob``): the un-normalised numerator
``sum sem * prob`` (``GF_PROB_NUMERATOR``), ``probability`` and ``density`` add up,
``1 - bin_logits = prod (1 - e)`` multiplies.
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


def normalise_prob(numerator, keep, density, probability):
    """Epilogue of the prob splat on reduced pieces (localagg_prob/src/forward.cu:92-101):
    ``logits = numerator / probability`` where ``probability > 1e-9``, else ``1/17`` in the first
    17 channels and 0 in the last; ``bin_logits = 1 - keep``."""
    C = numerator.shape[1]
    fallback = torch.full((C,), 1.0 / (C - 1), dtype=numerator.dtype, device=numerator.device)
    fallback[C - 1] = 0.0
    ok = (probability.double() > 1e-9)[:, None]
    logits = torch.where(ok, numerator / probability[:, None], fallback[None, :])
    return logits, 1 - keep, density


def sharded_splat_forward_prob(local_splat_pieces, pts, means3D, opacities, semantics, scales, cov3D, group=None):
    """Gaussian-sharded forward of the prob variant (inference).  ``local_splat_pieces(pts, means3D,
    opacities, semantics, scales, cov3D) -> (numerator [N,18], bin_logits [N], density [N], probability [N])``
    is the single-GPU op with ``GF_PROB_NUMERATOR`` (``LocalAggregatorProb.forward_pieces``).  The
    numerator, density and probability of the shards are summed in ONE all-reduce of a packed
    ``[N,20]`` buffer, ``1 - bin_logits`` is multiplied in a second one, then every rank normalises.
    Returns ``(logits, bin_logits, density)`` like ``LocalAggregatorProb.forward``."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    P = means3D.shape[1]
    lo, hi = shard_bounds(P, rank, world)
    numerator, bin_logits, density, probability = local_splat_pieces(
        pts, means3D[:, lo:hi], opacities[:, lo:hi], semantics[:, lo:hi], scales[:, lo:hi], cov3D[:, lo:hi])
    packed = torch.cat([numerator, density[:, None], probability[:, None]], dim=1).contiguous()
    keep = (1 - bin_logits).contiguous()
    if world > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(keep, op=dist.ReduceOp.PRODUCT, group=group)
    C = numerator.shape[1]
    return normalise_prob(packed[:, :C], keep, packed[:, C], packed[:, C + 1])
