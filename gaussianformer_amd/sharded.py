"""Splat forward across the GPUs of one node: the Gaussian-sharded partition north_star names (partial
grids + one all-reduce) and the spatial slab partition that needs no reduction (``slab_splat_forward``).

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


EXCHANGES = ("all_reduce", "direct", "reduce_scatter")


def sum_across_ranks(t, group=None, how="all_reduce"):
    """Sum of every rank's ``t`` (contiguous, same shape everywhere), left in ``t`` on every rank.  Three ways to move the
    46 MB of partial logits:

    ``"all_reduce"``      one RCCL all-reduce (what north_star names; RCCL picks the algorithm -- a ring over the node's
                          xGMI links is bound by ONE link, 7/4 of the tensor through it);
    ``"direct"``          the exchange the fully connected xGMI topology suggests: the tensor in ``world``
                          chunks, chunk j of every rank sent straight to rank j (``all_to_all_single``: ``world - 1``
                          point-to-point transfers per rank, one per link, all links busy at once), summed there in rank
                          order by one kernel, and the reduced chunks all-gathered the same way -- 2/world of the tensor per
                          link instead of a ring's 2 (world - 1)/world through every link in turn; every chunk is reduced by
                          exactly one rank in a fixed order, so all ranks hold the same bits;
    ``"reduce_scatter"``  RCCL's own reduce-scatter + all-gather.

    Which one wins is a property of the node and the RCCL build: ``bench.py --gpus N`` times all three in its warm-up and
    reports the choice.  Sizes that ``world`` does not divide are padded for the chunked forms."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return t
    if how == "all_reduce":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t
    if how not in EXCHANGES:
        raise ValueError(f"exchange {how!r}: one of {EXCHANGES}")
    flat = t.view(-1)
    n = flat.numel()
    chunk = (n + world - 1) // world
    if chunk * world != n:
        padded = flat.new_zeros(chunk * world)
        padded[:n] = flat
    else:
        padded = flat
    mine = flat.new_empty(chunk)
    if how == "direct":
        recv = flat.new_empty(world * chunk)
        dist.all_to_all_single(recv, padded, group=group)       # recv[r] = rank r's partial of MY chunk
        torch.sum(recv.view(world, chunk), dim=0, out=mine)
    else:
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM, group=group)
    # (the partial grid has been sent / reduced: the gathered sum lands where it was -- no second buffer, no copy, when world divides n)
    dist.all_gather_into_tensor(padded, mine, group=group)
    if padded is not flat:
        flat.copy_(padded[:n])
    return t


def sharded_splat_forward(local_splat, pts, means3D, opacities, semantics, scales, cov3D, group=None, exchange="all_reduce"):
    """``local_splat(pts, means3D, opacities, semantics, scales, cov3D) -> logits [N,18]`` is
    the single-GPU op (``local_aggregate.LocalAggregator``); Gaussian arguments carry the
    leading batch dim of 1 like the reference's.  Returns the full logits on every rank; ``exchange``: how the partial
    grids are summed (``sum_across_ranks``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    P = means3D.shape[1]
    lo, hi = shard_bounds(P, rank, world)
    logits = local_splat(pts, means3D[:, lo:hi], opacities[:, lo:hi], semantics[:, lo:hi], scales[:, lo:hi],
                         cov3D[:, lo:hi])
    if world > 1:
        logits = sum_across_ranks(logits.contiguous(), group, exchange)
    return logits


def slab_bounds(H, rank, world_size, align=8):
    """Voxel rows ``[x0, x1)`` of ``rank``'s slab of an H-row grid: the rows in blocks of ``align`` (8 = the supertile,
    the granule the forward bins Gaussians by -- a slab that starts on it keeps the full call's tiles, so its rows come
    out bit-identical to the single-GPU result with either render kernel), blocks dealt contiguously and balanced."""
    lo, hi = shard_bounds((H + align - 1) // align, rank, world_size)
    return min(lo * align, H), min(hi * align, H)


def _own_slab(H, group):
    """``(rank, world, x0, x1)`` of the calling rank.  The bounds of EVERY rank are evaluated on every rank and an empty
    slab anywhere raises everywhere: a rank that raised alone would leave the others waiting in the all-gather."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds = [slab_bounds(H, r, world) for r in range(world)]
    empty = [r for r, (a, b) in enumerate(bounds) if b <= a]
    if empty:
        raise ValueError(f"ranks {empty} of {world} own no rows of an {H}-row grid (8-row blocks): use at most "
                         f"{(H + 7) // 8} ranks")
    return rank, world, bounds[rank][0], bounds[rank][1]


def slab_splat_forward(aggregator, pts, means3D, opacities, semantics, scales, cov3D, group=None, gather=True):
    """SPATIAL partition of one frame's splat over the ranks (SURVEY.md §8e "slab partition with halo Gaussians"): rank r
    renders the voxel rows ``slab_bounds(H, r, world)`` with ``aggregator.forward_slab`` -- every Gaussian is passed,
    the ones whose box misses the slab are clipped away inside the op (no device-side compaction, no host read) -- and
    the slabs are ALL-GATHERED.  No reduction: a voxel's value is computed by exactly one rank from the same Gaussians
    in the same order as in the single-GPU call, so the gathered grid equals the single-GPU grid bit for bit, and the
    exchange is each rank's own slab (46 MB / world of fp32 logits) instead of a 46 MB all-reduce -- or, through
    ``slab_splat_labels``, 8 bytes per voxel of labels.  Works for the prob aggregators too (their ratio and product
    never cross a slab).  ``aggregator`` is a ``LocalAggregator*`` built for the FULL grid; ``gather=False`` returns the
    rank's own slab ``(x0, x1, outputs)``, which is differentiable like ``aggregator.forward`` (training: a per-voxel
    loss is a sum over slabs, only the Gaussians' gradients -- 112 B each -- need an all-reduce).  With ``gather=True`` it
    returns the VALUES ``aggregator.forward`` returns for the full grid: the all-gather is inference-only, its result
    carries no autograd history (asserted: gathered outputs never require grad)."""
    rank, world, x0, x1 = _own_slab(aggregator.H, group)
    out = aggregator.forward_slab(x0, x1, pts, means3D, opacities, semantics, scales, cov3D)
    if not gather:
        return x0, x1, out
    if world == 1:
        return out
    return _gather_slabs(out, aggregator, world, group)


def _gather_slabs(out, aggregator, world, group):
    """All-gather of per-rank slabs of unequal height: padded to the tallest slab, trimmed on arrival.  Values only:
    the inputs are detached (a collective has no backward here), so nothing downstream can silently drop a gradient
    it believes it has -- differentiate the per-rank slab (``gather=False``) instead."""
    out = out.detach() if not isinstance(out, (tuple, list)) else tuple(t.detach() for t in out)
    H, plane = aggregator.H, aggregator.W * aggregator.D
    bounds = [slab_bounds(H, r, world) for r in range(world)]
    tallest = max(b - a for a, b in bounds) * plane
    single = not isinstance(out, (tuple, list))
    parts = [out] if single else list(out)
    gathered = []
    for t in parts:
        mine = t.new_zeros((tallest,) + tuple(t.shape[1:]))
        mine[:t.shape[0]] = t
        if t.is_cuda and dist.get_backend(group) == "gloo":   # test mode (ranks sharing one GPU): collective on host copies
            host = torch.empty((world * tallest,) + tuple(t.shape[1:]), dtype=t.dtype)
            dist.all_gather_into_tensor(host, mine.cpu().contiguous(), group=group)
            buf = host.to(t.device)
        else:
            buf = t.new_empty((world * tallest,) + tuple(t.shape[1:]))
            dist.all_gather_into_tensor(buf, mine.contiguous(), group=group)
        gathered.append(torch.cat([buf[r * tallest:r * tallest + (b - a) * plane] for r, (a, b) in enumerate(bounds)], dim=0))
    return gathered[0] if single else tuple(gathered)


def slab_splat_labels(aggregator, labels_fn, pts, means3D, opacities, semantics, scales, cov3D, group=None):
    """Slab-partitioned inference that ends in labels: every rank labels its own slab (``labels_fn(outputs) -> int64 [n]``,
    e.g. ``head.occupancy_labels``) and only the labels are all-gathered (5 MB for the 640 000 voxels)."""
    rank, world, x0, x1 = _own_slab(aggregator.H, group)
    out = aggregator.forward_slab(x0, x1, pts, means3D, opacities, semantics, scales, cov3D)
    labels = labels_fn(out)
    if world == 1:
        return labels
    return _gather_slabs(labels, aggregator, world, group)


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
