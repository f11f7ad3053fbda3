"""Head epilogue: labels from the splat outputs, and the sharded labels path
(2-rank gloo on the CPU with the per-rank ops stubbed, GPU kernel against the torch formulas
of model/head/gaussian_head.py:164-185)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _reference_labels(logits, bin_logits=None, threshold=0.5, empty_label=17, combine_geosem=False):
    """gaussian_head.py:164-185, op for op (semantics = (logits, bin_logits, ...))."""
    if bin_logits is not None:
        if combine_geosem:
            sem = logits[:, :-1] * bin_logits.unsqueeze(-1)
            geo = 1 - bin_logits.unsqueeze(-1)
            geosem = torch.cat([sem, geo], dim=-1)
        else:
            geosem = logits
        prediction = geosem[None].transpose(1, 2)
        if not combine_geosem:
            final_semantics = prediction.argmax(dim=1)
            final_occupancy = bin_logits[None] > threshold
            final_prediction = torch.ones_like(final_semantics) * empty_label
            final_prediction[final_occupancy] = final_semantics[final_occupancy]
            return final_prediction[0]
        return prediction.argmax(dim=1)[0]
    return logits[None].transpose(1, 2).argmax(dim=1)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 63, 64, 1000, 640000])
def test_head_labels_match_torch(N):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.head import occupancy_labels
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N)
    logits = torch.randn(N, 18, generator=g).to(dev)
    bins = torch.rand(N, generator=g).to(dev)
    # ties: duplicate the maximum of some rows into a later channel -> the first one must win
    if N >= 64:
        rows = torch.arange(0, N, 7, device=dev)
        m, am = logits[rows].max(dim=1)
        later = torch.clamp(am + 3, max=17)
        logits[rows, later] = m
    for kw in (dict(), dict(bin_logits=bins, threshold=0.4, empty_label=17), dict(bin_logits=bins, combine_geosem=True)):
        got = occupancy_labels(logits, **kw)
        want = _reference_labels(logits, **kw)
        assert got.dtype == torch.int64 and got.shape == (N,)
        assert torch.equal(got, want), kw
    # a non-16-byte-aligned view takes the scalar load path
    if N >= 64:
        flat = torch.zeros(N * 18 + 1, device=dev)
        view = flat[1:].view(N, 18)
        view.copy_(logits)
        assert torch.equal(occupancy_labels(view), _reference_labels(logits))


# ------------------------------------------------------------------ sharded labels, 2-rank gloo
def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussianformer_amd.head import sharded_splat_labels
    from test_sharded_cpu import _inputs, _local_splat_factory
    si, args = _inputs()
    labels = sharded_splat_labels(_local_splat_factory(si), *args, labels_fn=lambda lg: lg.argmax(dim=1))
    q.put((rank, labels.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_labels_match_single_rank():
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gaussianformer_amd.head import sharded_splat_labels
    from test_sharded_cpu import _inputs, _local_splat_factory
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    si, args = _inputs()
    logits = _local_splat_factory(si)(*args)
    want = logits.argmax(dim=1).numpy()
    # N = 16*12*8 = 1536 points is even; summation order differs between 1 and 2 ranks, so allow
    # a flip only where the top two logits are within rounding
    top2 = torch.topk(logits, 2, dim=1).values
    close = (top2[:, 0] - top2[:, 1]).abs().numpy() < 1e-5
    for r in range(2):
        assert got[r].shape == want.shape
        assert np.all((got[r] == want) | close)
    assert np.array_equal(got[0], got[1])
    # world_size 1 path (no process group): plain labels
    assert np.array_equal(sharded_splat_labels(_local_splat_factory(si), *args, labels_fn=lambda lg: lg.argmax(dim=1)).numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("config,per_axis,dense", [("nuscenes_gs25600_solid", False, True), ("prob_gs6400", False, True),
                                                   ("prob_gs6400", True, False), ("nuscenes_gs144000", False, False)])
def test_fused_label_epilogue_matches_separate_kernels(config, per_axis, dense):
    """gf_splat_forward_labels (labels from the render kernel's accumulators, with and without
    writing the logits) == gf_splat_forward followed by gf_head_labels, for the dense grid and
    for arbitrary points, base and prob heads."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gaussianformer_amd import _lib
    from gaussianformer_amd.head import occupancy_labels
    from gaussianformer_amd.local_aggregate import splat_forward, splat_forward_labels
    from gaussianformer_amd.synthetic import make_splat_inputs
    from util import prep, to_dev
    dev = torch.device("cuda:0")
    si = make_splat_inputs(config, seed=51, P=900, H=40, W=36, D=16, dense_pts=dense, N=None if dense else 5000)
    pi, mi, radii, cov6 = prep(si, per_axis)
    t = to_dev(dev, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)
    prob = si.variant == "prob"
    variant = _lib.GF_SPLAT_PROB if prob else _lib.GF_SPLAT_BASE
    # the label epilogue lives in the exact-fp32 kernels and, for the base head on the dense grid, in the wave-autonomous
    # matrix-core kernel (the default): each compared with the same kernel's separate forward, labels AND logits bit for bit
    flag_sets = [_lib.GF_EXACT_FP32] + ([0] if (not prob and dense) else [])
    for fl in flag_sets:
        logits, bl, de, pr, state = splat_forward(variant, *t, si.H, si.W, si.D, flags=fl)
        if fl == 0:
            assert state.view(torch.int32)[1].item() == _lib.GF_PATH_MATRIX_CORE_WAVE
        for kw in ([dict()] if not prob else [dict(threshold=0.3), dict(combine_geosem=True)]):
            want = occupancy_labels(logits, bin_logits=bl, empty_label=17, **kw)
            got = splat_forward_labels(variant, *t, si.H, si.W, si.D, flags=fl, **kw)
            assert torch.equal(got, want), (fl, kw)
            outs = splat_forward_labels(variant, *t, si.H, si.W, si.D, keep_logits=True, flags=fl, **kw)
            bits = lambda x: x.view(torch.int32)   # bitwise: the prob config can produce NaN (negative fp32 determinant)
            assert torch.equal(outs[0], want) and torch.equal(bits(outs[1]), bits(logits))
            if prob:
                assert all(torch.equal(bits(a), bits(b)) for a, b in zip(outs[2:], (bl, de, pr)))


@pytest.mark.gpu
def test_prob_pieces_of_two_shards_equal_the_full_forward():
    """GF_PROB_NUMERATOR / LocalAggregatorProb.forward_pieces: the pieces of two Gaussian shards, combined
    the way sharded_splat_forward_prob combines them (sum, sum, sum, product, then normalise), equal the
    single forward -- including the voxels no Gaussian reaches (uniform fallback)."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import local_aggregate_prob
    from gaussianformer_amd.sharded import normalise_prob, shard_bounds
    from gaussianformer_amd.synthetic import make_splat_inputs
    dev = torch.device("cuda:0")
    si = make_splat_inputs("prob_gs6400", seed=41, P=900, H=40, W=36, D=16)
    agg = local_aggregate_prob.LocalAggregator(si.scale_multiplier, si.H, si.W, si.D, list(si.pc_min), si.grid_size).to(dev)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev)[None] for a in (si.pts, si.means3D, si.opacities, si.semantics, si.scales, si.cov3D)]
    logits, bin_logits, density = agg(*t)
    parts = []
    for r in range(2):
        lo, hi = shard_bounds(t[1].shape[1], r, 2)
        parts.append(agg.forward_pieces(t[0], *[x[:, lo:hi] for x in t[1:]]))
    num = parts[0][0] + parts[1][0]
    keep = (1 - parts[0][1]) * (1 - parts[1][1])
    dens = parts[0][2] + parts[1][2]
    psum = parts[0][3] + parts[1][3]
    got_logits, got_bin, got_dens = normalise_prob(num, keep, dens, psum)
    assert torch.allclose(got_logits, logits, rtol=1e-4, atol=1e-5)
    assert torch.allclose(got_bin, bin_logits, rtol=1e-5, atol=1e-6)
    assert torch.allclose(got_dens, density, rtol=1e-5, atol=1e-6)
    # one shard = the plain forward, bit for bit after normalisation of its own pieces
    one = agg.forward_pieces(*t)
    l1, b1, d1 = normalise_prob(one[0], 1 - one[1], one[2], one[3])
    assert torch.equal(b1, bin_logits) and torch.equal(d1, density)
    assert torch.allclose(l1, logits, rtol=1e-6, atol=1e-7)
