"""The spatial (slab) partition on the real kernels: ``LocalAggregator*.forward_slab`` renders a band of voxel rows from
ALL Gaussians (boxes clipped to the band like getRect clips them to the grid); bands that start on a multiple of 8 rows
reproduce the full call's rows bit for bit -- with the matrix-core kernel and with the exact-fp32 kernel, base and prob --
and their gradients add up to the full call's."""
import numpy as np
import pytest

from gaussianformer_amd.synthetic import make_splat_inputs

pytestmark = pytest.mark.gpu


def _module(config, si, gpu, **kw):
    import local_aggregate, local_aggregate_prob
    pkg = local_aggregate_prob if si.variant == "prob" else local_aggregate
    return pkg.LocalAggregator(si.scale_multiplier, si.H, si.W, si.D, list(si.pc_min), si.grid_size, **kw).to(gpu)


@pytest.mark.parametrize("config,kw", [("nuscenes_gs25600_solid", {}), ("nuscenes_gs25600_solid", {"matrix_cores": False}),
                                       ("nuscenes_gs144000", {}), ("prob_gs6400", {})])
def test_slabs_equal_the_full_call_bit_for_bit(gpu, config, kw):
    import torch
    from gaussianformer_amd.sharded import slab_bounds
    si = make_splat_inputs(config, seed=61, P=1500, H=40, W=24, D=16)
    agg = _module(config, si, gpu, **kw)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(gpu)[None] for a in (si.pts, si.means3D, si.opacities, si.semantics, si.scales, si.cov3D)]
    full = agg(*t)
    full = full if isinstance(full, tuple) else (full,)
    plane = si.W * si.D
    bits = lambda x: x.contiguous().view(torch.int32)
    for world in (2, 3):
        for r in range(world):
            x0, x1 = slab_bounds(si.H, r, world)
            out = agg.forward_slab(x0, x1, *t)
            out = out if isinstance(out, tuple) else (out,)
            for a, b in zip(out, full):
                assert torch.equal(bits(a), bits(b[x0 * plane:x1 * plane])), (config, world, r)
    # a band that does NOT start on the binning granule: identical with the exact-fp32 kernel (per-voxel order is the
    # only thing that matters there), within tolerance with the matrix-core kernel (other tiles, other brick centres)
    out = agg.forward_slab(5, 21, *t)
    out = out if isinstance(out, tuple) else (out,)
    for a, b in zip(out, full):
        ref = b[5 * plane:21 * plane]
        if kw.get("matrix_cores") is False or si.variant == "prob":
            assert torch.equal(bits(a), bits(ref))
        else:
            assert float(((a - ref).abs() / ref.abs().clamp(min=1.0)).max()) <= 1e-4


def test_slab_gradients_add_up(gpu):
    import torch
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=62, P=800, H=32, W=20, D=16)
    agg = _module("nuscenes_gs25600_solid", si, gpu)
    g = torch.from_numpy(np.random.default_rng(63).standard_normal((si.pts.shape[0], 18)).astype(np.float32)).to(gpu)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)[None]

    def leaves():
        return [tt(a).requires_grad_(True) for a in (si.means3D, si.opacities, si.semantics, si.cov3D)]
    m, o, s_, c = leaves()
    agg(tt(si.pts), m, o, s_, tt(si.scales), c).backward(g)
    want = [x.grad.clone() for x in (m, o, s_, c)]
    m, o, s_, c = leaves()
    plane = si.W * si.D
    for x0, x1 in ((0, 16), (16, 32)):
        agg.forward_slab(x0, x1, tt(si.pts), m, o, s_, tt(si.scales), c).backward(g[x0 * plane:x1 * plane])
    for got, ref in zip((m.grad, o.grad, s_.grad, c.grad), want):
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max()))
