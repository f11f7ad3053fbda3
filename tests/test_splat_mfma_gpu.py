"""The matrix-core render kernel (GF_MFMA_SPLAT: base variant, dense lattice, split-f16 MFMA with fp32 accumulation)
through the C ABI: same bound as the default kernel against the CPU oracle / the reference's own kernels (1e-4 scaled),
the device-side lattice check, and the fallbacks."""
import numpy as np
import pytest

import oracle
from gaussianformer_amd.synthetic import make_splat_inputs

from util import assert_logits_close, hip_splat_forward, prep

pytestmark = pytest.mark.gpu

SMALL = [
    # config, P, H, W, D
    ("nuscenes_gs25600_solid", 300, 24, 20, 16),
    ("nuscenes_gs25600_solid", 257, 23, 21, 16),   # H, W not multiples of the tile
    ("nuscenes_gs25600_solid", 200, 20, 20, 10),   # D not a multiple of 4: scalar row stores, upper brick partly outside
    ("nuscenes_gs25600_solid", 200, 20, 20, 40),   # several z groups per tile
    ("nuscenes_gs25600_solid", 90, 9, 7, 4),       # grid smaller than one double brick
    ("nuscenes_gs144000", 1000, 40, 44, 16),
    ("nuscenes_gs25600_solid", 0, 12, 12, 8),      # only the appended whole-grid Gaussian
]


def _oracle_logits(si, pi, mi, radii, cov6):
    return oracle.splat_forward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"]


@pytest.mark.parametrize("config,P,H,W,D", SMALL)
def test_mfma_forward_small(gpu, config, P, H, W, D):
    from gaussianformer_amd import _lib
    si = make_splat_inputs(config, seed=3, P=P, H=H, W=W, D=D)
    pi, mi, radii, cov6 = prep(si)
    ref = _oracle_logits(si, pi, mi, radii, cov6)
    for flags in (_lib.GF_MFMA_SPLAT, _lib.GF_MFMA_SPLAT | _lib.GF_PTS_ASSUME_DENSE):
        got, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=flags)
        assert_logits_close(got["logits"], ref, tol=1e-4)


def test_mfma_membership_is_exact(gpu):
    """Sigma^-1 = 0, opacity = semantics = 1: every covered voxel receives exactly 1 per covering Gaussian -- the box
    term of the exponent (0 inside, <= -32768 outside) and the split-f16 accumulation are exact on these values."""
    from gaussianformer_amd import _lib
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=6, P=2000, H=40, W=36, D=16)
    pi, mi, radii, cov6 = prep(si)
    si.opacities[:] = 1.0
    si.semantics[:] = 1.0
    cov6 = np.zeros_like(cov6)
    ref = _oracle_logits(si, pi, mi, radii, cov6)
    got, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    assert np.array_equal(got["logits"], ref)


def test_mfma_crowded_tile(gpu):
    """Thousands of Gaussians in one supertile: list refills, queue carried across them, many groups per wave."""
    from gaussianformer_amd import _lib
    si = make_splat_inputs("nuscenes_gs144000", seed=9, P=6000, H=20, W=20, D=16)
    pi, mi, radii, cov6 = prep(si)
    ref = _oracle_logits(si, pi, mi, radii, cov6)
    got, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    assert_logits_close(got["logits"], ref, tol=1e-4)


@pytest.mark.parametrize("config,kw", [
    ("nuscenes_gs25600_solid", {}),                                   # the headline shape: 401 bitmask words, ~165 candidates per supertile
    ("nuscenes_gs144000", dict(P=6000, H=20, W=20, D=16)),            # crowded: lists longer than the wave kernel's, refills
    ("nuscenes_gs25600_solid", dict(P=39000, H=44, W=36, D=24)),     # 610 words (nearly the longest row it takes), three z bricks
    ("nuscenes_gs25600_solid", dict(P=700, H=9, W=7, D=4)),          # one supertile column, one partial brick
])
def test_mfma_wave_and_tile_kernels_agree_bit_for_bit(gpu, config, kw, monkeypatch):
    """The two matrix-core kernels -- one wave per double brick (rows of <= 618 words) and one workgroup per tile
    (GF_MFMA_TILE=1 forces it) -- take a double brick's hits in the same groups of 32 in ascending index and run the same
    arithmetic on them: equal bits, whatever path (fast fill, chunked refill) built the lists."""
    from gaussianformer_amd import _lib
    si = make_splat_inputs(config, seed=4, **kw)
    pi, mi, radii, cov6 = prep(si)
    monkeypatch.delenv("GF_MFMA_TILE", raising=False)
    wave, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    monkeypatch.setenv("GF_MFMA_TILE", "1")
    tile, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    monkeypatch.delenv("GF_MFMA_TILE", raising=False)
    assert np.isfinite(wave["logits"]).all()
    assert np.array_equal(wave["logits"], tile["logits"])
    ref = _oracle_logits(si, pi, mi, radii, cov6)
    assert_logits_close(wave["logits"], ref, tol=1e-4)


def test_mfma_inexact_lattice_falls_back(gpu):
    """pts one ulp off the lattice in a few voxels (still inside their voxels): the device-side check must notice -- the
    kernel evaluates its polynomial on the lattice, not on pts -- and the arbitrary-points body must produce the result
    for the positions actually given.  Anisotropic but exact steps are a lattice and stay on the matrix cores."""
    import torch
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_forward
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=11, P=300, H=24, W=20, D=16)
    pi, mi, radii, cov6 = prep(si)
    rng = np.random.default_rng(1)
    pick = rng.choice(si.pts.shape[0], 50, replace=False)
    si.pts[pick, 0] = np.nextafter(si.pts[pick, 0], np.float32(1e9))
    ref = _oracle_logits(si, pi, mi, radii, cov6)          # the oracle reads pts: the perturbed positions
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    logits, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_MFMA_SPLAT)
    # state block: the points ARE in their voxels (word 0 = 0, the backward keeps its dense path), the call was rendered
    # by the arbitrary-points body (word 1) because the lattice verdict (bit 1 of word 2) failed
    assert state.view(torch.int32)[:3].tolist() == [0, _lib.GF_PATH_ARBITRARY, 2]
    assert_logits_close(logits.cpu().numpy(), ref, tol=1e-4)
    # the thinnest Gaussians make one ulp visible: the result must be the one for the given positions, bit for bit the
    # arbitrary-points kernel's
    general, *_ = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D, flags=_lib.GF_PTS_GENERAL | _lib.GF_COMP_EXP)
    assert torch.equal(general, logits)


def test_mfma_coefficient_range_verdict(gpu):
    """A Gaussian whose exponent coefficients would leave the f16 range of the matrix-core kernel's operands (thin AND
    far-reaching: scale 0.004 m with a radius of 12 voxels) trips the prep launch's range verdict: the call is rendered
    by the arbitrary-points body (finite, within tolerance of the oracle) and the state block says so."""
    import torch
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import splat_forward
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=13, P=200, H=24, W=24, D=16)
    pi, mi, radii, cov6 = prep(si)
    cov6[5] = (np.float32(1.0 / 0.004 ** 2), np.float32(1.0), np.float32(1.0), 0, 0, 0)   # (xx, yy, zz, xy, yz, xz)
    radii[5] = 12
    ref = _oracle_logits(si, pi, mi, radii, cov6)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(gpu) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    logits, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
    assert state.view(torch.int32)[:3].tolist() == [0, _lib.GF_PATH_ARBITRARY, 4]
    assert bool(torch.isfinite(logits).all())
    assert_logits_close(logits.cpu().numpy(), ref, tol=1e-4)
    # with an ordinary covariance in its place the same call stays on the matrix cores
    cov6[5] = cov6[6]
    t[7] = torch.from_numpy(cov6).to(gpu)
    _, _, _, _, state = splat_forward(_lib.GF_SPLAT_BASE, *t, si.H, si.W, si.D)
    assert state.view(torch.int32)[:3].tolist() == [0, _lib.GF_PATH_MATRIX_CORE_WAVE, 0]   # (a small P: the wave kernel)


@pytest.mark.parametrize("config", ["nuscenes_gs25600_solid", "nuscenes_gs144000"])
def test_mfma_forward_full_size(gpu, config):
    """BASELINE shapes: against the reference's own kernels when oracle/_ref is built, else the CPU oracle; linear in the
    semantics; deterministic."""
    from gaussianformer_amd import _lib
    from oracle import ref as oref
    si = make_splat_inputs(config, seed=0)
    pi, mi, radii, cov6 = prep(si)
    if oref.available():
        want = oref.splat_forward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"]
    else:
        want = _oracle_logits(si, pi, mi, radii, cov6)
    got, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    assert_logits_close(got["logits"], want, tol=1e-4)
    again, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    assert np.array_equal(again["logits"], got["logits"])
    si.semantics *= np.float32(2.0)
    twice, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    # (not bit-exact like the fp32 kernel: the low f16 halves of small operands are subnormal, where a factor 2 rounds anew)
    assert np.allclose(twice["logits"], np.float32(2.0) * got["logits"], rtol=2e-6, atol=1e-6)


def test_mfma_flag_is_ignored_where_it_does_not_apply(gpu):
    """prob variant and arbitrary points: the flag changes nothing."""
    from gaussianformer_amd import _lib
    si = make_splat_inputs("prob_gs6400", seed=3, P=120, H=24, W=20, D=16)
    pi, mi, radii, cov6 = prep(si)
    a, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6)
    b, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=4, P=150, H=20, W=24, D=16, dense_pts=False, N=5000)
    pi, mi, radii, cov6 = prep(si)
    a, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6)
    b, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
    assert np.array_equal(a["logits"], b["logits"])


def test_mfma_random_shapes_sweep(gpu):
    """Seeded sweep over grid shapes, Gaussian counts and scale ranges (thin and wide Gaussians, with and without the
    whole-grid one): the matrix-core kernel within the bound of the default one everywhere."""
    from gaussianformer_amd import _lib
    rng = np.random.default_rng(2024)
    worst = 0.0
    for it in range(10):
        H, W, D = int(rng.integers(5, 45)), int(rng.integers(5, 45)), int(rng.choice([4, 8, 12, 16, 24]))
        P = int(rng.integers(1, 1500))
        config = "nuscenes_gs144000" if it % 3 == 0 else "nuscenes_gs25600_solid"
        si = make_splat_inputs(config, seed=100 + it, P=P, H=H, W=W, D=D)
        pi, mi, radii, cov6 = prep(si)
        ref = _oracle_logits(si, pi, mi, radii, cov6)
        got, *_ = hip_splat_forward(gpu, si, pi, mi, radii, cov6, flags=_lib.GF_MFMA_SPLAT)
        assert_logits_close(got["logits"], ref, what=f"sweep {it} ({config}, P={P}, {H}x{W}x{D})", tol=1e-4)
        worst = max(worst, float((np.abs(got["logits"].astype(np.float64) - ref) / np.maximum(1.0, np.abs(ref))).max()))
    assert worst <= 1e-4


def test_mfma_module_keyword_and_autograd(gpu):
    """``LocalAggregator(..., matrix_cores=True)``: forward on the matrix cores, the same backward; gradients against the
    exact module on the same inputs."""
    import torch
    from gaussianformer_amd.local_aggregate import LocalAggregator
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=12, P=400, H=24, W=24, D=16)
    mods = [LocalAggregator(si.scale_multiplier, si.H, si.W, si.D, list(si.pc_min), si.grid_size, matrix_cores=mc).to(gpu)
            for mc in (False, True)]
    outs, grads = [], []
    for m in mods:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)[None]
        means, opa, sem, cov = (t(si.means3D).requires_grad_(True), t(si.opacities).requires_grad_(True),
                                t(si.semantics).requires_grad_(True), t(si.cov3D).requires_grad_(True))
        out = m(t(si.pts), means, opa, sem, t(si.scales), cov)
        g = torch.Generator(device="cpu").manual_seed(3)
        out.backward(torch.randn(out.shape, generator=g).to(gpu))
        outs.append(out.detach())
        grads.append([x.grad.clone() for x in (means, opa, sem, cov)])
    assert float(((outs[0] - outs[1]).abs() / outs[0].abs().clamp(min=1.0)).max()) <= 1e-4
    assert not torch.equal(outs[0], outs[1])                       # it really is the other kernel
    for a, b in zip(*grads):
        # the backward does not depend on the forward kernel (boxes split over several waves combine with float atomics:
        # equal up to summation order)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()))
