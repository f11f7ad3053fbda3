"""Submanifold sparse convolution against its definition (oracle/subm_ref.py): a dense
K^3 convolution of the scattered-and-summed features, read back at the points (fp64 torch, autograd
for the gradients).  spconv itself is not available (no ROCm build, not in the reference tree)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


from oracle.subm_ref import subm_conv3d_dense as _dense_reference  # noqa: E402  (the checker)


def _points(rng, N, batch, shape, dup=0.1, outside=2):
    X, Y, Z = shape
    idx = np.stack([rng.integers(0, batch, N), rng.integers(0, X, N), rng.integers(0, Y, N), rng.integers(0, Z, N)], 1)
    ndup = int(N * dup)
    if ndup and N > 1:
        src = rng.integers(0, N, ndup)
        dst = rng.integers(0, N, ndup)
        idx[dst] = idx[src]                      # points sharing a cell
    if outside and N > outside:
        idx[:outside, 1] = X + 3                 # inactive points (outside the grid)
    return torch.from_numpy(idx.astype(np.int32))


@pytest.mark.parametrize("N,batch,shape,K,cin,cout", [
    (600, 1, (12, 10, 6), 5, 128, 128),      # the reference's layer: 5^3, 128 -> 128
    (900, 2, (9, 14, 4), 3, 32, 64),
    (300, 1, (6, 6, 6), 5, 64, 32),
    (1, 1, (4, 4, 4), 3, 32, 32),
    (2000, 1, (8, 8, 3), 5, 128, 128),       # crowded: ~10 points per cell, many duplicates
])
def test_subm_conv_forward_backward(N, batch, shape, K, cin, cout):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import Rulebook, subm_conv3d
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N + K)
    idx = _points(rng, N, batch, shape)
    g = torch.Generator().manual_seed(N)
    feat = torch.randn(N, cin, generator=g)
    weight = torch.randn(K ** 3, cin, cout, generator=g) * 0.1
    gout = torch.randn(N, cout, generator=g)
    f64, w64 = feat.double().requires_grad_(True), weight.double().requires_grad_(True)
    ref = _dense_reference(f64, idx.long(), w64, batch, shape, K)
    (ref * gout.double()).sum().backward()

    fd, wd = feat.to(dev).requires_grad_(True), weight.to(dev).requires_grad_(True)
    rb = Rulebook(idx.to(dev), batch, shape, K)
    out = subm_conv3d(fd, idx.to(dev), wd, batch, shape, K, rulebook=rb)
    scale = float(ref.detach().abs().max()) + 1e-6
    assert (out.detach().cpu().double() - ref.detach()).abs().max() <= 2e-5 * scale
    out.backward(gout.to(dev))
    assert (fd.grad.cpu().double() - f64.grad).abs().max() <= 2e-5 * (float(f64.grad.abs().max()) + 1e-6)
    assert (wd.grad.cpu().double() - w64.grad).abs().max() <= 5e-5 * (float(w64.grad.abs().max()) + 1e-6)
    # the pair count is the number of (out, in, offset) incidences of the definition
    cells = {}
    for n, c in enumerate(idx.tolist()):
        if 0 <= c[1] < shape[0]:
            cells.setdefault(tuple(c), []).append(n)
    r = K // 2
    want = 0
    for c, members in cells.items():
        for dx in range(-r, r + 1):
            for dy in range(-r, r + 1):
                for dz in range(-r, r + 1):
                    want += len(members) * len(cells.get((c[0], c[1] + dx, c[2] + dy, c[3] + dz), ()))
    assert rb.total == want
    # deterministic: a second rulebook + apply gives the same bits -- shared cells included (their lists are walked in
    # ascending point index since round 4; they used to be walked in insertion order, which differs from run to run)
    out2 = Rulebook(idx.to(dev), batch, shape, K).apply(fd.detach(), wd.detach())
    assert torch.equal(out2, out.detach())


def test_subm_conv_is_reproducible_with_crowded_cells():
    """VERDICT r3 (`graph_labels_equal_eager: false` at 144 000 points): with several points per cell the rulebook's pair
    order inside an (output point, offset) run used to follow the atomicExch insertion order of the cell lists, so the
    fp32 sum of a row's partial rows rounded differently from run to run.  Now every rulebook -- with the host read or
    sized in advance (the graph-capturable mode) -- lists a cell's points in ascending index: ten builds, equal bits, and
    pair_in ascending inside every run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import Rulebook
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7)
    N, shape, K = 2000, (10, 10, 4), 5                        # 5 points per cell on average, ~0.8 M pairs
    idx = _points(rng, N, 1, shape, dup=0.3).to(dev)
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(N, 32, generator=g).to(dev)
    weight = (torch.randn(K ** 3, 32, 32, generator=g) * 0.1).to(dev)
    first = None
    for it in range(10):
        rb = Rulebook(idx, 1, shape, K) if it % 2 == 0 else Rulebook(idx, 1, shape, K, pair_capacity=1 << 22)
        out = rb.apply(feat, weight)
        if it % 2:
            rb.check()
        assert torch.isfinite(out).all()
        if first is None:
            first = out.clone()
            total = rb.total
            po, pin = rb.pair_out[:total].cpu().numpy(), rb.pair_in[:total].cpu().numpy()
            ih = idx.cpu().numpy().astype(np.int64)
            off = ih[pin] - ih[po]                              # the offset (0, dx, dy, dz) of every pair
            same_run = (po[1:] == po[:-1]) & (off[1:] == off[:-1]).all(axis=1)   # consecutive slots of one (output point, offset) run
            assert same_run.sum() > 1000                        # the point set really has crowded cells
            assert (pin[1:][same_run] > pin[:-1][same_run]).all()   # ... list their inputs in ascending index
        else:
            assert torch.equal(out, first), f"build {it} differs"


def test_sparse_conv3d_module():
    """SparseConv3D block at the nuScenes geometry (pc_range +-40 m, 0.5 m cells -> 160x160x16 grid,
    spconv3d_module.py:47-83): shapes, gradient flow, agreement with the functional op."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import SparseConv3D, subm_conv3d
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = SparseConv3D(128, 128, pc_range=[-40.0, -40.0, -1.0, 40.0, 40.0, 7.0], grid_size=[0.5, 0.5, 0.5], use_out_proj=True).to(dev)
    anchor = torch.randn(2, 3000, 11, device=dev)
    feat = torch.randn(2, 3000, 128, device=dev, requires_grad=True)
    out = m(feat, anchor)
    assert out.shape == (2, 3000, 128) and torch.isfinite(out).all()
    out.square().mean().backward()
    assert torch.isfinite(feat.grad).all() and float(feat.grad.abs().max()) > 0
    assert torch.isfinite(m.layer.weight.grad).all() and float(m.layer.weight.grad.abs().max()) > 0
    # same indices -> same result through the functional op
    xyz = anchor[..., :3].clamp(-9.21, 9.21).sigmoid() * torch.tensor([80.0, 80.0, 8.0], device=dev) + torch.tensor([-40.0, -40.0, -1.0], device=dev)
    idx = ((xyz.flatten(0, 1) - torch.tensor([-40.0, -40.0, -1.0], device=dev)) / 0.5).to(torch.int32)
    indices = torch.cat([torch.arange(2, device=dev, dtype=torch.int32).repeat_interleave(3000)[:, None], idx], -1)
    y = subm_conv3d(feat.detach().flatten(0, 1), indices, m.layer.weight.detach(), 2, (160, 160, 16), 5)
    assert torch.allclose(m.output_proj(y.unflatten(0, (2, 3000))), out.detach(), rtol=1e-5, atol=1e-5)


def test_sparse_conv3d_multi_layer():
    """``use_multi_layer`` (spconv3d_module.py:26-37): three conv(+bias) + LayerNorm + ReLU stages over
    one shared rulebook, against the same chain built from the dense fp64 definition."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import SparseConv3D, SubMConv3d
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    pc_range, cell = [-4.0, -4.0, -1.0, 4.0, 4.0, 1.0], [0.5, 0.5, 0.5]
    m = SparseConv3D(32, 64, pc_range=pc_range, grid_size=cell, use_multi_layer=True, kernel_size=3).to(dev)
    anchor = torch.randn(2, 500, 11, device=dev)
    feat = torch.randn(2, 500, 32, device=dev, requires_grad=True)
    out = m(feat, anchor)
    out.square().mean().backward()
    assert out.shape == (2, 500, 64)

    idx = m.voxel_indices(anchor).cpu()
    assert m._spatial == [16, 16, 4]
    f64 = feat.detach().cpu().double().flatten(0, 1).requires_grad_(True)
    x, params = f64, []
    for layer in m.layer:
        if isinstance(layer, SubMConv3d):
            w = layer.weight.detach().cpu().double().requires_grad_(True)
            params.append((layer.weight, w))
            x = _dense_reference(x, idx.long(), w, 2, (16, 16, 4), 3) + layer.bias.detach().cpu().double()
        elif isinstance(layer, torch.nn.LayerNorm):
            x = torch.nn.functional.layer_norm(x, (64,), layer.weight.detach().cpu().double(), layer.bias.detach().cpu().double(), layer.eps)
        else:
            x = torch.relu(x)
    x.square().mean().backward()
    assert torch.allclose(out.detach().cpu().double().flatten(0, 1), x.detach(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(feat.grad.cpu().double().flatten(0, 1), f64.grad, rtol=1e-3, atol=1e-4 * float(f64.grad.abs().max()))
    for p, w in params:
        assert torch.allclose(p.grad.cpu().double(), w.grad, rtol=1e-3, atol=1e-4 * float(w.grad.abs().max()))


def test_subm_conv_degenerate_inputs():
    """No points, every point outside the grid, and an isolated point (only the centre offset pairs with itself)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import subm_conv3d, Rulebook
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    w = torch.randn(27, 32, 64, device=dev)
    # empty
    out = subm_conv3d(torch.zeros(0, 32, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), w, 1, (4, 4, 4), 3)
    assert out.shape == (0, 64)
    # all outside -> no pairs, zero output, zero gradients
    idx = torch.tensor([[0, 9, 0, 0], [0, -1, 2, 2], [3, 1, 1, 1]], dtype=torch.int32, device=dev)
    f = torch.randn(3, 32, device=dev, requires_grad=True)
    wp = w.clone().requires_grad_(True)
    rb = Rulebook(idx, 1, (4, 4, 4), 3)
    assert rb.total == 0
    out = subm_conv3d(f, idx, wp, 1, (4, 4, 4), 3, rulebook=rb)
    assert float(out.abs().max()) == 0.0
    out.sum().backward()
    assert float(f.grad.abs().max()) == 0.0 and float(wp.grad.abs().max()) == 0.0
    # one isolated point: out = feat . W[centre]
    idx = torch.tensor([[0, 2, 2, 2]], dtype=torch.int32, device=dev)
    f = torch.randn(1, 32, device=dev)
    out = subm_conv3d(f, idx, w, 1, (5, 5, 5), 3)
    assert torch.allclose(out, f @ w[13], rtol=1e-5, atol=1e-5)


def test_subm_conv_random_sweep():
    """Random point counts, grids, kernel sizes and channel pairs (GF_SWEEP_SEED / GF_SWEEP_TRIALS select others),
    forward and both gradients against the dense definition."""
    import os
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import subm_conv3d
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(int(os.environ.get("GF_SWEEP_SEED", "11")))
    for trial in range(int(os.environ.get("GF_SWEEP_TRIALS", "10"))):
        K = int(rng.choice([1, 3, 5, 7]))
        batch = int(rng.integers(1, 4))
        shape = tuple(int(v) for v in rng.integers(1, 12, 3))
        N = int(rng.integers(1, 1500))
        cin, cout = (int(v) for v in rng.choice([32, 64, 128], 2))
        idx = _points(rng, N, batch, shape, dup=float(rng.random()) * 0.5, outside=int(rng.integers(0, 4)))
        g = torch.Generator().manual_seed(1000 + trial)
        feat = torch.randn(N, cin, generator=g)
        weight = torch.randn(K ** 3, cin, cout, generator=g) * 0.1
        gout = torch.randn(N, cout, generator=g)
        f64, w64 = feat.double().requires_grad_(True), weight.double().requires_grad_(True)
        ref = _dense_reference(f64, idx.long(), w64, batch, shape, K)
        (ref * gout.double()).sum().backward()
        fd, wd = feat.to(dev).requires_grad_(True), weight.to(dev).requires_grad_(True)
        out = subm_conv3d(fd, idx.to(dev), wd, batch, shape, K)
        out.backward(gout.to(dev))
        what = f"trial {trial}: N={N} batch={batch} shape={shape} K={K} {cin}->{cout}"
        scale = float(ref.detach().abs().max()) + 1e-6
        assert (out.detach().cpu().double() - ref.detach()).abs().max() <= 3e-5 * scale, what
        assert (fd.grad.cpu().double() - f64.grad).abs().max() <= 3e-5 * (float(f64.grad.abs().max()) + 1e-6), what
        assert (wd.grad.cpu().double() - w64.grad).abs().max() <= 1e-4 * (float(w64.grad.abs().max()) + 1e-6), what


def test_subm_conv_crowded_cells():
    """Hundreds of points per cell (a sweep once caught 8-bit per-offset counts saturating at 255), and the loud
    refusal of a cell beyond the 16-bit limit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import Rulebook, subm_conv3d
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    N, batch, shape, K = 1400, 1, (1, 2, 2), 1
    idx = _points(rng, N, batch, shape, dup=0.0, outside=0)
    g = torch.Generator().manual_seed(9)
    feat, weight = torch.randn(N, 64, generator=g), torch.randn(1, 64, 128, generator=g) * 0.1
    ref = _dense_reference(feat.double(), idx.long(), weight.double(), batch, shape, K)
    out = subm_conv3d(feat.to(dev), idx.to(dev), weight.to(dev), batch, shape, K)
    assert (out.cpu().double() - ref).abs().max() <= 3e-5 * float(ref.abs().max())
    crowd = torch.zeros(66000, 4, dtype=torch.int32, device=dev)          # 66 000 points in one cell
    with pytest.raises(RuntimeError, match="65535"):
        Rulebook(crowd, 1, (2, 2, 2), 1)


def test_subm_conv_without_host_read():
    """gf_subm_rulebook_build: pair arrays sized in advance, no count read back.  Same output and gradients as the exact
    rulebook when the pairs fit; an EMPTY rulebook (NaN output: loud, not a silent zero) and a refusal that check() raises
    -- and that SparseConv3D reports by itself on a later call, without blocking -- when they do not."""
    from gaussianformer_amd.sparse_conv import Rulebook, SparseConv3D, subm_conv3d
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    N, batch, shape, K = 1500, 1, (14, 12, 6), 5
    idx = _points(rng, N, batch, shape).to(dev)
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(N, 128, generator=g).to(dev)
    weight = (torch.randn(K ** 3, 128, 128, generator=g) * 0.1).to(dev)
    gout = torch.randn(N, 128, generator=g).to(dev)
    exact = Rulebook(idx, batch, shape, K)
    outs = []
    for rb in (exact, Rulebook(idx, batch, shape, K, pair_capacity=exact.total + 777)):
        f, w = feat.clone().requires_grad_(True), weight.clone().requires_grad_(True)
        out = subm_conv3d(f, idx, w, batch, shape, K, rulebook=rb)
        out.backward(gout)
        outs.append((out.detach(), f.grad, w.grad))
    # two builds may chain the points of a shared cell in a different order (atomicExch): equal up to summation order
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max()))
    small = Rulebook(idx, batch, shape, K, pair_capacity=exact.total - 1)
    out = subm_conv3d(feat, idx, weight, batch, shape, K, rulebook=small)
    torch.cuda.synchronize()
    assert bool(torch.isnan(out).all())
    with pytest.raises(RuntimeError, match="pair_capacity"):
        small.check()
    # module keyword
    m = SparseConv3D(128, 128, [0.0, 0.0, 0.0, 30.0, 30.0, 8.0], [0.5, 0.5, 0.5], pairs_per_point=125).to(dev)
    ref = SparseConv3D(128, 128, [0.0, 0.0, 0.0, 30.0, 30.0, 8.0], [0.5, 0.5, 0.5]).to(dev)
    ref.load_state_dict(m.state_dict())
    anchor = torch.randn(1, 800, 11, device=dev)
    x = torch.randn(1, 800, 128, device=dev)
    ya, yb = m(x, anchor), ref(x, anchor)
    assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-5 * float(yb.abs().max()))
    assert m.last_rulebook.check() == ref.last_rulebook.total
    # a module whose capacity is too small: NaN out, and the refusal surfaces from a LATER forward (deferred, non-blocking
    # poll of the status copied to pinned memory) -- at the latest after a synchronisation
    tight = SparseConv3D(128, 128, [0.0, 0.0, 0.0, 30.0, 30.0, 8.0], [0.5, 0.5, 0.5], pairs_per_point=1).to(dev)
    y = tight(x, anchor)
    torch.cuda.synchronize()
    assert bool(torch.isnan(y).all())
    with pytest.raises(RuntimeError, match="pair_capacity"):
        tight(x, anchor)


@pytest.mark.parametrize("N,cin,cout", [(9000, 128, 128), (9000, 64, 32), (24000, 128, 128), (24000, 32, 64)])
def test_subm_conv_capacity_rulebook_takes_the_organisation_its_pair_count_asks_for(N, cin, cout):
    """A rulebook sized by a capacity hands the library an upper bound of its pair count.  Where the bound says "long segments" the
    gather-GEMM is launched in both organisations (runs of eight tiles; one tile per workgroup) and the rulebook's own count, read
    on the device, lets exactly one of them write: the same bits as with the exact rulebook -- 9 000 points: bound long, count
    short; 24 000 points: both long."""
    from gaussianformer_amd.sparse_conv import Rulebook
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(13)
    batch, shape, K = 1, (40, 40, 16), 5
    idx = _points(rng, N, batch, shape).to(dev)
    g = torch.Generator().manual_seed(6)
    feat = torch.randn(N, cin, generator=g).to(dev)
    weight = (torch.randn(K ** 3, cin, cout, generator=g) * 0.05).to(dev)
    exact = Rulebook(idx, batch, shape, K)
    roomy = Rulebook(idx, batch, shape, K, pair_capacity=128 * N)
    long_bound, long_count = (128 * N) // 128 >= K ** 3 * 32, exact.total // 128 >= K ** 3 * 32
    assert long_bound and long_count == (N == 24000), (exact.total, N)
    a, b = exact.apply(feat, weight), roomy.apply(feat, weight)
    assert roomy.check() == exact.total
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
    assert float(a.abs().max()) > 0


@pytest.mark.parametrize("cin,cout", [(128, 128), (32, 64)])
def test_subm_conv_bf16_split_against_f32_mfma(cin, cout):
    """The default gather-GEMM (fp32 operands split into three bf16 terms, six bf16 MFMAs per product) against the
    f32-MFMA kernel (library option "subm.f32_mfma": bitwise an fmaf chain) on the same rulebook, and both against the fp64
    definition: the split must stay in fp32's error class, also for operands spanning many binades (a plain bf16 GEMM
    would be at 4e-3)."""
    from gaussianformer_amd.sparse_conv import Rulebook
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    N, batch, shape, K = 1200, 1, (12, 10, 6), 5
    idx = _points(rng, N, batch, shape)
    g = torch.Generator().manual_seed(9)
    feat = torch.randn(N, cin, generator=g) * torch.exp2(torch.randint(-12, 12, (N, cin), generator=g).float())
    weight = torch.randn(K ** 3, cin, cout, generator=g) * torch.exp2(torch.randint(-8, 8, (K ** 3, cin, cout), generator=g).float())
    ref = _dense_reference(feat.double(), idx.long(), weight.double(), batch, shape, K)
    # the error scale of an fp32 dot product: eps * sum |a||b|, per output element
    mag = _dense_reference(feat.double().abs(), idx.long(), weight.double().abs(), batch, shape, K)
    rb = Rulebook(idx.to(dev), batch, shape, K)
    from gaussianformer_amd import _lib
    split = rb.apply(feat.to(dev), weight.to(dev)).cpu().double()     # round 6: two f16 terms under row / column scales, three products
    with _lib.option("subm.bf16x3", 1):
        split3 = rb.apply(feat.to(dev), weight.to(dev)).cpu().double()   # three bf16 terms, six products (rounds 2 - 5)
    with _lib.option("subm.f32_mfma", 1):
        exact = rb.apply(feat.to(dev), weight.to(dev)).cpu().double()
    assert not torch.equal(split, exact) and not torch.equal(split3, exact) and not torch.equal(split, split3)   # three different kernels did run
    bound = 2.0 ** -20 * mag + 1e-30        # 16 units of fp32 roundoff of sum |a||b| (a plain bf16 product: 2^-8, a plain f16 one: 2^-11)
    assert bool(((exact - ref).abs() <= bound).all())
    assert bool(((split3 - ref).abs() <= bound).all())
    assert bool(((split - ref).abs() <= bound).all())
    assert bool(((split - exact).abs() <= bound).all())
    print("max error in units of 2^-24 sum|a||b|: f16 x 2", float(((split - ref).abs() / (2.0 ** -24 * mag + 1e-300)).max()),
          "bf16 x 3", float(((split3 - ref).abs() / (2.0 ** -24 * mag + 1e-300)).max()),
          "f32 mfma", float(((exact - ref).abs() / (2.0 ** -24 * mag + 1e-300)).max()))


def test_subm_conv_f16_split_rows_of_extreme_magnitude():
    """The two-term f16 split scales every feature row and every weight column by an exact power of two: rows of magnitude 1e-30
    and 1e+30 next to ordinary ones, an all-zero row, a zero weight column and denormal weights give what the f32-MFMA kernel
    gives, to the same bound."""
    from gaussianformer_amd import _lib
    from gaussianformer_amd.sparse_conv import Rulebook
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    N, batch, shape, K, cin, cout = 900, 1, (10, 9, 6), 3, 64, 64
    idx = _points(rng, N, batch, shape)
    g = torch.Generator().manual_seed(10)
    feat = torch.randn(N, cin, generator=g)
    feat[::7] *= 1e-30
    feat[3::7] *= 1e+15
    feat[5] = 0.0
    weight = torch.randn(K ** 3, cin, cout, generator=g) * 0.05
    weight[:, :, 3] = 0.0
    weight[:, :, 4] *= 1e-36          # denormal-range weights
    weight[:, :, 5] *= 1e+12
    ref = _dense_reference(feat.double(), idx.long(), weight.double(), batch, shape, K)
    mag = _dense_reference(feat.double().abs(), idx.long(), weight.double().abs(), batch, shape, K)
    rb = Rulebook(idx.to(dev), batch, shape, K)
    split = rb.apply(feat.to(dev), weight.to(dev)).cpu().double()
    assert bool(torch.isfinite(split).all())
    # (column 4: products of 1e-36 weights underflow fp32 in every kernel; judged against fp32's smallest normal)
    bound = 2.0 ** -20 * mag + 2.0 ** -120
    assert bool(((split - ref).abs() <= bound).all()), float(((split - ref).abs() / bound).max())
    assert bool((split[:, 3] == 0).all())


@pytest.mark.parametrize("N,shape,cin,cout", [(24000, (40, 40, 16), 128, 128), (9000, (20, 24, 8), 64, 32), (6000, (12, 12, 6), 32, 128)])
def test_subm_conv_runs_of_tiles_equal_single_tiles(N, shape, cin, cout):
    """Round 6: on long segments (>= 32 tiles of 128 pairs per offset) the gather-GEMM walks RUNS of eight tiles per workgroup
    (W slice converted once, two-stage gather pipeline) -- the same operands in the same order as one tile per workgroup
    (library option "subm.tile_gemm"): equal bits, ragged segment ends and crowded cells included; and both agree with the
    fp64 definition."""
    from gaussianformer_amd import _lib
    from gaussianformer_amd.sparse_conv import Rulebook
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    batch, K = 1, 5
    idx = _points(rng, N, batch, shape)
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(N, cin, generator=g)
    weight = torch.randn(K ** 3, cin, cout, generator=g) * 0.05
    rb = Rulebook(idx.to(dev), batch, shape, K)
    assert rb.total // 128 >= K ** 3 * 32, rb.total                  # the shape does take the run kernel
    runs = rb.apply(feat.to(dev), weight.to(dev))
    with _lib.option("subm.tile_gemm", 1):
        tiles = rb.apply(feat.to(dev), weight.to(dev))
    assert torch.equal(runs, tiles)
    with _lib.option("subm.bf16x3", 1):                              # (the three-term bf16 kernels: the same two organisations)
        runs3 = rb.apply(feat.to(dev), weight.to(dev))
        with _lib.option("subm.tile_gemm", 1):
            tiles3 = rb.apply(feat.to(dev), weight.to(dev))
    assert torch.equal(runs3, tiles3) and not torch.equal(runs3, runs)
    if N <= 9000:
        ref = _dense_reference(feat.double(), idx.long(), weight.double(), batch, shape, K)
        mag = _dense_reference(feat.double().abs(), idx.long(), weight.double().abs(), batch, shape, K)
        assert bool(((runs.cpu().double() - ref).abs() <= 2.0 ** -20 * mag + 1e-30).all())


def _representative_reference(feat, idx, weight, batch, shape, K):
    """``duplicates="last"`` stated directly (fp64, differentiable): a hash of cell -> the LARGEST point index in it, then
    out[i] = sum_k feat[table[cell(i) + offset_k]] . W[k] -- what spconv's SubMConv3d computes when that duplicate is the
    one its hash insert kept."""
    X, Y, Z = shape
    table = {}
    for i, (b, x, y, z) in enumerate(idx.tolist()):
        if 0 <= b < batch and 0 <= x < X and 0 <= y < Y and 0 <= z < Z:
            table[(b, x, y, z)] = i                      # ascending i: the last one stays
    r = K // 2
    rows = []
    for i, (b, x, y, z) in enumerate(idx.tolist()):
        acc = feat.new_zeros(weight.shape[2])
        if (b, x, y, z) in table:
            for kx in range(K):
                for ky in range(K):
                    for kz in range(K):
                        j = table.get((b, x + kx - r, y + ky - r, z + kz - r))
                        if j is not None:
                            acc = acc + feat[j] @ weight[(kx * K + ky) * K + kz]
        rows.append(acc)
    return torch.stack(rows)


def test_subm_conv_last_duplicate_mode():
    """``duplicates="last"``: one representative per cell as the neighbour source (the spconv-compatible reading of points
    that share a cell), forward and both gradients against the direct statement; without duplicates it equals "sum"."""
    from gaussianformer_amd.sparse_conv import SparseConv3D, subm_conv3d
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    N, batch, shape, K, cin, cout = 260, 2, (5, 6, 3), 3, 32, 32
    idx = _points(rng, N, batch, shape, dup=0.5, outside=3)
    g = torch.Generator().manual_seed(4)
    feat, weight = torch.randn(N, cin, generator=g), torch.randn(K ** 3, cin, cout, generator=g) * 0.2
    f64, w64 = feat.double().requires_grad_(True), weight.double().requires_grad_(True)
    ref = _representative_reference(f64, idx, w64, batch, shape, K)
    go = torch.randn(N, cout, generator=g)
    ref.backward(go.double())
    fd, wd = feat.to(dev).requires_grad_(True), weight.to(dev).requires_grad_(True)
    out = subm_conv3d(fd, idx.to(dev), wd, batch, shape, K, duplicates="last")
    out.backward(go.to(dev))
    scale = float(ref.abs().max())
    assert (out.detach().cpu().double() - ref.detach()).abs().max() <= 3e-5 * scale
    assert (fd.grad.cpu().double() - f64.grad).abs().max() <= 3e-5 * (float(f64.grad.abs().max()) + 1e-6)
    assert (wd.grad.cpu().double() - w64.grad).abs().max() <= 1e-4 * (float(w64.grad.abs().max()) + 1e-6)
    # points that are not the representative of their cell are no source: no gradient reaches them
    keys = [tuple(r) for r in idx.tolist()]
    last = {k: i for i, k in enumerate(keys)}
    shadowed = [i for i, k in enumerate(keys) if last[k] != i and 0 <= k[1] < shape[0]]
    assert shadowed and float(fd.grad[shadowed].abs().max()) == 0.0
    # and it differs from the default there
    out_sum = subm_conv3d(feat.to(dev), idx.to(dev), weight.to(dev), batch, shape, K)
    assert float((out_sum - out.detach()).abs().max()) > 1e-3
    # no duplicates: the two modes are the same operator
    uniq = torch.from_numpy(np.unique(idx.numpy(), axis=0))
    fu = torch.randn(uniq.shape[0], cin, generator=g).to(dev)
    a = subm_conv3d(fu, uniq.to(dev), weight.to(dev), batch, shape, K, duplicates="last")
    b = subm_conv3d(fu, uniq.to(dev), weight.to(dev), batch, shape, K)
    assert torch.equal(a, b)
    # module keyword
    m = SparseConv3D(32, 32, [0, 0, 0, 5, 6, 3], [1.0, 1.0, 1.0], kernel_size=3, duplicates="last").to(dev)
    assert m.layer.duplicates == "last"
    with pytest.raises(ValueError):
        SparseConv3D(32, 32, [0, 0, 0, 5, 6, 3], [1.0, 1.0, 1.0], duplicates="first")


@pytest.mark.parametrize("pairs_per_point", [None, 64])
def test_sparse_conv3d_output_range(pairs_per_point):
    """Round 5 (the anchor-sharded frame): ``out_range=(lo, hi)`` evaluates the block for a slice of the anchors against the WHOLE
    set as neighbourhood.  The rows are those of the whole block bit for bit (same pairs in the same order per output row), for
    every slice of a partition; the rulebook holds that slice's pairs only; autograd through such a rulebook is refused."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gaussianformer_amd.sparse_conv import Rulebook, SparseConv3D, subm_conv3d
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = SparseConv3D(128, 128, pc_range=[-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], grid_size=[0.5, 0.5, 0.5], use_out_proj=True,
                     pairs_per_point=pairs_per_point).to(dev).eval()
    g = 9000
    anchor = 0.6 * torch.randn(1, g, 11, device=dev)      # (crowded enough for cells with several anchors)
    feat = torch.randn(1, g, 128, device=dev)
    with torch.no_grad():
        whole = m(feat, anchor)
        total_whole = m.last_rulebook.check()
        idx0 = m.voxel_indices(anchor)
        conv_whole = subm_conv3d(feat[0], idx0, m.layer.weight, 1, m._spatial, 5)
        bounds = [0, 1, 1125, 4000, 4001, 8999, g]
        pairs = 0
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            part = m(feat, anchor, out_range=(lo, hi))
            assert part.shape == (1, hi - lo, 128)
            # (the output projection is a torch GEMM whose tiling changes with the row count: last-bit differences)
            assert torch.allclose(part, whole[:, lo:hi], rtol=1e-5, atol=1e-5), (lo, hi)
            pairs += m.last_rulebook.check()
            # the convolution itself: the slice's rows bit for bit, zeros elsewhere
            conv = subm_conv3d(feat[0], idx0, m.layer.weight, 1, m._spatial, 5, rulebook=m.last_rulebook)
            assert torch.equal(conv[lo:hi], conv_whole[lo:hi]), (lo, hi)
            assert float(conv[:lo].abs().sum()) == 0.0 and float(conv[hi:].abs().sum()) == 0.0
        assert pairs == total_whole                        # the slices' rulebooks partition the whole rulebook
        empty = m(feat, anchor, out_range=(77, 77))
        assert empty.shape == (1, 0, 128)
    idx = m.voxel_indices(anchor)
    rb = Rulebook(idx, 1, m._spatial, 5, out_range=(10, 20))
    with pytest.raises(RuntimeError):
        subm_conv3d(feat[0].clone().requires_grad_(True), idx, m.layer.weight, 1, m._spatial, 5, rulebook=rb)
    with pytest.raises(RuntimeError):
        Rulebook(idx, 1, m._spatial, 5, out_range=(5, g + 1))


@pytest.mark.gpu
@pytest.mark.parametrize("activation", ["sigmoid", "identity"])
def test_voxel_indices_in_one_launch_equal_the_torch_op_sequence(gpu, activation):
    """``gf_subm_voxelize`` (round 5) against the reference's dozen elementwise ops (spconv3d_module.py:56-66): the same int32
    indices, element for element, for a million anchors -- random, far out on both sides of the clamp, and exactly on cell
    boundaries of the grid."""
    import torch
    from gaussianformer_amd.sparse_conv import SparseConv3D
    pc_range, grid = [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], [0.5, 0.5, 0.5]
    blk = SparseConv3D(16, 16, pc_range, grid, xyz_activation=activation).to(gpu)
    g = torch.Generator().manual_seed(5)
    n = 1 << 19
    if activation == "sigmoid":
        a = torch.randn(2, n, 11, generator=g) * 3.0
        a[0, :1000, :3] = torch.linspace(-30, 30, 1000)[:, None]          # both sides of the clamp
        # centres that land exactly on cell boundaries: logit of (k * 0.5 + 50) / 100
        t = (torch.arange(1, 200, dtype=torch.float64) * 0.5) / 100.0
        a[1, :199, 0] = torch.log(t / (1 - t)).float()
    else:
        a = torch.rand(2, n, 11, generator=g)
        a[0, :100, :3] = torch.linspace(-0.5, 1.5, 100)[:, None]
        a[1, :199, 1] = ((torch.arange(1, 200, dtype=torch.float64) * 0.5) / 100.0).float()
    a = a.to(gpu)
    native = blk._voxel_indices_native(a)
    ref = blk._voxel_indices_torch(a)
    assert native.dtype == ref.dtype == torch.int32 and native.shape == ref.shape == (2 * n, 4)
    assert torch.equal(native, ref), int((native != ref).any(dim=1).sum())
    # a view with a row stride (anchors sliced out of a wider tensor) goes through the same entry
    wide = torch.cat([a, a], dim=-1)[..., :11]
    assert torch.equal(blk.voxel_indices(wide), ref)
