"""Seeded synthetic inputs shaped like the reference's nuScenes configs.

No dataset or checkpoint is available offline, so every test and benchmark runs on these.
Shapes and distributions follow the reference configs:
  - grid 200x200x16, cell 0.5 m, pc_min (-50,-50,-5)   (config/nuscenes_gs25600_solid.py:185-190)
  - query points = voxel centres, x-major / z-fastest    (dataset/transform_3d.py:487-499)
  - scales ~ U(scale_range) per axis (lifter init, model/lifter/gaussian_lifter.py:34-36)
  - the appended whole-grid "empty" Gaussian              (model/head/gaussian_head.py:90-102)
  - feature pyramid 108x200 / 54x100 / 27x50 / 14x25, 6 cams, 128 ch, 4 groups
                                                         (config/_base_/model.py:2-7,32-40)
"""
from dataclasses import dataclass

import numpy as np

GRID = dict(H=200, W=200, D=16, grid_size=0.5, pc_min=(-50.0, -50.0, -5.0))

SPLAT_CONFIGS = {
    # name: (variant, P (before empty), scale_range, scale_multiplier, with_empty, semantics kind)
    "nuscenes_gs25600_solid": dict(variant="base", P=25600, scale_range=(0.08, 0.64), scale_multiplier=3,
                                   with_empty=True, sem="softplus17"),
    "nuscenes_gs144000": dict(variant="base", P=144000, scale_range=(0.08, 0.32), scale_multiplier=3,
                              with_empty=False, sem="normal18"),
    "prob_gs6400": dict(variant="prob", P=6400, scale_range=(0.01, 3.2), scale_multiplier=4,
                        with_empty=False, sem="softmax17"),
}

DAF_LEVELS = ((108, 200), (54, 100), (27, 50), (14, 25))


@dataclass
class SplatInputs:
    """Arguments of ``LocalAggregator.forward`` (batch dim squeezed) + grid constants."""
    variant: str
    pts: np.ndarray        # [N,3] f32
    means3D: np.ndarray    # [P,3] f32
    opacities: np.ndarray  # [P]   f32
    semantics: np.ndarray  # [P,C] f32
    scales: np.ndarray     # [P,3] f32
    cov3D: np.ndarray      # [P,3,3] f32 (inverse covariance)
    H: int
    W: int
    D: int
    grid_size: float
    pc_min: tuple
    scale_multiplier: float


def voxel_centres(H, W, D, grid_size, pc_min):
    """Voxel-centre query points in x-major / z-fastest order (dataset/transform_3d.py:487-499)."""
    xs = (np.arange(H, dtype=np.float32) + np.float32(0.5)) * np.float32(grid_size) + np.float32(pc_min[0])
    ys = (np.arange(W, dtype=np.float32) + np.float32(0.5)) * np.float32(grid_size) + np.float32(pc_min[1])
    zs = (np.arange(D, dtype=np.float32) + np.float32(0.5)) * np.float32(grid_size) + np.float32(pc_min[2])
    g = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), axis=-1)
    return np.ascontiguousarray(g.reshape(-1, 3), dtype=np.float32)


def quat_to_rot(q):
    """Unit quaternion (w,x,y,z) -> rotation matrix; same matrix as
    model/utils/utils.py:20-69 produces (standard Hamilton convention)."""
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3), dtype=q.dtype)
    R[..., 0, 0] = 1 - 2 * (y * y + z * z)
    R[..., 0, 1] = 2 * (x * y - w * z)
    R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z)
    R[..., 1, 1] = 1 - 2 * (x * x + z * z)
    R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y)
    R[..., 2, 1] = 2 * (y * z + w * x)
    R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def cov_inverse(scales, quats):
    """Sigma^-1 for Sigma = (S R)^T (S R) (model/head/gaussian_head.py:111-119), computed
    in closed form in fp64: Sigma^-1 = R^T S^-2 R."""
    R = quat_to_rot(quats.astype(np.float64))
    s2 = 1.0 / (scales.astype(np.float64) ** 2)
    return np.einsum("pki,pk,pkj->pij", R, s2, R)


def make_splat_inputs(config="nuscenes_gs25600_solid", seed=0, P=None, H=None, W=None, D=None,
                      dense_pts=True, N=None, C=18, clustered=False):
    """Synthetic splat inputs.  ``P``/``H``/``W``/``D`` override the config for small cases;
    ``dense_pts=False`` draws ``N`` random query points instead of all voxel centres; ``clustered=True`` draws the
    centres as sigmoid(N(0, 1)) of the range per axis -- the lifter's initial anchors (model/lifter/gaussian_lifter.py:
    28-33), denser towards the middle of the scene like real nuScenes Gaussians -- instead of uniformly."""
    cfg = SPLAT_CONFIGS[config]
    rng = np.random.default_rng(seed)
    H = H or GRID["H"]
    W = W or GRID["W"]
    D = D or GRID["D"]
    gs, pc_min = GRID["grid_size"], GRID["pc_min"]
    P = cfg["P"] if P is None else P
    ext = np.array([H, W, D], dtype=np.float64) * gs
    lo = np.array(pc_min, dtype=np.float64)
    if clustered:
        u = 1.0 / (1.0 + np.exp(-rng.standard_normal((P, 3))))
        means = (lo + np.clip(u, 0.001, 0.999) * ext).astype(np.float32)
    else:
        means = (lo + (0.001 + 0.998 * rng.random((P, 3))) * ext).astype(np.float32)
    smin, smax = cfg["scale_range"]
    scales = (smin + (smax - smin) * rng.random((P, 3))).astype(np.float32)
    quats = rng.standard_normal((P, 4))
    if cfg["sem"] == "softplus17":
        sem = np.log1p(np.exp(rng.standard_normal((P, C - 1))))
        sem = np.concatenate([sem, np.zeros((P, 1))], axis=1)
        opa = rng.random(P)
    elif cfg["sem"] == "normal18":
        sem = rng.standard_normal((P, C))
        opa = np.ones(P)
    else:  # softmax17 (prob)
        z = rng.standard_normal((P, C - 1))
        z = np.exp(z - z.max(1, keepdims=True))
        sem = np.concatenate([z / z.sum(1, keepdims=True), np.zeros((P, 1))], axis=1)
        opa = 1.0 / (1.0 + np.exp(-rng.standard_normal(P)))
    if cfg["with_empty"]:
        # gaussian_head.py:90-102 with config empty_args mean [0,0,-1], scale [100,100,8]
        c = lo + ext / 2
        means = np.concatenate([means, np.array([[c[0], c[1], lo[2] + ext[2] * 0.5]], dtype=np.float32)])
        scales = np.concatenate([scales, np.array([[100.0, 100.0, 8.0]], dtype=np.float32)])
        quats = np.concatenate([quats, np.array([[1.0, 0.0, 0.0, 0.0]])])
        e = np.zeros((1, C))
        e[0, C - 1] = 10.0
        sem = np.concatenate([sem, e])
        opa = np.concatenate([opa, np.ones(1)])
    cov = cov_inverse(scales, quats).astype(np.float32)
    if dense_pts:
        pts = voxel_centres(H, W, D, gs, pc_min)
    else:
        N = N or 1000
        pts = (lo + (0.001 + 0.998 * rng.random((N, 3))) * ext).astype(np.float32)
    return SplatInputs(cfg["variant"], pts, means.astype(np.float32), opa.astype(np.float32),
                       np.ascontiguousarray(sem, dtype=np.float32), scales, cov, H, W, D, gs, pc_min,
                       cfg["scale_multiplier"])


def make_daf_inputs(num_pts=230400, seed=0, B=1, cams=6, C=128, G=4, levels=DAF_LEVELS, mask_frac=0.3):
    """Op-level deformable-aggregation inputs (loc ~ U(-0.2,1.2), weights =
    softmax over (cams, levels) of N(0,1) with ``mask_frac`` masked to zero)."""
    rng = np.random.default_rng(seed)
    spatial_shape = np.array(levels, dtype=np.int32)
    sizes = spatial_shape[:, 0] * spatial_shape[:, 1]
    start = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    num_feat = int(sizes.sum())
    L = len(levels)
    feat = rng.standard_normal((B, cams, num_feat, C), dtype=np.float32)
    loc = (-0.2 + 1.4 * rng.random((B, num_pts, cams, 2))).astype(np.float32)
    logits = rng.standard_normal((B, num_pts, cams * L, G)).astype(np.float32)
    keep = rng.random((B, num_pts, cams * L, G)) >= mask_frac
    logits = np.where(keep, logits, -np.inf)
    m = np.max(logits, axis=2, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    e = np.exp(logits - m)
    w = e / np.maximum(e.sum(axis=2, keepdims=True), 1e-20)
    weights = np.ascontiguousarray(w.reshape(B, num_pts, cams, L, G), dtype=np.float32)
    return dict(mc_ms_feat=feat, spatial_shape=spatial_shape, scale_start_index=start,
                sampling_location=loc, weights=weights)
