import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import oracle
from gaussianformer_amd import _lib
from gaussianformer_amd.local_aggregate import SplatForwardPlan
from gaussianformer_amd.synthetic import make_splat_inputs
dev = torch.device("cuda:0")
def run(si, tag):
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size, si.scale_multiplier)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    plan = SplatForwardPlan(0, *t, si.H, si.W, si.D, flags=1)
    lib = _lib.load()
    for _ in range(10): plan.run()
    torch.cuda.synchronize()
    lib.gf_profile_stride(1); lib.gf_profile_enable(200)
    for _ in range(200): plan.run()
    torch.cuda.synchronize()
    import ctypes
    buf = (ctypes.c_float*200)(); n = lib.gf_profile_read(buf, 200); lib.gf_profile_enable(0)
    # iterations (gaussian x double brick)
    r3 = np.repeat(radii[:,None],3,1); dims=np.array([si.H,si.W,si.D])
    lo = np.minimum(dims, np.maximum(0, mi-r3)); hi = np.minimum(dims, np.maximum(0, mi+r3+1))
    ok = (hi>lo).all(1)
    a = lo//np.array([4,4,8]); b=(hi-1)//np.array([4,4,8])
    it = ((b-a+1).prod(1)*ok).sum()
    pairs = ((hi-lo).prod(1)).sum()
    k = float(np.mean(buf[:n]))*1e3
    print(f"{tag}: render kernel {k:.1f} us, iterations {it}, pairs {pairs}, model iterations*190cyc/1024 SIMDs = {it*190/1024/2400:.1f} us, ratio {k/(it*190/1024/2400):.2f}")
si = make_splat_inputs("nuscenes_gs25600_solid", seed=0)
run(si, "random (bench workload)")
# uniform lattice: same count, equal scales, regular positions, no empty gaussian
P = 25600
si2 = make_splat_inputs("nuscenes_gs25600_solid", seed=0)
n = 40  # 40x40x16 = 25600 lattice
gx, gy, gz = np.meshgrid(np.arange(n), np.arange(n), np.arange(16), indexing="ij")
means = np.stack([(gx+0.5)*(100.0/n)-50, (gy+0.5)*(100.0/n)-50, (gz+0.5)*0.5-5], -1).reshape(-1,3).astype(np.float32)
for scale in (0.36, 0.5):
    si2.means3D = means; si2.scales = np.full((P,3), scale, np.float32)
    from gaussianformer_amd.synthetic import cov_inverse
    q = np.tile(np.array([[1.0,0,0,0]]), (P,1))
    si2.cov3D = cov_inverse(si2.scales, q).astype(np.float32)
    si2.opacities = si2.opacities[:P]; si2.semantics = si2.semantics[:P]
    run(si2, f"uniform lattice, scale {scale}")
