"""Generates the frozen fixtures under tests/golden/ by running the REFERENCE's own kernels
(oracle/_ref: the reference's CUDA sources compiled for gfx950 by oracle/ref_build.py) on fixed
seeded inputs.  Needs a GPU:

    gpurun -- 'GF_GOLDEN_OUT=gpurun_out/golden python tools/make_golden.py'
    cp gpurun_out/golden/*.npz tests/golden/

Every output array in the fixtures (logits, bin_logits, density, probability, the four gradients,
num_rendered / tiles_touched / offsets, the deformable-aggregation output and gradients) is what the
reference computed; the inputs are the seeded synthetic cases below.  The CPU tests check the C
restatement (oracle/gf_oracle.c) against them, the GPU tests check the HIP path against them.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from oracle import ref  # noqa: E402
from gaussianformer_amd.synthetic import make_daf_inputs, make_splat_inputs  # noqa: E402

OUT = os.environ.get("GF_GOLDEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
PRODUCER = "oracle/_ref: reference CUDA kernels compiled for gfx950 (hipcc), executed on MI355X"

SPLAT_CASES = {
    "splat_base": dict(config="nuscenes_gs25600_solid", P=48, H=12, W=10, D=8, per_axis=False),
    "splat_base_signed": dict(config="nuscenes_gs144000", P=64, H=12, W=12, D=8, per_axis=False),
    "splat_prob": dict(config="prob_gs6400", P=24, H=12, W=10, D=8, per_axis=False),
    "splat_prob_fast": dict(config="prob_gs6400", P=24, H=12, W=10, D=8, per_axis=True),
}


def splat_case(name, c):
    si = make_splat_inputs(c["config"], seed=100, P=c["P"], H=c["H"], W=c["W"], D=c["D"])
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min,
                                                      si.grid_size, si.scale_multiplier, per_axis=c["per_axis"],
                                                      radii_min=1 if si.variant == "prob" else None)
    rng = np.random.default_rng(101)
    N = si.pts.shape[0]
    g = rng.standard_normal((N, 18)).astype(np.float32)
    gb = rng.standard_normal(N).astype(np.float32)
    gd = rng.standard_normal(N).astype(np.float32)
    fwd, grads, _ = ref.splat_forward_backward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii,
                                               cov6, si.H, si.W, si.D, g, gb if si.variant == "prob" else None,
                                               gd if si.variant == "prob" else None)
    binning = ref.splat_forward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                si.H, si.W, si.D, with_binning=True)
    touched, offsets, R = binning["tiles_touched"], binning["point_offsets"], binning["num_rendered"]
    assert all(np.isfinite(v).all() for v in list(grads) + [fwd["logits"]]), name
    d = dict(producer=PRODUCER, variant=si.variant, H=si.H, W=si.W, D=si.D, grid_size=si.grid_size, pc_min=np.array(si.pc_min),
             scale_multiplier=si.scale_multiplier, per_axis=c["per_axis"],
             pts=si.pts, means3D=si.means3D, opacities=si.opacities, semantics=si.semantics, scales=si.scales,
             cov3D=si.cov3D, points_int=pi, means_int=mi, radii=radii, cov6=cov6,
             logits=fwd["logits"], num_rendered=R, tiles_touched=touched, offsets=offsets,
             out_grad=g, bin_grad=gb, density_grad=gd,
             means3D_grad=grads[0], opacity_grad=grads[1], semantics_grad=grads[2], cov3D_grad=grads[3])
    if si.variant == "prob":
        d.update(bin_logits=fwd["bin_logits"], density=fwd["density"], probability=fwd["probability"])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)


def daf_case():
    d = make_daf_inputs(num_pts=40, seed=102, B=2, cams=3, C=16, G=4, levels=((6, 9), (3, 5), (2, 2)))
    d["sampling_location"][0, 0, 0] = [0.0, 0.5]
    d["sampling_location"][0, 1, 0] = [0.999, 0.001]
    out = ref.daf_forward(**d)
    g = np.random.default_rng(103).standard_normal(out.shape).astype(np.float32)
    gf, gl, gw = ref.daf_backward(d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"],
                                  d["sampling_location"], d["weights"], g)
    np.savez_compressed(os.path.join(OUT, "daf.npz"), **d, producer=PRODUCER, output=out, grad_output=g, grad_mc_ms_feat=gf,
                        grad_sampling_location=gl, grad_weights=gw)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, c in SPLAT_CASES.items():
        splat_case(name, c)
    daf_case()
    print(sorted(os.listdir(OUT)))
