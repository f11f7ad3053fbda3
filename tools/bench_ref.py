"""The reference's own kernels and this repository's kernels on the same MI355X, same inputs (GPU box).

    python tools/bench_ref.py <case> <ref|hip> [reps]        # run under rocprofv3 --kernel-trace --stats
    python tools/bench_ref.py --summarise <dir> <out.txt>    # table from the stats CSVs tools/gpu/ref_compare.sh wrote

`ref` runs oracle/_ref (the reference's CUDA sources compiled for gfx950 where they lie, test infrastructure) through
its host-pointer wrapper, `hip` runs libgf_hip.so through the C ABI; every GPU kernel either side launches is counted,
host copies are not (they are not kernels).  One forward + one backward per repetition.
"""
import csv
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SPLAT_CASES = {"base_gs25600": ("nuscenes_gs25600_solid", False), "base_gs144000": ("nuscenes_gs144000", False),
               "prob_gs6400": ("prob_gs6400", False), "prob_fast_gs6400": ("prob_gs6400", True)}
DAF_CASES = {"daf_gs6400": 83200, "daf_gs25600": 230400}


def run(case, which, reps):
    import torch
    from gaussianformer_amd.synthetic import make_daf_inputs, make_splat_inputs
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    if case in SPLAT_CASES:
        from util import hip_splat_backward, hip_splat_forward, prep
        config, per_axis = SPLAT_CASES[case]
        si = make_splat_inputs(config, seed=0)
        pi, mi, radii, cov6 = prep(si, per_axis)
        N = si.pts.shape[0]
        g = rng.standard_normal((N, 18)).astype(np.float32)
        gb, gd = (rng.standard_normal(N).astype(np.float32) for _ in range(2))
        if si.variant != "prob":
            gb = gd = None
        for _ in range(reps):
            if which == "ref":
                from oracle import ref
                ref.splat_forward_backward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                                           si.H, si.W, si.D, g, gb, gd)
            else:
                _, t, state, fwd_t = hip_splat_forward(dev, si, pi, mi, radii, cov6)
                hip_splat_backward(dev, si, t, state, fwd_t, g, gb, gd)
    else:
        d = make_daf_inputs(num_pts=DAF_CASES[case], seed=0)
        go = rng.standard_normal((1, DAF_CASES[case], 128)).astype(np.float32)
        args = (d["mc_ms_feat"], d["spatial_shape"], d["scale_start_index"], d["sampling_location"], d["weights"])
        if which == "ref":
            from oracle import ref
            for _ in range(reps):
                ref.daf_forward(*args)
                ref.daf_backward(*args, go)
        else:
            from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction as DAF
            t = [torch.from_numpy(a).to(dev) for a in args]
            for i in (0, 3, 4):
                t[i].requires_grad_(True)
            gt = torch.from_numpy(go).to(dev)
            for _ in range(reps):
                out = DAF.apply(*t)
                out.backward(gt)
                for i in (0, 3, 4):
                    t[i].grad = None
        torch.cuda.synchronize()
    torch.cuda.synchronize()


def kernel_totals(d):
    """[(kernel name, calls, total ns)] from the rocprofv3 output under ``d`` (stats CSV, or the rocpd database)."""
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if files:
        return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(files[-1]))]
    dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getmtime)
    if not dbs:
        return []
    import sqlite3
    cur = sqlite3.connect(dbs[-1]).cursor()
    return [(n, int(c), float(t)) for n, c, t in cur.execute("select name, count(*), sum(duration) from kernels group by name")]


def is_backward(name):
    """Kernel belongs to the backward pass (either side)."""
    if name.startswith("void renderCUDA"):
        return "(int, unsigned int const*" in name        # backward.cu's renderCUDA takes (P, offsets, voxel2pts, ...)
    return any(k in name for k in ("bwd", "grad_kernel", "voxel2pts", "gf_daf_accumulate", "gf_daf_bucket", "gf_daf_tilescan", "gf_daf_colscan"))


def summarise(directory, out_path):
    lines = ["GPU kernel time per forward + backward on one MI355X, same inputs: the reference's own kernels (oracle/_ref, compiled for",
             "gfx950 from the sources where they lie) against libgf_hip.so.  rocprofv3 --kernel-trace --stats, every kernel either side",
             "launches, per repetition.  torch fill/copy kernels of the harness (uploading inputs) are listed but not counted.", ""]
    harness = ("at::native", "__amd_rocclr_copyBuffer")
    for case in list(SPLAT_CASES) + list(DAF_CASES):
        row = {}
        for which in ("ref", "hip"):
            ks = kernel_totals(os.path.join(directory, f"{case}_{which}"))
            if not ks:
                continue
            reps = int(open(os.path.join(directory, f"{case}_{which}", "reps")).read())
            counted = [(n, c, t) for n, c, t in ks if not n.startswith(harness) and "at::native" not in n]
            row[which] = (sum(t for _, _, t in counted) / reps / 1e3, counted, reps,
                          sum(t for n, _, t in counted if not is_backward(n)) / reps / 1e3)
        if len(row) < 2:
            continue
        lines.append(f"== {case}: reference {row['ref'][0]:10.1f} us   this repository {row['hip'][0]:10.1f} us   ratio {row['ref'][0] / row['hip'][0]:6.1f}x"
                     f"   | forward only: {row['ref'][3]:9.1f} us vs {row['hip'][3]:8.1f} us = {row['ref'][3] / row['hip'][3]:6.1f}x")
        for which in ("ref", "hip"):
            total, counted, reps, _ = row[which]
            for n, c, t in sorted(counted, key=lambda x: -x[2])[:7]:
                lines.append(f"     {which}  {t / reps / 1e3:10.1f} us  x{c / reps:<5.1f} {n[:110]}")
        lines.append("")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if sys.argv[1] == "--summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
