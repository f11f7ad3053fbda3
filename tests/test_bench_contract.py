"""The bench.py output contract, checked on the committed evidence line (profiles/bench_r*.json) and on the
pure helpers of bench.py -- no GPU needed."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r*.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1]))


def test_bench_line_has_the_contract_fields():
    b = _latest()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in b, key
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    # the first clause of BASELINE.json's metric (evidence lines recorded before the exact spelling was adopted
    # write the multiplication signs as "x")
    assert b["metric"].replace("x", "\u00d7") == base["metric"].split(";")[0].strip().replace("x", "\u00d7")
    assert b["unit"] == "Gaussians/s"
    assert b["n_gpus"] == 1 and b["higher_is_better"] is True and b["scaling"] in ("strong", "weak") and b["vs_baseline"] is None
    assert b["dtype"] == "f32" and b["data"] == "synthetic"
    assert "workload" in b["config"] and "nuscenes_gs25600_solid" in b["config"]["workload"]
    assert "model" not in b["config"]
    # value = units / time of a step
    assert abs(b["value"] - b["config"]["P_per_gpu"] / (b["ms_per_step"] * 1e-3)) <= 1e-6 * b["value"]
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes"] / (r["kernel_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes"]
    c = b["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_algorithmic_bytes_formula():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    # every op input read once + logits written once
    assert bench.algorithmic_bytes(25601, 640000) == 128 * 25601 + 24 * 640000 + 72 * 640000 == 64716928
    assert bench.algorithmic_bytes(144000, 640000) == 79872000
    t, step, note = bench.committed_traffic("nuscenes_gs25600_solid")
    assert t is None or (0.9 * 64716928 <= t <= 2 * 64716928 and "committed" in note["source"])
    assert step is None or step >= t        # the step's traffic = prep + render kernels
    assert bench.committed_traffic("no_such_config") == (None, None, None)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert bench.headline_metric() == base["metric"].split(";")[0].strip()


def test_torch_cpu_baseline_formulation_matches_the_oracle():
    """The PyTorch-CPU pair-list formulation timed by bench.py's ``cpu_baseline_torch`` leg computes the splat."""
    import numpy as np
    import torch
    import oracle
    from oracle.torch_cpu_splat import splat_forward_torch
    from gaussianformer_amd.synthetic import make_splat_inputs
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=3, P=300, H=24, W=20, D=16)
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size,
                                                      si.scale_multiplier)
    ref = oracle.splat_forward("base", si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6, si.H, si.W, si.D)["logits"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out = splat_forward_torch(t(pi), t(si.pts), t(si.means3D), t(mi), t(si.opacities), t(si.semantics), t(radii), t(cov6),
                              si.H, si.W, si.D, chunk_pairs=1 << 14).numpy()
    assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
