"""bench.py's own code paths on the GPU box: the N = 1 line (reduced steps) and the N = 2 strong-scaling path
(`gaussianformer_amd.sharded`), run as two real processes.  The box has one GPU, so the two ranks share it and the
collective goes through gloo on host copies (GF_BENCH_SHARED_GPU=1); everything else -- sharding, per-rank plans, the
timed loop, max over ranks, the JSON line -- is the code the driver runs with RCCL on an 8-GPU node.  GF_BENCH_CHECK=1
makes bench.py compare the sharded sum with the single-device result of the whole set."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_line(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 1 and b["scaling"] == "strong" and b["config"]["P_total"] == 25601 == b["config"]["P_per_gpu"]
    assert abs(b["value"] - 25601 / (b["ms_per_step"] * 1e-3)) <= 1e-6 * b["value"]
    assert b["roofline"]["kernel_launches_timed"] >= 8 and 0 < b["roofline"]["frac"] < 1
    f = b["frames_per_s"]
    assert "error" not in f, f
    for cfg in ("nuscenes_gs25600_solid", "nuscenes_gs144000"):
        assert f[cfg]["frames_per_s"] > 0 and f[cfg]["voxels"] == 640000
        # the frame is reproducible -- eager twice and eager against the captured graph give the same labels --
        # and the comparison is not vacuous (more than one class wins somewhere)
        assert f[cfg]["labels_used"] > 1, f[cfg]["label_histogram"]
        assert f[cfg]["eager_labels_equal_eager"] is True
        assert f[cfg].get("graph_labels_equal_eager") is True, f[cfg]
    assert b["dtype"] == "f32" and "f16" in b["arithmetic"]
    g3 = b["gs144000_forward"]
    assert "error" not in g3 and g3["roofline"]["kernel_launches_timed"] >= 8 and 0 < g3["roofline"]["frac"] < 1
    v1 = b["verified_once"]
    assert "error" not in v1 and v1["bit_identical_to_headline"] is True
    bw = b["splat_backward"]
    assert "error" not in bw and bw["finite"] is True and bw["forward_prepared_rows"] is True
    assert 0 < bw["us_per_call"] < bw["exact_fp32_us_per_call"] * 1.2
    b3 = g3["backward"]   # round 6: config [3]'s backward on the matrix cores (long rows)
    assert "error" not in b3 and b3["forward_prepared_rows"] is True and 0 < b3["us_per_call"] < b3["exact_fp32_us_per_call"]


def test_bench_two_rank_strong_scaling_path(gpu):
    env = dict(os.environ, GF_BENCH_SHARED_GPU="1", GF_BENCH_CHECK="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    b = _last_json(r.stdout)
    assert b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert b["config"]["P_total"] == 25601 and b["config"]["P_per_gpu"] == 12801      # shard_bounds(25601, 0, 2)
    assert b["check_max_scaled_err_vs_single_device"] <= 1e-4
    assert b["kernel_only"]["ms_per_step"] <= b["ms_per_step"]
    # round 6: the exchange of the partial grids is chosen in the warm-up among the three of sharded.sum_across_ranks
    assert b["exchange"] in ("all_reduce", "direct", "reduce_scatter") and set(b["exchange_autotune_ms_per_step"]) == {"all_reduce", "direct", "reduce_scatter"}
    assert b["exchange"] in b["config"]["parallelism"]
    # round 6: the scaling curve's numbers as flat keys of the line
    assert b["kernel_only_ms_per_step"] == b["kernel_only"]["ms_per_step"] and b["collective_ms_per_step"] >= 0
    for key in ("gs25600", "gs144000"):
        for head in ("slab", "allreduce"):
            assert b[f"frames_per_s_sharded_{key}_{head}"] > 0
    g = b["gs144000"]
    assert "error" not in g and g["kernel_only_ms_per_step"] > 0 and "144000" in g["config"]
    fs = b["frame_sharded"]
    assert "error" not in fs, fs
    for cfg in ("nuscenes_gs25600_solid", "nuscenes_gs144000"):
        for head in ("slab", "allreduce"):
            r = fs[cfg][head]
            assert r["ms_per_frame"] > 0
            # the anchor-sharded frame labels the grid like the single-GPU frame (GEMM tilings change with the row count: last-bit
            # ties flip -- with random weights 0.07 - 0.18 % of the voxels at gs144000, varying between launches of the SAME
            # library: 0.99926 and 0.9982 were both seen in round 6 with nothing changed in between, while each native kernel and
            # the whole single-GPU frame repeat bit for bit, tools/long_rows_repro.py)
            assert r["labels_equal_single_gpu_fraction"] >= 0.995, (cfg, head, r)
    for key in ("slab_partition", "slab_partition_gs144000"):
        sp = b[key]
        assert "error" not in sp, sp
        assert sp["check_bit_identical_to_single_device"] is True      # no reduction: the gathered grid IS the single-GPU grid
        assert sp["splat_only"]["ms_per_step"] <= sp["all_gather_logits"]["ms_per_step"]


def test_bench_gpus_n_spawns_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset) re-executes itself under
    torch.distributed.run -- one line, printed by rank 0, with n_gpus 2 -- instead of warning and measuring one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(GF_BENCH_SHARED_GPU="1", GF_BENCH_CHECK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["P_per_gpu"] == 12801 and b["timing"]["loops"] == 5
