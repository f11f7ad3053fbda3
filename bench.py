#!/usr/bin/env python
"""Benchmark of the hot path: Gaussian -> voxel splat forward on synthetic nuScenes-shaped input.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the splat forward (C ABI ``gf_splat_forward``, automatic point-layout
detection, inputs and outputs resident in HBM) over one frame: P = 25 601 Gaussians
(``nuscenes_gs25600_solid``: 25 600 + the appended whole-grid "empty" Gaussian) into the
200x200x16 grid with 18 semantic channels.  With N > 1 ranks every rank splats its own shard of
P Gaussians into a full partial grid and the partial logits are summed with one RCCL
all-reduce (weak scaling: per-GPU work is fixed); ``value`` counts the Gaussians
of all ranks.  Rank 0 prints one JSON line.  Extras next to `value`, never instead of it: `two_stream` (N = 1, two
frames in flight), `hip_graph` (N = 1, the step replayed as one captured HIP graph) and `reduce_scatter_labels` (N > 1, the label-producing variant with half the xGMI traffic).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PROFILE_STRIDE = 8
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling


def headline_metric():
    """The first clause of BASELINE.json's metric (its second, "frames/sec end-to-end", belongs to the full model)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"].split(";")[0].strip()
    except Exception:
        return "Gaussians/sec splatted into 200\u00d7200\u00d716\u00d718 voxel grid (fwd)"


def algorithmic_bytes(P, N, C=18):
    """every op input read once + logits written once."""
    return 128 * P + 24 * N + 4 * C * N


def measured_traffic_bytes(config):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this workload
    (profiles/traffic_*.json, FETCH_SIZE x2 correction applied there); None if absent."""
    import glob
    pattern = {"nuscenes_gs25600_solid": "traffic_r*.json", "nuscenes_gs144000": "traffic_gs144000_r*.json"}.get(config)
    if pattern is None:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    try:
        return json.load(open(files[-1])).get("render_kernel_hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(si, pi, mi, radii, cov6, budget_s=10.0):
    """The CPU oracle (a restatement of the reference kernels, kind "port") timed on this
    box's host cores on the SAME workload, all OpenMP threads."""
    import oracle
    threads = oracle.num_threads()
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 or ((time.perf_counter() - t_start) < budget_s and len(times) < 1000):
        t0 = time.perf_counter()
        oracle.splat_forward(si.variant, si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6,
                             si.H, si.W, si.D, nthreads=threads)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    P = si.means3D.shape[0]
    return {"value": P / t, "unit": "Gaussians/s", "cores": threads, "kind": "port",
            "sample": f"{len(times)} full forward passes of the same workload (P={P}, N={si.pts.shape[0]}) "
                      f"in {sum(times):.1f} s of wall time on {threads} threads, median pass",
            "seconds_per_pass": t}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="nuscenes_gs25600_solid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-stream", action="store_true",
                    help="skip the extra two-frames-in-flight measurement (used for the rocprofv3 kernel trace, whose "
                         "per-kernel average would otherwise mix overlapped and sequential launches)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import oracle  # checker side only: host pre-processing restatement + cpu_baseline leg
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import SplatForwardPlan
    from gaussianformer_amd.synthetic import make_splat_inputs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; there is no CPU path for the product")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # GF_BENCH_FORCE_DIST=1 takes the RCCL path with a single rank too (a 1-GPU check of the plumbing)
    use_dist = world > 1 or os.environ.get("GF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()

    # each rank owns a different shard of Gaussians (seed = rank); same query grid
    si = make_splat_inputs(args.config, seed=rank)
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min,
                                                      si.grid_size, si.scale_multiplier,
                                                      radii_min=1 if si.variant == "prob" else None)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev)
         for a in (si.pts, pi, si.means3D, mi, si.opacities, si.semantics, radii, cov6)]
    variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
    plan = SplatForwardPlan(variant, *t, si.H, si.W, si.D, flags=_lib.GF_PTS_AUTO)
    P, N = plan.P, plan.N
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        plan.run(stream)
        if use_dist:
            dist.all_reduce(plan.logits, op=dist.ReduceOp.SUM)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # the dominant kernel is timed with hipEvents on every PROFILE_STRIDE-th launch of the timed
    # region (an event pair costs ~3 us of stream time; sampling keeps the region representative)
    _lib.check(lib.gf_profile_stride(PROFILE_STRIDE), "gf_profile_stride")
    _lib.check(lib.gf_profile_enable(args.steps), "gf_profile_enable")
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # dominant kernel (the render kernel): per-launch duration from the hipEvents recorded
    # around it on the launch stream during the timed region
    buf = (ctypes.c_float * args.steps)()
    n_ev = lib.gf_profile_read(buf, args.steps)
    lib.gf_profile_enable(0)
    lib.gf_profile_stride(1)
    kernel_ms = float(np.mean(buf[:n_ev])) if n_ev > 0 else None

    # Extra, N=1 only: the same K steps with two frames in flight -- two pre-bound plans (own
    # outputs / workspace) alternating on two HIP streams, so the latency-bound prep kernel, the cold
    # start and the tail of one step overlap the render kernel of the other.  Reported next to
    # `value` (which stays the strict one-step-at-a-time figure), never instead of it.
    two_stream = None
    if world == 1 and not args.no_two_stream:
        plans = [plan, SplatForwardPlan(variant, *t, si.H, si.W, si.D, flags=_lib.GF_PTS_AUTO)]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        for i in range(2 * max(2, args.warmup // 2)):
            plans[i % 2].run(streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            plans[i % 2].run(streams[i % 2].cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        two_stream = {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                      "note": "same K steps, two frames in flight on two HIP streams (double-buffered outputs and workspace)"}

    # Extra, N > 1 only: the same K steps ending in occupancy labels instead of replicated logits --
    # reduce-scatter of the partial grids (each rank receives the summed logits of its 1/N of the
    # voxels), labels on the owned slab (gf_head_labels), all-gather of the labels: half the xGMI
    # traffic of the all-reduce (gaussianformer_amd.head.sharded_splat_labels).  Never `value`.
    rs_labels = None
    if use_dist and si.variant != "prob" and N % world == 0:
        try:
            from gaussianformer_amd.head import occupancy_labels
            mine = torch.empty(N // world, 18, dtype=torch.float32, device=dev)
            labels = torch.empty(N, dtype=torch.int64, device=dev)

            def label_step():
                plan.run(stream)
                dist.reduce_scatter_tensor(mine, plan.logits, op=dist.ReduceOp.SUM)
                dist.all_gather_into_tensor(labels, occupancy_labels(mine))

            for _ in range(max(2, args.warmup // 2)):
                label_step()
            dist.barrier()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                label_step()
            torch.cuda.synchronize()
            dist.barrier()
            tt = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            rs_labels = {"value": world * P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                         "note": "same K steps ending in labels: reduce-scatter of the partial logits, labels on the owned "
                                 "slab, all-gather of the labels"}
        except Exception as exc:  # an extra must never cost the headline line
            rs_labels = {"error": f"{type(exc).__name__}: {exc}"}

    # Extra, N = 1 only: the same step captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed K times --
    # what a serving loop that splats the same buffers every frame would do.  Never `value`.
    hip_graph = None
    if world == 1 and not args.no_two_stream:
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                plan.run()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                plan.run()
            for _ in range(max(2, args.warmup // 2)):
                graph.replay()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t3
            hip_graph = {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                         "note": "same K steps as replays of one captured HIP graph (prep + render)"}
        except Exception as exc:  # an extra must never cost the headline line
            hip_graph = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * P / (elapsed / args.steps)
        abytes = algorithmic_bytes(P, N)
        roofline = None
        if kernel_ms:
            achieved = abytes / (kernel_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS,
                        "traffic": measured_traffic_bytes(args.config),
                        "kernel": "gf_splat_render_kernel", "kernel_us": kernel_ms * 1e3,
                        "kernel_launches_timed": n_ev,
                        "algorithmic_bytes": abytes}
        out = {
            "metric": headline_metric(),
            "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: splat forward, P={P} Gaussians/GPU -> {si.H}x{si.W}x{si.D}x18 "
                                   f"grid (N={N} voxel-centre points), bs=1",
                       "P_per_gpu": P, "N": N, "pts_layout": "auto-detected dense grid",
                       "parallelism": "single GPU" if world == 1 else f"gaussian-shard x{world} + RCCL all-reduce of logits"},
            "roofline": roofline,
        }
        if two_stream:
            out["two_stream"] = two_stream
        if hip_graph:
            out["hip_graph"] = hip_graph
        if rs_labels:
            out["reduce_scatter_labels"] = rs_labels
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(si, pi, mi, radii, cov6)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
