#!/usr/bin/env python
"""Benchmark of the hot path: Gaussian -> voxel splat forward on synthetic nuScenes-shaped input.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the splat forward (C ABI ``gf_splat_forward``, automatic point-layout detection, inputs and
outputs resident in HBM) over ONE frame: the P = 25 601 Gaussians of ``nuscenes_gs25600_solid`` (25 600 + the appended
whole-grid "empty" Gaussian) into the 200x200x16 grid with 18 semantic channels.

N = 1: ``value`` = P / step time.  N > 1 is STRONG scaling of the same frame, the experiment north_star names: the one
Gaussian set is cut into N contiguous shards (``gaussianformer_amd.sharded.shard_bounds``), every rank splats its shard
into a full partial grid and the partial logits are summed with one RCCL all-reduce over xGMI
(``sharded.sharded_splat_forward``); ``value`` = P / (splat + all-reduce) time, max over ranks.  The reference itself
only runs data-parallel replicas (train.py:41-43, :86-91).

Rank 0 prints ONE JSON line.  Next to ``value``, never instead of it:
  N = 1  ``roofline`` (render kernel, hipEvents on the launch stream in a separate loop after the headline's K steps;
         ``op_frac`` = the whole op by the headline's own clock), ``cpu_baseline`` (C port of the reference kernels)
         and ``cpu_baseline_torch`` (vectorised PyTorch-CPU pair-list formulation), ``frames_per_s`` (one inference frame
         of the whole hot path: 4 encoder blocks + head, tools/bench_frame.py; eager and as one HIP graph), ``train_step``
         (BASELINE config [2]: the native ops of a training step chained through autograd, tools/bench_step.py),
         ``exact_fp32_kernel``, ``two_stream``, ``hip_graph``;
  N > 1  ``kernel_only`` (the same shards without the collective), ``gs144000`` (BASELINE config [4]: the 144 000-Gaussian
         set sharded N-way, with and without the all-reduce), ``reduce_scatter_labels`` (reduce-scatter -> labels on the
         owned slab -> all-gather of labels) and ``slab_partition`` / ``slab_partition_gs144000`` (the SPATIAL partition:
         every rank renders its band of voxel rows from all Gaussians, no reduction, all-gather of logits or labels).
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling
KERNEL_SAMPLES = 16   # bracketed launches of the separate kernel-timing loop (>= 8)
SETTLE_STEPS = 300     # untimed steps ahead of the W warm-up steps (GPU clock settle, ~15 ms); reported in the JSON line


def headline_metric():
    """The first clause of BASELINE.json's metric (its second, "frames/sec end-to-end", is the ``frames_per_s`` block)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"].split(";")[0].strip()
    except Exception:
        return "Gaussians/sec splatted into 200×200×16×18 voxel grid (fwd)"


def algorithmic_bytes(P, N, C=18):
    """every op input read once + logits written once."""
    return 128 * P + 24 * N + 4 * C * N


def committed_traffic(config):
    """HBM bytes per launch from the committed PMC passes of this workload (profiles/traffic_*.json; FETCH_SIZE x2
    correction applied there).  The counters cannot be read inside this process, so the figures are labelled with
    their source.  Returns (render-kernel bytes, step bytes = prep + render kernels, note) or (None, None, None)."""
    import glob
    pattern = {"nuscenes_gs25600_solid": "traffic_r*.json", "nuscenes_gs144000": "traffic_gs144000_r*.json"}.get(config)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))) if pattern else []
    if not files:
        return None, None, None
    try:
        d = json.load(open(files[-1]))
        render, prep = d.get("render_kernel_hbm_bytes_per_launch"), d.get("prep_kernel_hbm_bytes_per_launch")
        step = d.get("step_hbm_bytes", (render + prep) if (render is not None and prep is not None) else None)
        return render, step, {
            "source": "committed rocprofv3 PMC pass " + os.path.relpath(files[-1], ROOT) + " (not measured in this run)",
            "render_kernel": d.get("render_kernel"), "prep_kernel_hbm_bytes_per_launch": prep}
    except Exception:
        return None, None, None


def cpu_baseline(si, pi, mi, radii, cov6, budget_s=10.0):
    """The CPU oracle (a C restatement of the reference kernels, pinned against oracle/_ref; kind "port") timed on
    this box's host cores on the SAME workload, all OpenMP threads."""
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


def cpu_baseline_torch(si, pi, mi, radii, cov6, budget_s=20.0):
    """A vectorised PyTorch-CPU formulation (pair list -> index_add_, oracle/torch_cpu_splat.py): what a pure-PyTorch
    fallback of the reference's splat would be (the reference has none).  Bounded sample: every 8th
    Gaussian of the frame first; the full frame only if that took under budget / 10."""
    import torch
    from oracle.torch_cpu_splat import splat_forward_torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    P = si.means3D.shape[0]

    def timed(sel):
        args = (t(pi), t(si.pts), t(si.means3D[sel]), t(mi[sel]), t(si.opacities[sel]), t(si.semantics[sel]), t(radii[sel]), t(cov6[sel]))
        t0 = time.perf_counter()
        splat_forward_torch(*args, si.H, si.W, si.D)
        return time.perf_counter() - t0
    sel = np.arange(0, P, 8)
    dt = timed(sel)
    sample = f"every 8th Gaussian of the frame ({len(sel)} of {P}, into the full N={si.pts.shape[0]} grid), one pass"
    n = len(sel)
    if dt < budget_s / 10:
        sel = np.arange(P)
        dt = timed(sel)
        n = P
        sample = f"one full forward pass of the same workload (P={P})"
    return {"value": n / dt, "unit": "Gaussians/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": sample + f", {dt:.1f} s on {torch.get_num_threads()} torch threads", "seconds": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="nuscenes_gs25600_solid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-two-stream", dest="no_extras", action="store_true",
                    help="headline measurement only (used for the rocprofv3 kernel trace, whose per-kernel averages would "
                         "otherwise mix the extras' launches into the headline's)")
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): re-exec under torch.distributed.run, one
    # rank per GPU -- the same command line the driver uses for N > 1; rank 0 of the child prints the one JSON line (VERDICT r4:
    # this form used to warn and measure ONE GPU, reporting n_gpus 1).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    import oracle  # checker side only: host pre-processing restatement + cpu_baseline legs
    from gaussianformer_amd import _lib
    from gaussianformer_amd.local_aggregate import SplatForwardPlan
    from gaussianformer_amd.sharded import EXCHANGES, shard_bounds, sharded_splat_forward, sum_across_ranks
    from gaussianformer_amd.synthetic import make_splat_inputs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X; there is no CPU path for the product")
    # GF_BENCH_SHARED_GPU=1 (tests on a 1-GPU box): every rank uses cuda:0 and the collectives go through gloo on host copies
    shared_gpu = os.environ.get("GF_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # GF_BENCH_FORCE_DIST=1 takes the collective path with a single rank too (a 1-GPU check of the plumbing)
    use_dist = world > 1 or os.environ.get("GF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()

    def all_reduce(t, op=None):
        op = op or dist.ReduceOp.SUM
        if shared_gpu:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def max_over_ranks(seconds):
        if not use_dist:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device=dev)
        all_reduce(tt, dist.ReduceOp.MAX)
        return float(tt.item())

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    class Workload:
        """One frame's Gaussian set (seed 0, identical on every rank), this rank's contiguous shard bound to a plan."""
        exchange = "all_reduce"   # how the partial grids are summed at N > 1 (sharded.sum_across_ranks); chosen in the warm-up below

        def __init__(self, config):
            self.config = config
            si = make_splat_inputs(config, seed=0)
            self.si = si
            self.prep = oracle.prepare_splat_inputs(si.pts, si.means3D, si.scales, si.cov3D, si.pc_min, si.grid_size,
                                                    si.scale_multiplier, radii_min=1 if si.variant == "prob" else None)
            pi, mi, radii, cov6 = self.prep
            self.P_total, self.N = si.means3D.shape[0], si.pts.shape[0]
            self.lo, self.hi = shard_bounds(self.P_total, rank, world)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            sl = slice(self.lo, self.hi)
            self.tensors = [up(si.pts), up(pi), up(si.means3D[sl]), up(mi[sl]), up(si.opacities[sl]), up(si.semantics[sl]),
                            up(radii[sl]), up(cov6[sl])]
            self.variant = _lib.GF_SPLAT_PROB if si.variant == "prob" else _lib.GF_SPLAT_BASE
            self.plan = SplatForwardPlan(self.variant, *self.tensors, si.H, si.W, si.D, flags=_lib.GF_PTS_AUTO)
            self.stream = torch.cuda.current_stream(dev).cuda_stream
            # full-set views for sharded_splat_forward (it slices [lo, hi) itself; the plan is bound to exactly that slice)
            self.full = [torch.empty(1, self.P_total, 0, device=dev)] * 5

        def local(self, *_):
            return self.plan.run(self.stream)

        def path(self):
            """Which body rendered the last call (state block, word 1) and the device verdict bits (word 2)."""
            w = self.plan.state_words()
            names = {_lib.GF_PATH_EXACT_TILE: "gf_splat_render_kernel (exact-fp32 tile kernel)",
                     _lib.GF_PATH_MATRIX_CORE: "gf_splat_render_mfma_kernel (split-f16 MFMA, fp32 accumulate; one workgroup per tile)",
                     _lib.GF_PATH_MATRIX_CORE_WAVE: "gf_splat_render_mfma_wave_kernel (split-f16 MFMA, fp32 accumulate; one wave per double brick)",
                     _lib.GF_PATH_MATRIX_CORE_PAIR: "gf_splat_render_mfma_pair_kernel (split-f16 MFMA, fp32 accumulate; two waves per double brick, one brick each)",
                     _lib.GF_PATH_MATRIX_CORE_SOLO: "gf_splat_render_mfma_solo_kernel (split-f16 MFMA, fp32 accumulate; one wave per double brick, round-5 instruction diet)",
                     _lib.GF_PATH_ARBITRARY: "arbitrary-points body (FALL-BACK: a device verdict failed)"}
            return names.get(w[1], str(w[1])), w[2]

        def step(self):
            if not use_dist:
                return self.plan.run(self.stream)
            if shared_gpu:
                logits = self.plan.run(self.stream)
                h = logits.cpu()
                sum_across_ranks(h, None, Workload.exchange)
                logits.copy_(h)
                return logits
            return sharded_splat_forward(self.local, None, *self.full, exchange=Workload.exchange)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        return max_over_ranks(time.perf_counter() - t0)

    wl = Workload(args.config)
    si, P, N = wl.si, wl.P_total, wl.N
    pi, mi, radii, cov6 = wl.prep

    # ---- untimed: SETTLE_STEPS steps to bring the GPU out of its idle power state (the first ~10 ms after start-up run at
    # a lower clock: 47.7 us per step measured with 20 warm-up steps, 46.3 with 200 or 2000), then the W warm-up steps
    for _ in range(SETTLE_STEPS + args.warmup):
        wl.step()
    torch.cuda.synchronize()

    # ---- N > 1, still untimed: which exchange sums the partial grids fastest on THIS node (one RCCL all-reduce; chunks sent
    # straight to their owners over all xGMI links at once, summed there, all-gathered; RCCL's reduce-scatter + all-gather --
    # sharded.sum_across_ranks).  Ten steps each, max over ranks (so every rank sees the same numbers and takes the same one).
    exchange_tune = None
    if use_dist and world > 1:
        exchange_tune = {}
        for how in EXCHANGES:
            Workload.exchange = how
            try:
                exchange_tune[how] = timed(wl.step, 10, 3) / 10 * 1e3
            except Exception as exc:   # noqa: BLE001  (a collective this RCCL build refuses: refused on every rank alike; the all-reduce stays)
                if how == "all_reduce":
                    raise
                exchange_tune[how] = float("inf")
                print(f"exchange {how!r} not available: {type(exc).__name__}: {exc}", file=sys.stderr)
        Workload.exchange = min(exchange_tune, key=exchange_tune.get)
        exchange_tune = {k: (None if v == float("inf") else v) for k, v in exchange_tune.items()}
        for _ in range(args.warmup):
            wl.step()
        torch.cuda.synchronize()

    # ---- headline: exactly K steps, nothing else on the stream (no event records: a hipEvent pair costs ~3 us of
    # stream time, which at --steps 20 used to bracket every timed launch)
    # The loop is run LOOPS times (each: barrier + synchronize, exactly K steps, synchronize + barrier, max over ranks); the line
    # reports the MEDIAN loop and, in `timing`, every loop's time (a single 0.9 ms sample said nothing about spread).
    LOOPS = 5
    loop_s = []
    for _ in range(LOOPS):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        loop_s.append(max_over_ranks(time.perf_counter() - t0))
    elapsed = sorted(loop_s)[LOOPS // 2]

    # ---- the dominant kernel, timed live in a SEPARATE loop after the headline: hipEvents around every launch of
    # KERNEL_SAMPLES further steps of the same workload, on the stream the kernel is launched on
    _lib.check(lib.gf_profile_stride(1), "gf_profile_stride")
    _lib.check(lib.gf_profile_enable(KERNEL_SAMPLES), "gf_profile_enable")
    for _ in range(KERNEL_SAMPLES):
        wl.step()
    torch.cuda.synchronize()
    buf = (ctypes.c_float * KERNEL_SAMPLES)()
    n_ev = lib.gf_profile_read(buf, KERNEL_SAMPLES)
    lib.gf_profile_enable(0)
    kernel_ms = float(np.mean(buf[:n_ev])) if n_ev > 0 else None
    kernel_name, verdict_bits = wl.path()

    extras = {}

    def extra(name, fn):
        try:
            r = fn()
            if r is not None:
                extras[name] = r
        except Exception as exc:  # an extra must never cost the headline line
            extras[name] = {"error": f"{type(exc).__name__}: {exc}"}

    single = world == 1 and not use_dist
    if not args.no_extras and single:
        def two_stream():
            # two frames in flight: two pre-bound plans (own outputs / workspace) alternating on two HIP streams, so the
            # latency-bound prep kernel, the cold start and the tail of one step overlap the render kernel of the other
            plans = [wl.plan, SplatForwardPlan(wl.variant, *wl.tensors, si.H, si.W, si.D, flags=_lib.GF_PTS_AUTO)]
            streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
            for i in range(2 * max(2, args.warmup // 2)):
                plans[i % 2].run(streams[i % 2].cuda_stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                plans[i % 2].run(streams[i % 2].cuda_stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            return {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                    "note": "same K steps, two frames in flight on two HIP streams (double-buffered outputs and workspace)"}

        def hip_graph():
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                wl.plan.run()
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                wl.plan.run()
            for _ in range(max(2, args.warmup // 2)):
                graph.replay()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t3
            return {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                    "note": "same K steps as replays of one captured HIP graph (prep + render)"}

        def exact_kernel():
            # the exact-fp32 VALU tile kernel (GF_EXACT_FP32: the default before the matrix-core kernel was), same K steps
            if wl.variant != _lib.GF_SPLAT_BASE:
                return None
            plan = SplatForwardPlan(wl.variant, *wl.tensors, si.H, si.W, si.D, flags=_lib.GF_PTS_AUTO | _lib.GF_EXACT_FP32)
            ref_out = wl.plan.run(wl.stream).clone()
            torch.cuda.synchronize()   # (wl.stream is a raw handle: nothing orders torch's ops behind the launches on it but this)
            got = plan.run(wl.stream)
            torch.cuda.synchronize()
            err = float(((got - ref_out).abs() / got.abs().clamp(min=1.0)).max())
            err_abs = float((got - ref_out).abs().max())
            for _ in range(max(2, args.warmup // 2)):
                plan.run(wl.stream)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(args.steps):
                plan.run(wl.stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t4
            return {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                    "max_scaled_diff_of_the_default_kernel": err, "max_abs_diff_of_the_default_kernel": err_abs,
                    "max_abs_logit": float(got.abs().max()),
                    "note": "same K steps with GF_EXACT_FP32: the exact-fp32 VALU tile kernel (2e-6 from the reference)"}

        def frames():
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_frame
            out = {}
            for cfg, nf in (("nuscenes_gs25600_solid", 20), ("nuscenes_gs144000", 6)):
                out[cfg] = bench_frame.run(cfg, frames=nf, warmup=2, device=str(dev), graph=True)
            out["unit"] = "frames/s"
            out["note"] = ("one inference frame of the hot path per config (tools/bench_frame.py), single GPU, synthetic inputs in "
                           "HBM; frames_per_s = eager launches, frames_per_s_graph = the same frame replayed as one captured HIP graph")
            return out

        def gs144000_forward():
            # BASELINE config [3]: the same measurement for nuscenes_gs144000 (P = 144 000 into the same grid), so the
            # driver's line records it: K/4 plain steps for the step time, then bracketed launches for the render kernel
            w3 = Workload("nuscenes_gs144000")
            # (its own step count: at the driver's --steps 20 a quarter of K was ten steps, and the fixed cost of one
            # launch-to-synchronise bracket -- ~0.3 ms -- read as 7 us per step: 91 us where tools/fwd_time.py measures 84)
            steps = max(200, args.steps)
            for _ in range(50):
                w3.step()
            torch.cuda.synchronize()
            loops3 = []
            for _ in range(3):
                t5 = time.perf_counter()
                for _ in range(steps):
                    w3.step()
                torch.cuda.synchronize()
                loops3.append((time.perf_counter() - t5) / steps)
            dt = sorted(loops3)[1]
            _lib.check(lib.gf_profile_enable(KERNEL_SAMPLES), "gf_profile_enable")
            for _ in range(KERNEL_SAMPLES):
                w3.step()
            torch.cuda.synchronize()
            b3 = (ctypes.c_float * KERNEL_SAMPLES)()
            n3 = lib.gf_profile_read(b3, KERNEL_SAMPLES)
            lib.gf_profile_enable(0)
            k_ms = float(np.mean(b3[:n3])) if n3 > 0 else None
            name3, bits3 = w3.path()
            ab = algorithmic_bytes(w3.P_total, w3.N)
            traffic3, traffic_step3, note3 = committed_traffic("nuscenes_gs144000")
            r = {"config": f"nuscenes_gs144000: splat forward of ONE frame, P={w3.P_total} -> {w3.si.H}x{w3.si.W}x{w3.si.D}x18, bs=1",
                 "value": w3.P_total / dt, "unit": "Gaussians/s", "ms_per_step": dt * 1e3, "steps": steps, "loops": 3, "reported": "median loop",
                 "roofline": None if not k_ms else {
                     "bound": "hbm", "achieved": ab / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic3, "traffic_step": traffic_step3,
                     "kernel": name3, "verdict_bits": bits3, "kernel_us": k_ms * 1e3, "kernel_launches_timed": n3,
                     "algorithmic_bytes": ab, "op_achieved": ab / dt / 1e9, "op_frac": ab / dt / 1e9 / HBM_PEAK_GBS}}
            if note3:
                r["roofline"]["traffic_note"] = note3
            # ... and its backward (round 6: the matrix-core backward takes rows of up to 4 096 words), as the autograd module
            # calls the pair, with the Gaussian-major exact kernels beside it
            try:
                from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
                t3, s3 = w3.tensors, w3.si
                lg3, _, _, _, st3 = splat_forward(w3.variant, *t3, s3.H, s3.W, s3.D, flags=_lib.GF_PREPARE_BACKWARD)
                torch.cuda.synchronize()
                words = st3.view(torch.int32)[:5].tolist()
                fast = words[0] == 0 and words[1] in _lib.GF_PATHS_MATRIX_CORE and (words[4] & 1)
                g3 = torch.randn(lg3.shape, generator=torch.Generator().manual_seed(1)).to(dev)
                tb = {}
                for name, fl in (("matrix_core", (_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if fast else 0), ("exact_fp32", _lib.GF_EXACT_FP32)):
                    fn = lambda: splat_backward(w3.variant, *t3, s3.H, s3.W, s3.D, g3, state=st3, flags=fl)
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                    t9 = time.perf_counter()
                    for _ in range(steps):
                        fn()
                    torch.cuda.synchronize()
                    tb[name] = (time.perf_counter() - t9) / steps
                by3 = 128 * w3.P_total + 24 * w3.N + 72 * w3.N + 112 * w3.P_total
                r["backward"] = {"us_per_call": tb["matrix_core"] * 1e6, "exact_fp32_us_per_call": tb["exact_fp32"] * 1e6,
                                 "algorithmic_bytes": by3, "frac_of_8TBs": by3 / tb["matrix_core"] / 8e12,
                                 "forward_prepared_rows": bool(fast), "state_words": words}
            except Exception as e:  # noqa: BLE001  (an extra must not take the line down)
                r["backward"] = {"error": repr(e)}
            return r

        def verified_once():
            # the same K steps with the pts scan skipped (GF_PTS_ASSUME_DENSE: the grid was verified once, as
            # LocalAggregator.register_grid does for a tensor it is handed every frame); the range verdicts still run.
            # NEVER the headline: `value` above re-verifies the grid in every step.
            plan = SplatForwardPlan(wl.variant, *wl.tensors, si.H, si.W, si.D, flags=_lib.GF_PTS_ASSUME_DENSE)
            ref_out = wl.plan.run(wl.stream).clone()
            torch.cuda.synchronize()
            got_v = plan.run(wl.stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(got_v, ref_out))
            for _ in range(max(2, args.warmup // 2)):
                plan.run(wl.stream)
            torch.cuda.synchronize()
            t6 = time.perf_counter()
            for _ in range(args.steps):
                plan.run(wl.stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t6
            return {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                    "bit_identical_to_headline": same,
                    "note": "same K steps with GF_PTS_ASSUME_DENSE (pts / points_int not re-scanned; range verdicts kept)"}

        def splat_backward_extra():
            # SURVEY.md section 8 row (a), second half: the splat backward of the same frame through the C ABI, as the autograd
            # module calls it (forward with GF_PREPARE_BACKWARD, backward with GF_MFMA_SPLAT | GF_RECORDS_VALID when the forward ran
            # on the matrix cores; the Gaussian-major exact kernels beside it).  Algorithmic bytes: dL/dlogits once + parameters +
            # gradients (section 8d).
            if si.variant == "prob":
                return None
            from gaussianformer_amd.local_aggregate import splat_backward, splat_forward
            t = wl.tensors
            logits, _, _, _, state = splat_forward(wl.variant, *t, si.H, si.W, si.D, flags=_lib.GF_PREPARE_BACKWARD)
            torch.cuda.synchronize()
            words = state.view(torch.int32)[:5].tolist()
            fast = words[0] == 0 and words[1] in _lib.GF_PATHS_MATRIX_CORE and (words[4] & 1)
            g = torch.randn(logits.shape, generator=torch.Generator().manual_seed(1)).to(dev)
            res = {}
            for name, flags in (("matrix_core", (_lib.GF_MFMA_SPLAT | _lib.GF_RECORDS_VALID) if fast else 0), ("exact_fp32", _lib.GF_EXACT_FP32)):
                fn = lambda: splat_backward(wl.variant, *t, si.H, si.W, si.D, g, state=state, flags=flags)
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                t7 = time.perf_counter()
                for _ in range(args.steps):
                    out = fn()
                torch.cuda.synchronize()
                res[name] = (time.perf_counter() - t7) / args.steps
                if name == "matrix_core":
                    finite = all(bool(torch.isfinite(x).all()) for x in out)
            by = 128 * P + 24 * N + 72 * N + 112 * P   # SURVEY.md section 8d (the same figure tools/bench_ops.py uses)
            return {"us_per_call": res["matrix_core"] * 1e6, "exact_fp32_us_per_call": res["exact_fp32"] * 1e6,
                    "algorithmic_bytes": by, "frac_of_8TBs": by / res["matrix_core"] / 8e12, "finite": finite,
                    "forward_prepared_rows": bool(fast),
                    "note": "module-level calls (four output allocations included); matrix-core backward = gradient kernel + row sums"}

        def clustered_centres():
            # "distribution sensitivity nobody reports".  The same frame with the Gaussians' centres drawn from
            # sigmoid(N(0, 1)) -- clustered in the middle of the grid, as a trained model's are.  (The comparison with rounds 3-4's
            # contiguous bands of units lives in the development build: tools/interleave_probe.py, profiles/interleave_r05.txt.)
            sc = make_splat_inputs(args.config, seed=0, clustered=True)
            pi, mi, radii, cov6 = oracle.prepare_splat_inputs(sc.pts, sc.means3D, sc.scales, sc.cov3D, sc.pc_min, sc.grid_size,
                                                                sc.scale_multiplier, radii_min=1 if sc.variant == "prob" else None)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            tc = [up(sc.pts), up(pi), up(sc.means3D), up(mi), up(sc.opacities), up(sc.semantics), up(radii), up(cov6)]
            plan = SplatForwardPlan(wl.variant, *tc, sc.H, sc.W, sc.D, flags=_lib.GF_PTS_AUTO)
            for _ in range(max(2, args.warmup // 2)):
                plan.run(wl.stream)
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            for _ in range(args.steps):
                plan.run(wl.stream)
            torch.cuda.synchronize()
            dt8 = (time.perf_counter() - t8) / args.steps
            return {"ms_per_step": dt8 * 1e3, "value": sc.means3D.shape[0] / dt8, "unit": "Gaussians/s",
                    "note": "same workload with centres ~ sigmoid(N(0,1)); supertiles dealt to the XCDs round-robin"}

        extra("splat_backward", splat_backward_extra)
        extra("clustered_centres", clustered_centres)
        extra("two_stream", two_stream)
        extra("hip_graph", hip_graph)
        extra("verified_once", verified_once)
        extra("gs144000_forward", gs144000_forward)
        def train_step():
            # BASELINE config [2]: the native ops of one training step chained through autograd (tools/bench_step.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_step
            r = bench_step.run(anchors=25600, steps=5, warmup=2)
            r["unit"] = "ms per step"
            return r

        extra("exact_fp32_kernel", exact_kernel)
        extra("frames_per_s", frames)
        extra("train_step", train_step)

    if not args.no_extras and use_dist:
        def kernel_only():
            dt = timed(lambda: wl.plan.run(wl.stream), args.steps, max(2, args.warmup // 2))
            return {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                    "note": "the same shards, splat only (no collective), max over ranks"}

        def gs144000():
            w2 = Workload("nuscenes_gs144000")
            steps = max(10, args.steps // 4)
            both = timed(w2.step, steps, 3)
            alone = timed(lambda: w2.plan.run(w2.stream), steps, 3)
            return {"config": f"nuscenes_gs144000: P={w2.P_total} Gaussians sharded {world}-way ({w2.hi - w2.lo} on rank 0), "
                              f"all-reduce of the [{w2.N},18] fp32 logits",
                    "value": w2.P_total / (both / steps), "unit": "Gaussians/s", "ms_per_step": both / steps * 1e3,
                    "kernel_only_ms_per_step": alone / steps * 1e3, "collective_ms_per_step": (both - alone) / steps * 1e3, "steps": steps}

        def rs_labels():
            if si.variant == "prob" or N % world or shared_gpu:
                return None
            from gaussianformer_amd.head import occupancy_labels
            mine = torch.empty(N // world, 18, dtype=torch.float32, device=dev)
            labels = torch.empty(N, dtype=torch.int64, device=dev)

            def label_step():
                logits = wl.plan.run(wl.stream)
                dist.reduce_scatter_tensor(mine, logits, op=dist.ReduceOp.SUM)
                dist.all_gather_into_tensor(labels, occupancy_labels(mine))
            dt = timed(label_step, args.steps, max(2, args.warmup // 2))
            return {"value": P / (dt / args.steps), "unit": "Gaussians/s", "ms_per_step": dt / args.steps * 1e3,
                    "note": "same K steps ending in labels: reduce-scatter of the partial logits, labels on the owned slab, "
                            "all-gather of the labels"}

        def slab(config):
            # SPATIAL partition of the same frame (sharded.slab_bounds): rank r renders voxel rows [x0, x1) from ALL the
            # Gaussians (boxes clipped to the slab inside the op), no reduction; the slabs are all-gathered -- the fp32
            # logits (46 MB / world per rank) or, for inference that ends in labels, 8 bytes per voxel.
            from gaussianformer_amd.head import occupancy_labels
            from gaussianformer_amd.sharded import slab_bounds
            w2 = wl if config == args.config else Workload(config)
            s2 = w2.si
            if s2.variant == "prob":
                return None
            pi2, mi2, radii2, cov62 = w2.prep
            x0, x1 = slab_bounds(s2.H, rank, world)
            plane = s2.W * s2.D
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            shift = np.array([x0, 0, 0], dtype=pi2.dtype)
            plan = SplatForwardPlan(w2.variant, up(s2.pts[x0 * plane:x1 * plane]), up(pi2[x0 * plane:x1 * plane] - shift), up(s2.means3D),
                                    up(mi2 - shift), up(s2.opacities), up(s2.semantics), up(radii2), up(cov62), x1 - x0, s2.W, s2.D,
                                    flags=_lib.GF_PTS_AUTO)
            bounds = [slab_bounds(s2.H, r, world) for r in range(world)]
            tallest = max(b - a for a, b in bounds) * plane

            def gather(t):
                mine = t.new_zeros((tallest,) + tuple(t.shape[1:]))
                mine[:t.shape[0]] = t
                if shared_gpu:
                    host = torch.empty((world * tallest,) + tuple(t.shape[1:]), dtype=t.dtype)
                    dist.all_gather_into_tensor(host, mine.cpu())
                    buf = host.to(dev)
                else:
                    buf = t.new_empty((world * tallest,) + tuple(t.shape[1:]))
                    dist.all_gather_into_tensor(buf, mine)
                return torch.cat([buf[r * tallest:r * tallest + (b - a) * plane] for r, (a, b) in enumerate(bounds)], dim=0)
            steps = args.steps if config == args.config else max(10, args.steps // 4)
            warm = max(2, args.warmup // 2)
            t_local = timed(lambda: plan.run(w2.stream), steps, warm)
            t_logits = timed(lambda: gather(plan.run(w2.stream)), steps, warm)
            t_labels = timed(lambda: gather(occupancy_labels(plan.run(w2.stream))), steps, warm)
            out = {"config": f"{config}: voxel rows {x0}..{x1} of {s2.H} on rank 0, all {w2.P_total} Gaussians passed to every rank",
                   "unit": "Gaussians/s", "steps": steps,
                   "splat_only": {"value": w2.P_total / (t_local / steps), "ms_per_step": t_local / steps * 1e3},
                   "all_gather_logits": {"value": w2.P_total / (t_logits / steps), "ms_per_step": t_logits / steps * 1e3},
                   "all_gather_labels": {"value": w2.P_total / (t_labels / steps), "ms_per_step": t_labels / steps * 1e3},
                   "note": "no reduction: every voxel is computed by one rank from the same Gaussians in the same order as on one GPU"}
            if os.environ.get("GF_BENCH_CHECK") == "1":
                full_plan = SplatForwardPlan(w2.variant, up(s2.pts), up(pi2), up(s2.means3D), up(mi2), up(s2.opacities), up(s2.semantics),
                                             up(radii2), up(cov62), s2.H, s2.W, s2.D, flags=_lib.GF_PTS_AUTO)
                want = full_plan.run().clone()
                got = gather(plan.run(w2.stream))
                torch.cuda.synchronize()
                out["check_bit_identical_to_single_device"] = bool(torch.equal(got.view(torch.int32), want.view(torch.int32)))
                assert out["check_bit_identical_to_single_device"]
            return out

        def frame_sharded():
            # one END-TO-END inference frame with the anchors split over the ranks: the configuration in which
            # N GPUs can win -- the per-anchor encoder work shards without a collective -- next to the 1-GPU `frames_per_s`
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_frame

            def gather(t, dim):
                t = t.contiguous()
                if world == 1:
                    return t
                parts = [torch.empty_like(t) for _ in range(world)]
                if shared_gpu:
                    hp = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
                    dist.all_gather(hp, t.cpu())
                    parts = [h.to(dev) for h in hp]
                else:
                    dist.all_gather(parts, t)
                return torch.cat(parts, dim=dim)

            def reduce_sum(t):
                if world > 1:
                    all_reduce(t)
                return t
            out = {}
            for cfg, nf in (("nuscenes_gs25600_solid", 10), ("nuscenes_gs144000", 4)):
                out[cfg] = bench_frame.run_sharded(cfg, rank, world, gather, reduce_sum, barrier, max_over_ranks, frames=nf, warmup=2,
                                                   device=str(dev), check=os.environ.get("GF_BENCH_CHECK") == "1")
            out["unit"] = "ms per frame (slowest rank)"
            return out

        extra("kernel_only", kernel_only)
        extra("frame_sharded", frame_sharded)
        extra("gs144000", gs144000)
        extra("reduce_scatter_labels", rs_labels)
        extra("slab_partition", lambda: slab(args.config))
        extra("slab_partition_gs144000", lambda: slab("nuscenes_gs144000"))
        if os.environ.get("GF_BENCH_CHECK") == "1":
            # the sharded sum against the single-device result of the whole set (tests)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            full_plan = SplatForwardPlan(wl.variant, up(si.pts), up(pi), up(si.means3D), up(mi), up(si.opacities), up(si.semantics),
                                         up(radii), up(cov6), si.H, si.W, si.D, flags=_lib.GF_PTS_AUTO)
            want = full_plan.run().clone()
            got = wl.step()
            torch.cuda.synchronize()
            err = float(((got - want).abs() / want.abs().clamp(min=1.0)).max())
            extras["check_max_scaled_err_vs_single_device"] = err
            assert err <= 1e-4, err

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P / (elapsed / args.steps)
        P_launch = wl.hi - wl.lo
        abytes = algorithmic_bytes(P_launch, N)
        roofline = None
        if kernel_ms:
            achieved = abytes / (kernel_ms * 1e-3) / 1e9
            traffic, traffic_step, traffic_note = committed_traffic(args.config) if single else (None, None, None)
            op_achieved = abytes / (ms_per_step * 1e-3) / 1e9
            roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                        "kernel": kernel_name, "verdict_bits": verdict_bits, "kernel_us": kernel_ms * 1e3,
                        "kernel_launches_timed": n_ev,
                        "kernel_timing": f"hipEvents around each of {n_ev} launches of a separate loop run after the "
                                         "headline's K steps (the headline loop records no events)",
                        "algorithmic_bytes": abytes,
                        # the whole op (prep + render launches) against the same roofline, by the headline's own clock
                        "op_achieved": op_achieved, "op_frac": op_achieved / HBM_PEAK_GBS,
                        "traffic_step": traffic_step}
            if traffic_note:
                roofline["traffic_note"] = traffic_note
        out = {
            "metric": headline_metric(),
            "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "timing": {"loops": LOOPS, "reported": "median loop", "ms_per_step_of_each_loop": [t / args.steps * 1e3 for t in loop_s],
                       "min": min(loop_s) / args.steps * 1e3, "max": max(loop_s) / args.steps * 1e3},
            "dtype": "f32",
            # what the arithmetic of the kernel that ran is (inputs, outputs and accumulators are fp32 in every case)
            "arithmetic": ("split-f16 operands (hi + lo) on v_mfma_f32_32x32x16_f16, fp32 accumulate; exponent coefficients "
                           "formed in fp64" if "mfma" in kernel_name else "fp32 VALU"),
            "data": "synthetic",
            "config": {"workload": f"{args.config}: splat forward of ONE frame, P={P} Gaussians -> {si.H}x{si.W}x{si.D}x18 "
                                   f"grid (N={N} voxel-centre points), bs=1",
                       "P_total": P, "P_per_gpu": P_launch, "N": N, "pts_layout": "auto-detected dense grid",
                       "parallelism": "single GPU" if not use_dist else
                                      f"the frame's Gaussians in {world} contiguous shards + the sum of the partial logits "
                                      f"({Workload.exchange}; {'gloo via host copies, shared GPU (test mode)' if shared_gpu else 'RCCL'})"},
            "roofline": roofline,
        }
        out.update(extras)
        if exchange_tune:
            out["exchange"] = Workload.exchange
            out["exchange_autotune_ms_per_step"] = exchange_tune   # ten untimed steps each in the warm-up, max over ranks
        if use_dist:
            # the numbers a scaling curve is made of, as flat top-level keys of the line (the nested extras keep the
            # details): splat kernels alone, the collective's share of a step, and the anchor-sharded END-TO-END frame
            ko, fs = extras.get("kernel_only"), extras.get("frame_sharded")
            if isinstance(ko, dict) and "ms_per_step" in ko:
                out["kernel_only_ms_per_step"] = ko["ms_per_step"]
                out["collective_ms_per_step"] = max(ms_per_step - ko["ms_per_step"], 0.0)
                out["kernel_only_value"] = ko["value"]
            if isinstance(fs, dict):
                for cfg, key in (("nuscenes_gs25600_solid", "gs25600"), ("nuscenes_gs144000", "gs144000")):
                    r = fs.get(cfg)
                    if isinstance(r, dict):
                        for splat in ("slab", "allreduce"):
                            if isinstance(r.get(splat), dict):
                                out[f"frames_per_s_sharded_{key}_{splat}"] = r[splat]["frames_per_s"]
        if single and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(si, pi, mi, radii, cov6)
            try:
                out["cpu_baseline_torch"] = cpu_baseline_torch(si, pi, mi, radii, cov6)
            except Exception as exc:
                out["cpu_baseline_torch"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
