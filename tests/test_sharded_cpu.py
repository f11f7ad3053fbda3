"""Multi-rank logic of the Gaussian-sharded splat forward on CPU (gloo, world_size 2).

The product's local op needs a GPU, so the per-rank splat is stubbed with the CPU oracle here
(the checker standing in for the kernel); what is under test is the sharding, the
all-reduce and that the N-way result equals the single-rank result."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from gaussianformer_amd.sharded import shard_bounds, sharded_splat_forward, sharded_splat_forward_prob
from gaussianformer_amd.synthetic import make_splat_inputs


def _local_splat_factory(si):
    def local_splat(pts, means3D, opacities, semantics, scales, cov3D):
        pi, mi, radii, cov6 = oracle.prepare_splat_inputs(pts[0].numpy(), means3D[0].numpy(), scales[0].numpy(),
                                                          cov3D[0].numpy(), si.pc_min, si.grid_size, si.scale_multiplier)
        out = oracle.splat_forward("base", pts[0].numpy(), pi, means3D[0].numpy(), mi, opacities[0].numpy(),
                                   semantics[0].numpy(), radii, cov6, si.H, si.W, si.D, nthreads=1)
        return torch.from_numpy(out["logits"])
    return local_splat


def _inputs():
    si = make_splat_inputs("nuscenes_gs25600_solid", seed=31, P=301, H=16, W=12, D=8)
    t = lambda a: torch.from_numpy(a)[None]
    return si, (t(si.pts), t(si.means3D), t(si.opacities), t(si.semantics), t(si.scales), t(si.cov3D))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    si, args = _inputs()
    out = sharded_splat_forward(_local_splat_factory(si), *args)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_partition():
    for P in (0, 1, 7, 25601, 144000):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(P, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == P
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_sharded_forward_matches_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    si, args = _inputs()
    single = _local_splat_factory(si)(*args).numpy()
    assert np.array_equal(results[0], results[1])          # every rank holds the full grid
    assert np.abs(results[0] - single).max() <= 1e-5 * max(1.0, np.abs(single).max())


# ---- prob variant: numerator / probability / density add up, 1 - bin multiplies, normalisation afterwards

def _prob_inputs():
    si = make_splat_inputs("prob_gs6400", seed=32, P=157, H=12, W=10, D=6)
    t = lambda a: torch.from_numpy(a)[None]
    return si, (t(si.pts), t(si.means3D), t(si.opacities), t(si.semantics), t(si.scales), t(si.cov3D))


def _prob_full(si, pts, means3D, opacities, semantics, scales, cov3D):
    pi, mi, radii, cov6 = oracle.prepare_splat_inputs(pts[0].numpy(), means3D[0].numpy(), scales[0].numpy(),
                                                      cov3D[0].numpy(), si.pc_min, si.grid_size, si.scale_multiplier, radii_min=1)
    return oracle.splat_forward("prob", pts[0].numpy(), pi, means3D[0].numpy(), mi, opacities[0].numpy(),
                                semantics[0].numpy(), radii, cov6, si.H, si.W, si.D, nthreads=1)


def _local_pieces_factory(si):
    def pieces(*args):
        out = _prob_full(si, *args)
        psum = out["probability"]
        # the oracle normalises like the reference; undo it where it divided (elsewhere the shard's numerator is < 1e-9)
        num = np.where(psum[:, None].astype(np.float64) > 1e-9, out["logits"] * psum[:, None], 0.0).astype(np.float32)
        return tuple(torch.from_numpy(a) for a in (num, out["bin_logits"], out["density"], psum))
    return pieces


def _prob_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    si, args = _prob_inputs()
    logits, bin_logits, density = sharded_splat_forward_prob(_local_pieces_factory(si), *args)
    q.put((rank, logits.numpy(), bin_logits.numpy(), density.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_prob_forward_matches_single_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prob_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    si, args = _prob_inputs()
    single = _prob_full(si, *args)
    for a, b in zip(results[0], results[1]):
        assert np.array_equal(a, b)                        # every rank ends with the same grids
    logits, bin_logits, density = results[0]
    assert np.abs(logits - single["logits"]).max() <= 1e-5
    assert np.abs(bin_logits - single["bin_logits"]).max() <= 1e-6
    assert np.abs(density - single["density"]).max() <= 1e-5 * max(1.0, np.abs(single["density"]).max())
    # a world of one is the plain op
    one = sharded_splat_forward_prob(_local_pieces_factory(si), *args)
    assert np.abs(one[0].numpy() - single["logits"]).max() <= 1e-6
