"""Multi-process data-parallel path on CPU (gloo, world_size 2): the passes of one iteration are sharded
round-robin, every rank accumulates its own passes with the GLOBAL divisor S, one sum all-reduce of the
gradient buffer follows, and every rank applies the identical update.  The compute
on each rank is the oracle (this box has no GPU); what is under test is the sharding + collective logic
of gaussian-splatterer_amd/dist.py that bench.py uses with the nccl (RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="2")
    import torch.distributed as dist

    import gsplat_amd as gs
    from oracle import pyoracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, M, W, H, n_cams = 300, 4, 64, 48, 3
    s = gs.synth.random_splats(P, M, 11)
    views = gs.camera.train_views(gs.camera.get_cameras(n_cams), W, H)
    V = views.shape[0]
    truths = np.random.default_rng(5).integers(0, 2 ** 32, (V, W * H), dtype=np.uint32)
    mine = gs.dist.shard_views(V, rank, world)
    o = orc.train_views(P, 1, M, W, H, s["loc"], s["sh"], s["scale"], s["opac"], s["rot"], views[mine], truths[mine], float(V))
    buf = np.concatenate([o[k] for k in ("loc", "sh", "scale", "opac", "rot", "var")])  # the library's plane order
    gs.dist.allreduce_sum_numpy(buf)
    p = {k: s[k].copy() for k in ("loc", "sh", "scale", "opac", "rot")}
    sizes = [3 * P, 3 * M * P, 3 * P, P, 4 * P, P]
    parts = np.split(buf, np.cumsum(sizes)[:-1])
    g = dict(zip(("loc", "sh", "scale", "opac", "rot", "var"), parts))
    orc.apply_sgd(p["loc"], p["sh"], p["scale"], p["opac"], p["rot"], g, (5e-5, 1e-4, 2e-5, 1e-4, 2.5e-5), 0.3, M)
    q.put((rank, mine, buf, np.concatenate([p[k] for k in ("loc", "sh", "scale", "opac", "rot")])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_sharding_matches_single_process():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, buf0, par0), (r1, mine1, buf1, par1) = res
    assert sorted(mine0 + mine1) == list(range(6)) and not set(mine0) & set(mine1)
    # replicas stay bit-identical: same reduced gradients, same updated parameters on both ranks
    assert np.array_equal(buf0.view(np.uint32), buf1.view(np.uint32))
    assert np.array_equal(par0.view(np.uint32), par1.view(np.uint32))
    # and equal (to fp32 summation-order accuracy) to the single-process iteration
    sys.path.insert(0, ROOT)
    import gsplat_amd as gs
    from oracle import pyoracle as orc
    P, M, W, H, n_cams = 300, 4, 64, 48, 3
    s = gs.synth.random_splats(P, M, 11)
    views = gs.camera.train_views(gs.camera.get_cameras(n_cams), W, H)
    truths = np.random.default_rng(5).integers(0, 2 ** 32, (6, W * H), dtype=np.uint32)
    o = orc.train_views(P, 1, M, W, H, s["loc"], s["sh"], s["scale"], s["opac"], s["rot"], views, truths, 6.0)
    single = np.concatenate([o[k] for k in ("loc", "sh", "scale", "opac", "rot", "var")])
    scale = np.abs(single).max()
    assert np.abs(buf0 - single).max() <= 1e-6 * scale


def test_shard_views_partition():
    sys.path.insert(0, ROOT)
    import gsplat_amd as gs
    for total in (1, 2, 16, 32, 33):
        for world in (1, 2, 4, 8):
            got = sorted(v for r in range(world) for v in gs.dist.shard_views(total, r, world))
            assert got == list(range(total))
            sizes = [len(gs.dist.shard_views(total, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
