"""Two ranks on the GPU.  The 1-GPU box cannot run RCCL between two devices, but it can run the PRODUCT's data-parallel
path with two processes sharing the one GPU and torch.distributed's gloo backend carrying the collective on the
device tensor (gloo stages CUDA tensors through the host, stream-ordered): each rank owns the passes v % 2 == rank,
gs_trainer_step runs accumulate -> all-reduce hook (dist.TorchAllReduce, the same hook bench.py installs with the nccl
backend) -> apply, three steps of the reference's update rule.  Checked: both ranks end with bit-identical replicas, and the replicas equal a
single-process run of all passes up to the re-association of the pass sum (1e-4 relative on the parameters' change)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, M, N_CAMS, W, H, STEPS = 3000, 4, 4, 160, 96, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    import gsplat_amd as gs
    s = gs.synth.random_splats(P, M, 2024)
    cams = gs.camera.get_cameras(N_CAMS)
    rng = np.random.default_rng(9)
    fw = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) for _ in range(N_CAMS)]   # truth content is irrelevant here
    fb = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) for _ in range(N_CAMS)]
    return s, cams, fw, fb


def _train(rank, world, hook_factory, adam_densify=False):
    import gsplat_amd as gs
    from gsplat_amd import capi
    s, cams, fw, fb = _scene()
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = s["D"]
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb)
    tr.shard(rank, world)
    hook = hook_factory(tr) if hook_factory else None
    proj = gs.Project()     # the reference's clamped ascent: linear in the gradient, so re-association of the pass sum stays tiny
    if adam_densify:
        proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM, lrLocation=1e-3, lrSh=2e-3, lrScale=5e-4, lrOpacity=1e-3, lrRotation=1e-3,
                          paramDensifyVariance=0.05, paramCullOpacity=0.15, paramSplitSize=0.06)
    for k in range(STEPS + (2 if adam_densify else 0)):
        tr.train(proj, densify=adam_densify and k == 2)          # no stats: the steps run ahead of the device, as in bench.py
    h = gs.ModelSplatsHost.fromDevice(tr.model)
    n = h.count
    out = np.concatenate([h.locations[:3 * n], h.shs[:3 * M * n], h.scales[:3 * n], h.opacities[:n], h.rotations[:4 * n]])
    tr.close()
    del hook
    return out, np.concatenate([s["loc"].reshape(-1), s["sh"].reshape(-1), s["scale"].reshape(-1), s["opac"].reshape(-1), s["rot"].reshape(-1)])


def _worker(rank, world, port, q, sharded=False, adam_densify=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsplat_amd import dist as gsdist
    hooks = []
    out, _ = _train(rank, world, lambda tr: hooks.append(gsdist.TorchShardedUpdate(tr, rank, world) if sharded else gsdist.TorchAllReduce(tr)) or hooks[-1],
                    adam_densify)
    if sharded and not adam_densify:
        assert hooks[0].calls == {"reduce_scatter": STEPS, "all_gather": STEPS}, hooks[0].calls
    if sharded and adam_densify:   # the densify step also gathers the gradient buffer and both Adam moments
        assert hooks[0].calls == {"reduce_scatter": STEPS + 2, "all_gather": STEPS + 2 + 3}, hooks[0].calls
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sharded", [False, True])
def test_two_ranks_on_one_gpu_match_the_single_process_run(sharded):
    """sharded: reduce-scatter -> each rank updates its half of the plane-major parameter buffer -> all-gather
    (gs_trainer_set_sharded_update) instead of all-reduce + replicated update."""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sharded)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single, start = _train(0, 1, None)
    a, b = res[0][1], res[1][1]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))            # replicas stay bit-identical
    moved = np.abs(single - start).max()
    assert moved > 1e-7                                                     # the steps did something
    assert np.abs(a - single).max() <= 1e-4 * moved, (np.abs(a - single).max(), moved)


def _run_two(sharded, adam_densify):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sharded, adam_densify)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res[0][1], res[1][1]


def test_sharded_update_with_adam_and_densify_equals_the_allreduce_form():
    """Adam moments exist per chunk only under the sharded update, and a densify step has to complete gradients and
    moments on every rank before it re-indexes them.  With the same sums on the wire (the gloo stand-in reduces both forms
    identically) the sharded form must leave bit for bit the model the all-reduce form leaves — through five Adam steps
    with a split / clone / prune in the middle — and both ranks must hold it."""
    sys.path.insert(0, ROOT)
    a0, a1 = _run_two(False, True)
    s0, s1 = _run_two(True, True)
    assert a0.size != (11 + 3 * M) * P            # densify changed the splat count
    assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32)) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
    assert s0.size == a0.size and np.array_equal(s0.view(np.uint32), a0.view(np.uint32))


def _grad_planes(tr, n_splats):
    import ctypes as C
    from gsplat_amd import capi
    ptr, n = tr.grad_buffer()
    tr.synchronize()
    buf = np.empty(n, np.float32)
    capi.check(capi.lib().gs_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4))
    planes = 12 + 3 * M
    return buf.reshape(planes, n // planes)[:, :n_splats].copy()


def _compact_run(rank, world, hook_factory, overlap=1):
    """One gradients-only step in the fused form, one in the per-pass form (what a densify step runs), then three training
    steps of Adam, the third with densify: (gradient planes fused, gradient planes per pass, final model)."""
    import gsplat_amd as gs
    from gsplat_amd import capi
    s, cams, fw, fb = _scene()
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = s["D"]
    host.capacity = P + 500
    tr = gs.Trainer(W, H)
    tr.set_option("exchange_overlap", overlap)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb)
    tr.shard(rank, world)
    hook = hook_factory(tr, cams) if hook_factory else None
    still = gs.Project(lrLocation=0.0, lrSh=0.0, lrScale=0.0, lrOpacity=0.0, lrRotation=0.0)
    tr.train(still)
    g_fused = _grad_planes(tr, P)
    tr.set_option("fuse_camera_passes", 0)
    tr.train(still)
    g_pass = _grad_planes(tr, P)
    tr.set_option("fuse_camera_passes", 1)
    proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM, lrLocation=1e-3, lrSh=2e-3, lrScale=5e-4, lrOpacity=1e-3, lrRotation=1e-3,
                      paramDensifyVariance=0.05, paramCullOpacity=0.15, paramSplitSize=0.06)
    for k in range(4):
        tr.train(proj, densify=(k == 2))
    h = gs.ModelSplatsHost.fromDevice(tr.model)
    n = h.count
    model = np.concatenate([h.locations[:3 * n], h.shs[:3 * M * n], h.scales[:3 * n], h.opacities[:n], h.rotations[:4 * n]])
    calls = dict(hook.calls) if hook is not None else None
    tr.close()
    del hook
    return g_fused, g_pass, model, calls


def _compact_worker(rank, world, port, q, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsplat_amd import dist as gsdist
    # a group of its own for the all-reduce, as bench.py gives it under nccl: the two collectives are in flight side by side
    reduce_group = dist.new_group(backend="gloo")
    out = _compact_run(rank, world, lambda tr, cams: gsdist.TorchCompactExchange(tr, rank, world, cams, reduce_group), overlap)
    q.put((rank,) + out)
    dist.barrier()
    dist.destroy_process_group()


def test_compact_exchange_two_ranks_on_one_gpu():
    """gs_trainer_set_compact_exchange with two processes on the GPU (gloo carries the two collectives): all-gather of the
    cameras' dL_dRGB records + all-reduce of the twelve non-SH planes on the trainer's second stream, SH planes rebuilt on
    every rank.  Checked against a single-process run of all four cameras:
      * the SH gradient planes are the single-GPU step's BIT FOR BIT, in the fused form and in the per-pass form (a densify
        step's) — no collective sums them;
      * the twelve other planes are the same sums re-associated by the all-reduce (<= 2e-6 of the plane's scale), `var` zero
        in the fused form and present in the per-pass form;
      * the replicas are bit-identical after Adam steps with a densify step in the middle, with the all-reduce overlapped
        (second stream + events) and with the two collectives issued one after the other — and both orders give the same bits,
        which a missing dependency between the streams would not."""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    ctx = mp.get_context("spawn")
    results = {}
    for overlap in (1, 0):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_compact_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in procs], key=lambda x: x[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        results[overlap] = res
    sf, sp, smodel, _ = _compact_run(0, 1, None)
    sh = slice(3, 3 + 3 * M)
    geo = [0, 1, 2] + list(range(3 + 3 * M, 12 + 3 * M))
    for overlap, res in results.items():
        (_, f0, p0, m0, c0), (_, f1, p1, m1, c1) = res
        assert c0 == c1 == {"all_gather": 6, "all_reduce": 6}, (c0, c1)
        for got_f, got_p in ((f0, p0), (f1, p1)):
            assert np.array_equal(got_f[sh].view(np.uint32), sf[sh].view(np.uint32))   # SH gradients: the single-GPU bits
            assert np.array_equal(got_p[sh].view(np.uint32), sp[sh].view(np.uint32))
            for got, want in ((got_f, sf), (got_p, sp)):
                for pl in geo:
                    assert np.abs(got[pl] - want[pl]).max() <= 2e-6 * np.abs(want[pl]).max() + 1e-30, pl
            assert not got_f[-1].any() and got_p[-1].any() and np.abs(sf[sh]).max() > 0
        assert np.array_equal(f0.view(np.uint32), f1.view(np.uint32)) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
        assert m0.size == m1.size and np.array_equal(m0.view(np.uint32), m1.view(np.uint32))            # replicas bit-identical
        assert m0.size != (11 + 3 * M) * P                                                                   # densify changed the count
    for k in (1, 2, 3):   # overlapped and serialised exchange: the same bits
        assert np.array_equal(results[1][0][k].view(np.uint32), results[0][0][k].view(np.uint32))
    moved = np.abs(smodel[:3 * 100] - results[1][0][3][:3 * 100]).max()
    assert moved < 1e-2   # and the run stays next to the single-process one (Adam amplifies the re-associated sums: no bit claim here)


@pytest.mark.parametrize("extra", [[], ["--collective", "torch-compact", "--exchange-overlap", "1"], ["--collective", "torch-sharded"]])
def test_bench_two_rank_rehearsal_on_one_gpu(extra):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), rehearsed on the one GPU with
    the gloo backend (RCCL refuses two ranks on one device: tools/try_nccl_two_ranks.py): camera sharding, the exchange chosen
    by --collective auto (cfg2 has M = 1: the all-reduce; the explicit runs take the compact exchange and the sharded update),
    the check of the data-parallel step against an unsharded step on rank 0, pre-warm loop, timed region, replica digests."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "5", "--warmup", "2",
           "--dist-backend", "gloo", "--device", "0", "--prewarm-seconds", "0.3", "--long-steps", "20", "--no-cpu-baseline"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["views_per_gpu"] == 4 and cfg["replicas_identical_after_run"] is True
    chk = cfg["exchange_checked_against_unsharded_step"]
    assert chk["ok"] and chk["max_plane_deviation"] <= 2e-5 and "after_fallback" not in chk, chk
    want = {0: "torch all-reduce", 4: "torch all-gather of 4 dL_dRGB records", 2: "torch reduce-scatter"}[len(extra)]
    assert cfg["collective"].startswith(want), cfg["collective"]
    if len(extra) == 4:
        assert chk["sh_planes_bit_identical_to_unsharded_step"] is True
    assert d["cold_start"]["value"] > 0 and d["prewarm"]["steps"] >= 20 and d["value"] > 0
    wb = cfg["wire_bytes_received_per_rank_per_step"]
    assert wb["allreduce"] == (12 + 3) * 10000 * 4 and wb["compact"] == 2 * 3 * 10000 * 4 + 12 * 10000 * 4


def test_eight_way_view_sharding_sums_to_the_single_shard_gradients():
    """shard the 16 passes of a step over 8 ranks (2 passes = one camera each, as bench.py --gpus 8
    does), run every shard's accumulate on the one GPU in turn, sum the eight averaged-gradient buffers on the host
    as the all-reduce would, and compare with the unsharded accumulate."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import gsplat_amd as gs
    from gsplat_amd import capi
    P8, M8, cams8, W8, H8 = 4000, 9, 8, 176, 128
    s = gs.synth.random_splats(P8, M8, 808)
    cams = gs.camera.get_cameras(cams8)
    rng = np.random.default_rng(3)
    fw = [rng.integers(0, 2 ** 32, W8 * H8, dtype=np.uint32) for _ in range(cams8)]
    fb = [rng.integers(0, 2 ** 32, W8 * H8, dtype=np.uint32) for _ in range(cams8)]

    def grads(rank, world):
        host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
        host.shDegree = s["D"]
        tr = gs.Trainer(W8, H8)
        tr.model = gs.ModelSplatsDevice(host)
        tr.captureTruths(cams, fw, fb)
        tr.shard(rank, world)
        st = tr.accumulate(stats=True)
        assert st.views == 2 * cams8 // world
        ptr, n = tr.grad_buffer()
        buf = np.empty(n, np.float32)
        capi.check(capi.lib().gs_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4))
        tr.close()
        planes = 12 + 3 * M8   # the padding lanes (plane stride is P rounded up to 64) are never read by anyone
        return buf.reshape(planes, n // planes)[:, :P8].astype(np.float64), st.num_rendered

    whole, R = grads(0, 1)
    parts = [grads(r, 8) for r in range(8)]
    assert sum(p[1] for p in parts) == R
    summed = np.sum([p[0] for p in parts], axis=0)
    scale = np.abs(whole).max()
    assert scale > 0 and np.abs(summed - whole).max() <= 2e-6 * scale


def test_rank_without_passes_steps_with_zero_gradient():
    """More ranks than passes (world 3, one camera): rank 2 owns nothing.  Its step must not fail with "no truth data":
    it contributes a zero gradient buffer, runs the collective hook and applies the update (a no-op on in-range
    parameters under the reference rule)."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    import gsplat_amd as gs
    from gsplat_amd import capi
    s = gs.synth.random_splats(500, 4, 7)
    cams = gs.camera.get_cameras(1)
    rng = np.random.default_rng(1)
    fw = [rng.integers(0, 2 ** 32, 64 * 48, dtype=np.uint32)]
    fb = [rng.integers(0, 2 ** 32, 64 * 48, dtype=np.uint32)]
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = s["D"]
    tr = gs.Trainer(64, 48)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb)
    tr.shard(2, 3)
    calls = []
    hook = capi.ALLREDUCE_FN(lambda buf, n, stream, user: calls.append(n) or 0)
    capi.check(capi.lib().gs_trainer_set_allreduce(tr.handle, C.cast(hook, C.c_void_p), None))
    st = tr.train(gs.Project(), stats=True)
    assert tr.local_views == [] and st.views == 0 and st.num_rendered == 0 and len(calls) == 1
    ptr, n = tr.grad_buffer()
    buf = np.empty(n, np.float32)
    capi.check(capi.lib().gs_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4))
    assert not buf.any()
    h = gs.ModelSplatsHost.fromDevice(tr.model)
    assert np.array_equal(h.locations[:1500], s["loc"]) and np.array_equal(h.opacities[:500], s["opac"])
    tr.close()
    # and the reference's own error is still there when there is no truth data at all
    tr2 = gs.Trainer(64, 48)
    with pytest.raises(RuntimeError, match="no truth data"):
        tr2.train(gs.Project())
    tr2.close()
