#!/usr/bin/env python3
"""bench.py — train steps/s of the Gaussian-splat training step on MI355X (BASELINE.json metric).

One "step" = Trainer::train (src/Trainer.cu:252-543): every pass of the iteration (forward raster,
loss, backward, gradient averaging), the gradient all-reduce when views are sharded, and the Adam
update.  Inputs are synthetic (PCG32 random-init splats, Fibonacci-sphere cameras,
truth = quantised render of a second splat set) and resident in HBM before the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, parameters replicated, the cameras (both passes of each) sharded over the
ranks (strong scaling), one RCCL sum all-reduce of the averaged-gradient buffer per step
(or, --collective *-sharded, reduce-scatter -> update of the rank's chunk -> all-gather).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling
N_SIMD = 256 * 4       # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4        # max shader clock; a wave64 VALU instruction occupies its SIMD-32 for 2 cycles

WORKLOAD_TEXT = {
    1: "cfg1: 1k random-init splats, 1 camera x (white,black) @256x256, SH degree 1 (M=4)",
    2: "cfg2: 10k random-init splats, 8 views (4 cameras x white/black) @512x512, SH degree 0 (M=1)",
    3: "cfg3: 100k random-init splats, 16 views (8 cameras x white/black) @1024x1024, SH degree 3 (M=16)",
    4: "cfg4: 100k random-init splats, 32 views (16 cameras x white/black) @1024x1024, SH degree 3 (M=16)",
    5: "cfg5: 1M random-init splats, 64 views (32 cameras x white/black) @2048x2048, SH degree 3 (M=16)",
}


def stage_bytes(stage, P, M, N, R, V):
    """ALGORITHMIC bytes of one launch of `stage` over V views (SURVEY.md §8d per-unit figures; R = mean
    num_rendered per view, N = pixels per view).  See DESIGN.md §Roofline for the derivation."""
    per_view = {
        "preprocess": (44 + 12 * M) * P + 75 * P,       # fwd param read + geometry-state write
        "scan": 8 * P,
        "scatter": 20 * P + 12 * R,                      # duplicate-stage read + key/value write
        "tile_sort": 24 * R + 8 * R,                     # one read+write of the 12-byte pair + range detection
        "render_forward": 40 * R + 20 * N,               # list read + (rgb, final_T, n_contrib) write
        "render_backward": 40 * R + 24 * N + 44 * P,     # list read + truth/colour/T/n_contrib read + intermediates
        "splat_backward": ((44 + 12 * M) + 44 + 31) * P,  # per-splat bwd reads
    }
    if stage in per_view:
        b = per_view[stage] * V
        if stage == "splat_backward":
            b += 2 * (48 + 12 * M) * P * V               # avg-grad + var accumulation (SURVEY counts an RMW per view)
        return b
    if stage == "update":
        return 7 * (44 + 12 * M) * P                     # Adam
    return 0


def stage_form_bytes(stage, P, M, N, R, V, G):
    """Compulsory bytes of the form that actually RAN: V passes in G camera groups.  Passes of one camera share projection,
    lists, forward blend (final_T, n_contrib) and, on a step without densify, the backward's list walk and gradient rows; the
    averaged-gradient planes are written once per step (SURVEY 8d counts a read-modify-write per view)."""
    form = {
        "preprocess": G * ((44 + 12 * M) + 75) * P,
        "scan": G * 8 * P,
        "scatter": G * (20 * P + 12 * R),
        "tile_sort": G * 32 * R,
        "render_forward": G * (40 * R + 8 * N) + V * 12 * N,
        "render_backward": G * (40 * R + 44 * P) + V * 24 * N,
        "splat_backward": G * ((44 + 12 * M) + 44 + 31) * P + (48 + 12 * M) * P,
        "update": 7 * (44 + 12 * M) * P,
    }
    return form.get(stage, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config index (default 3 = the metric's config)")
    ap.add_argument("--update", choices=["adam", "sgd"], default="adam")
    ap.add_argument("--collective", choices=["auto", "torch", "rccl", "torch-sharded", "rccl-sharded", "torch-compact", "rccl-compact"], default="auto",
                    help="torch / rccl: one sum all-reduce of the gradient buffer, update replicated on every rank; *-sharded: "
                         "reduce-scatter, update of the rank's chunk, all-gather of the parameters (gs_trainer_set_sharded_update); "
                         "*-compact: all-gather of the per-camera dL_dRGB records + all-reduce of the twelve non-SH planes, SH planes rebuilt "
                         "on every rank (gs_trainer_set_compact_exchange); auto (default): torch-compact where it moves fewer bytes "
                         "than the all-reduce (cameras < ~2 M and >= ranks), else torch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=16, help="views of the workload the CPU baseline leg times")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and the collective hook even with one rank (plumbing test)")
    ap.add_argument("--no-stage-events", action="store_true", help="diagnostic only: time the steps without the per-stage HIP events (no roofline object)")
    ap.add_argument("--sh-fp16", action="store_true", help="trainer option sh_fp16: the projection reads a half-precision copy of the SH planes (BASELINE config 5)")
    ap.add_argument("--long-steps", type=int, default=6000, help="steps of the untimed-by-the-metric long run reported as `long_run` (0: skip); the default keeps "
                    "the GPU busy for ~8 s at cfg3 (capped at ~8 s for every configuration): a run an external utilisation sampler with a period of seconds gets to see")
    ap.add_argument("--list-cut", type=int, default=1, help="diagnostic: 0 switches the depth cut of the tile lists off (trainer option list_cut; it only ever acts in dense scenes)")
    ap.add_argument("--fuse-update", type=int, default=1, help="diagnostic: 0 runs the update as a launch of its own (trainer option fuse_update) instead of inside the per-splat reduction")
    ap.add_argument("--views", type=int, default=0, help="diagnostic only: override the number of views per step (not the metric's config)")
    ap.add_argument("--exchange-overlap", type=int, default=0,
                    help="*-compact only: 1 runs the all-reduce of the twelve non-SH planes on the trainer's second stream and a communicator of its own, "
                         "beside the all-gather; 0 (default) issues the two collectives one after the other on one communicator — the form with "
                         "nothing concurrent in it, for a measurement that has never had an 8-GPU node to rehearse on")
    ap.add_argument("--dist-backend", default="cpu:gloo,cuda:nccl",
                    help="torch.distributed backend; the default carries device tensors over nccl (= RCCL).  'gloo' is for REHEARSALS of the multi-rank path "
                         "on a box with one GPU (RCCL refuses two ranks on one device): tests/test_gpu_dist2.py runs two ranks of this script that way")
    ap.add_argument("--device", type=int, default=-1, help="HIP device of this rank (default: LOCAL_RANK); rehearsals on one GPU pass 0")
    ap.add_argument("--no-verify-exchange", action="store_true",
                    help="skip the check of the data-parallel step against an unsharded step on rank 0 before the warm-up (N > 1)")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0,
                    help="run the same training steps untimed for about this long before the W warm-up steps, then put the model back to the random-init "
                         "splats (the pre-warm steps train it) and time the K steps: the timed region then sees the clocks a training run lives at.  An "
                         "MI355X that sat idle through the Python setup runs its first tens of steps 2-7 %% slower (box to box: 692 against 739 steps/s "
                         "for 20 against 2000 steps).  The window WITHOUT the pre-warm is measured first and reported beside it (`cold_start`, `prewarm` in the "
                         "JSON line); 0 switches the pre-warm off.  NOT a workload change: without the reset a pre-warm would measure a model that has "
                         "trained (num_rendered falls by several per cent over thousands of steps: 841-890 steps/s) — that mistake is on record in DESIGN.md 6")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)

    torch = dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        device = local_rank if args.device < 0 else args.device
        torch.cuda.set_device(device)
        if "nccl" in args.dist_backend:
            dist.init_process_group(backend=args.dist_backend, device_id=torch.device("cuda", device), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)

    import gsplat_amd as gs
    from gsplat_amd import capi
    L = capi.lib()
    if L.gs_device_count() < 1:
        raise SystemExit("bench.py needs a GPU: libgsplat_mi355 has no CPU fallback")
    P, M, V_total, W, H = gs.synth.CONFIGS[args.config]
    if args.views:
        V_total = args.views
    D = gs.synth.sh_degree_for(M)
    n_cams = max(V_total // 2, 1)
    V_total = 2 * n_cams
    seed = gs.synth.seed_for(args.config)
    s = gs.synth.random_splats(P, M, seed)
    cams = gs.camera.get_cameras(n_cams)

    # ---- truth images: the product's own preview render of a second splat set, quantised RGBA8 ----
    t0 = time.time()
    tsp = gs.synth.random_splats(max(P // 2, 1), M, seed + 1000)
    thost = gs.ModelSplatsHost.fromVectors(tsp["loc"], tsp["sh"], tsp["scale"], tsp["opac"], tsp["rot"])
    thost.shDegree = D
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(thost)
    mine = gs.dist.shard_views(V_total, rank, world)
    framesW, framesB = [None] * n_cams, [None] * n_cams
    verify = use_dist and world > 1 and not args.no_verify_exchange
    # rank 0 also renders the other ranks' truth images when the exchange is verified against an unsharded step (below)
    for v in (range(V_total) if (verify and rank == 0) else mine):
        cam = cams[v % n_cams]
        white = v < n_cams
        fb = tr.render(W, H, 1.0, cam, background=(1.0, 1.0, 1.0) if white else (0.0, 0.0, 0.0)).reshape(-1)
        (framesW if white else framesB)[v % n_cams] = fb
    blank = np.zeros(W * H, np.uint32)
    framesW = [f if f is not None else blank for f in framesW]   # passes owned by other ranks are never read here
    framesB = [f if f is not None else blank for f in framesB]

    # ---- the model under training ----
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = D
    tr.model = gs.ModelSplatsDevice(host)
    if args.sh_fp16:
        tr.set_option("sh_fp16", 1)
    if not args.fuse_update:
        tr.set_option("fuse_update", 0)
    if not args.list_cut:
        tr.set_option("list_cut", 0)
    tr.captureTruths(cams, framesW, framesB)
    tr.shard(rank, world)
    proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM if args.update == "adam" else capi.GS_UPDATE_SGD_CLAMP)
    hook = None
    collective = args.collective
    collective_note = None
    if use_dist:
        from gsplat_amd import dist as gsdist
        if collective == "auto":
            # The exchange form by the bytes it moves (gsdist.choose_exchange); its collectives by the library's OWN RCCL communicators
            # (gs_comm_*: ncclAllReduce / ncclAllGather enqueued from C on the trainer's streams — no Python inside the step) wherever
            # librccl loads on EVERY rank, else through torch.distributed (a ctypes -> Python callback per collective and step).
            form = gsdist.choose_exchange(n_cams, world, M)
            ident = (C.c_char * capi.GS_COMM_ID_BYTES)()
            native = L.gs_comm_unique_id(ident) == 0      # local probe: loads librccl, creates nothing
            why = None if native else capi.last_error()
            flag = torch.tensor([1 if native else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)    # CPU tensor: gloo under the mixed backend
            native = bool(int(flag[0])) and "nccl" in args.dist_backend     # (a gloo rehearsal on one GPU has no second device for RCCL)
            collective = ("rccl" if native else "torch") + ("-compact" if form == "compact" else "")
            collective_note = f"auto -> {collective}" + ("" if native else f" (native RCCL hooks not usable on every rank: {why or 'another rank, or a gloo rehearsal'})")
        if collective.endswith("compact") and n_cams < world:
            collective, collective_note = "torch", f"{collective} needs at least one camera per rank ({n_cams} cameras, {world} ranks): all-reduce instead"
        # with --exchange-overlap the all-reduce of the compact exchange runs beside its all-gather: a process group (communicator) of its own
        overlap = bool(args.exchange_overlap) and collective.endswith("compact")
        tr.set_option("exchange_overlap", 1 if overlap else 0)
        reduce_group = dist.new_group(backend="nccl" if "nccl" in args.dist_backend else args.dist_backend) if (collective == "torch-compact" and world > 1 and overlap) else None
        hook = {"torch": lambda: gsdist.TorchAllReduce(tr), "rccl": lambda: gsdist.NativeRcclComm(tr, rank, world),
                "torch-sharded": lambda: gsdist.TorchShardedUpdate(tr, rank, world),
                "rccl-sharded": lambda: gsdist.NativeRcclComm(tr, rank, world, sharded=True),
                "torch-compact": lambda: gsdist.TorchCompactExchange(tr, rank, world, cams, reduce_group),
                "rccl-compact": lambda: gsdist.NativeRcclComm(tr, rank, world, compact_cameras=cams)}[collective]()
    setup_s = time.time() - t0

    def sync_all():
        tr.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def replica_digests():
        import hashlib
        hm = gs.ModelSplatsHost.fromDevice(tr.model)
        digest = hashlib.sha256(b"".join(np.ascontiguousarray(a[:k * hm.count]).tobytes() for a, k in
                                         ((hm.locations, 3), (hm.shs, 3 * hm.shCoeffs), (hm.scales, 3), (hm.opacities, 1), (hm.rotations, 4)))).hexdigest()
        return gsdist.all_gather_bytes(digest.encode(), 64)   # CPU tensors: gloo under the mixed backend

    # ---- N > 1: the data-parallel step checked on THIS hardware against an unsharded step, before anything is timed ----
    # One gradients-only step (learning rates 0) through the installed exchange; rank 0 runs the same iteration unsharded on a second
    # trainer and compares the averaged-gradient planes: SH planes bit for bit under the compact exchange (no collective sums them),
    # everything to 2e-5 of the plane's scale (the collective re-associates the pass sums).  A mismatch falls back to the
    # all-reduce through torch.distributed, the form with the least machinery, and says so in the JSON line.
    exchange_check = None
    if verify:
        def grad_planes(trainer):
            ptr, n = trainer.grad_buffer()
            trainer.synchronize()
            buf = np.empty(n, np.float32)
            capi.check(L.gs_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4))
            return buf.reshape(12 + 3 * M, -1)[:, :P]

        def check_once():
            still = gs.Project(lrLocation=0.0, lrSh=0.0, lrScale=0.0, lrOpacity=0.0, lrRotation=0.0)
            # a collective that REFUSES its arguments does so on every rank alike (the hook reports it, the step returns
            # GS_ERR_COLLECTIVE): that is a verdict like any other, the ranks agree on it and move on to the fall-back
            failed = None
            try:
                tr.train(still, densify=False)
                tr.synchronize()
            except capi.GsError as e:
                failed = str(e)
            flags = torch.tensor([0 if failed is None else 1], dtype=torch.int32)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
            if int(flags[0]):
                return {"ok": False, "step_failed_on_some_rank": failed or "on another rank"}
            res = None
            if rank == 0:
                ref = gs.Trainer(W, H)
                if args.sh_fp16:
                    ref.set_option("sh_fp16", 1)
                ref.model = gs.ModelSplatsDevice(host)
                ref.captureTruths(cams, framesW, framesB)
                ref.train(still, densify=False)
                g_ref, g_dp = grad_planes(ref), grad_planes(tr)
                ref.close()
                scale = np.abs(g_ref).max(1) + 1e-30
                checked = "every gradient plane"
                if collective.endswith("sharded"):
                    # reduce-scatter leaves the iteration's sums in the rank's OWN chunk of the flat buffer only (the other chunks keep this
                    # rank's partial sums; the update reads nothing of them): compare rank 0's chunk [0, total / world), the library's
                    # one chunking of every plane-major buffer (gs_internal.h, shard_total_floats)
                    Pa = tr.grad_buffer()[1] // (12 + 3 * M)
                    n_flat, q = (12 + 3 * M) * Pa, world * 64
                    chunk = (n_flat + q - 1) // q * q // world
                    own = np.zeros((12 + 3 * M, Pa), bool)
                    own.reshape(-1)[:min(chunk, n_flat)] = True
                    own = own[:, :P]
                    g_dp = np.where(own, g_dp, g_ref)
                    checked = f"rank 0's chunk of the flat buffer ({int(own.sum())} of {own.size} entries): the reduce-scatter sums nothing else on this rank"
                dev = float((np.abs(g_dp - g_ref).max(1) / scale).max())
                sh_same = bool(np.array_equal(g_dp[3:3 + 3 * M].view(np.uint32), g_ref[3:3 + 3 * M].view(np.uint32)))
                res = {"max_plane_deviation": dev, "sh_planes_bit_identical_to_unsharded_step": sh_same, "compared": checked,
                       "ok": bool(dev <= 2e-5 and np.isfinite(g_dp).all() and np.abs(g_ref).max() > 0)}
            return json.loads(gsdist.broadcast_bytes(json.dumps(res).encode() if rank == 0 else b"", src=0).decode())
        exchange_check = dict(check_once(), collective=collective)
        if not exchange_check["ok"] and collective != "torch":
            collective_note = f"{collective} failed the check against the unsharded step ({exchange_check}): fell back to the torch all-reduce"
            capi.check(L.gs_trainer_set_compact_exchange(tr.handle, None, None, None, 0, 1, 0, None))
            capi.check(L.gs_trainer_set_sharded_update(tr.handle, None, None, None, 0, 1))
            collective = "torch"
            hook = gsdist.TorchAllReduce(tr)
            exchange_check = dict(check_once(), collective=collective, after_fallback=True)
    # ---- the cold-start window: W warm-up steps + K timed steps right after setup, as rounds 1-3 measured the metric ----
    cold_start = None
    if args.prewarm_seconds > 0 and args.steps > 0:
        for _ in range(args.warmup):
            tr.train(proj, densify=False)
        sync_all()
        t_c = time.perf_counter()
        for _ in range(args.steps):
            tr.train(proj, densify=False)
        sync_all()
        cold_s = time.perf_counter() - t_c
        if use_dist:
            tc = torch.tensor([cold_s], dtype=torch.float64)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            cold_s = float(tc[0])
        cold_start = {"value": args.steps / cold_s, "unit": "steps/s", "steps": args.steps, "ms_per_step": cold_s / args.steps * 1e3,
                      "note": "the same W warm-up + K timed steps run right after setup, BEFORE the pre-warm: a device that has been idle; what rounds 1-3 reported as the metric"}
    # ---- pre-warm (untimed, by wall time): bring the device to its sustained clocks ----
    prewarm = None
    if args.prewarm_seconds > 0:
        sync_all()
        t_p, n_p = time.perf_counter(), 0
        while True:
            for _ in range(20):
                tr.train(proj, densify=False)
            n_p += 20
            tr.synchronize()
            go_on = time.perf_counter() - t_p < args.prewarm_seconds
            if use_dist:   # every rank runs the same number of steps (a collective sits inside each): rank 0 decides
                flag = torch.tensor([1 if go_on else 0], dtype=torch.int32)
                dist.broadcast(flag, src=0)
                go_on = bool(flag[0])
            if not go_on:
                break
        # the pre-warm steps TRAINED the model (thousands of Adam steps: splats move and shrink, num_rendered drifts by several per
        # cent): the timed region measures the metric's own workload, random-init splats, so the model is put back (same bits on
        # every rank; this also resets the optimizer state) — the device stays warm across the few milliseconds this takes
        tr.synchronize()
        tr.model = gs.ModelSplatsDevice(host)
        prewarm = {"steps": n_p, "seconds": round(time.perf_counter() - t_p, 3),
                   "note": "untimed steps of the same workload run before the W warm-up steps (--prewarm-seconds) to reach sustained clocks; the model is reset "
                           "to the random-init splats afterwards, so the timed region runs the metric's own workload"}
    # ---- warm-up (untimed), then EXACTLY K timed steps ----
    st = None
    for _ in range(args.warmup):
        st = tr.train(proj, densify=False, stats=True)
    if use_dist and world > 1 and collective.endswith("compact") and args.warmup > 0:
        # The compact exchange has never run between real GPUs on the development box (one GPU): check what it must guarantee —
        # bit-identical replicas after the warm-up steps — before timing anything, and take the all-reduce form otherwise.
        sync_all()
        dg = replica_digests()
        if not all(x == dg[0] for x in dg):
            collective_note = f"{collective} left different replicas after the warm-up ({len(set(dg))} distinct): fell back to the all-reduce form"
            collective = "torch"
            capi.check(L.gs_trainer_set_compact_exchange(tr.handle, None, None, None, 0, 1, 0, None))
            tr.model = gs.ModelSplatsDevice(host)
            hook = gsdist.TorchAllReduce(tr)
            for _ in range(args.warmup):
                st = tr.train(proj, densify=False, stats=True)
    # Timed region: HIP events bracket ONLY the dominant kernel's stage (render_backward) — an event costs ~3 us of
    # stream time, which is not noise against a 0.55 ms step at 2 views/GPU.  The full stage table comes from an
    # extra, untimed pass below.
    DOM_STAGE = 5
    capi.check(L.gs_trainer_set_profiling(tr.handle, 0 if args.no_stage_events else (1 << (DOM_STAGE + 1))))
    sync_all()
    host_us = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        t_h = time.perf_counter()
        tr.train(proj, densify=False)
        host_us.append((time.perf_counter() - t_h) * 1e6)
    sync_all()
    elapsed = time.perf_counter() - t_start
    ms = (C.c_double * capi.GS_STAGE_COUNT)()
    launches = (C.c_longlong * capi.GS_STAGE_COUNT)()
    capi.check(L.gs_trainer_stage_times(tr.handle, ms, launches))
    dom_timed = (ms[DOM_STAGE], launches[DOM_STAGE])
    # not the metric: a LONG run of the same steps (no events, no statistics), so that the 20-50-step timed window is not the
    # only figure on record — clocks, caches and the run-ahead host queue have all settled by then
    capi.check(L.gs_trainer_set_profiling(tr.handle, 0))
    n_long = args.long_steps
    if n_long > 0 and args.steps > 0:   # keep the long run within ~8 s whatever the configuration (cfg5: 5 ms per step)
        n_long = max(min(n_long, int(8.0 / max(elapsed / args.steps, 1e-6))), min(n_long, 50))
    long_run = None
    if n_long > 0:
        # learning rates 0 for these steps: the same kernels and the same bytes (the update runs, the Adam moments move), but the model
        # stays the metric's workload — with the real learning rates thousands of steps TRAIN it (num_rendered falls by several per cent
        # and the steps get faster: 803 instead of 740 steps/s over 2000 steps), which is not what this figure is for
        proj_still = gs.Project(updateRule=proj.updateRule, lrLocation=0.0, lrSh=0.0, lrScale=0.0, lrOpacity=0.0, lrRotation=0.0)
        sync_all()
        t_l = time.perf_counter()
        for _ in range(n_long):
            tr.train(proj_still, densify=False)
        sync_all()
        long_s = time.perf_counter() - t_l
        if use_dist:
            tl = torch.tensor([long_s], dtype=torch.float64)
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            long_s = float(tl[0])
        long_run = {"value": n_long / long_s, "unit": "steps/s", "steps": n_long, "ms_per_step": long_s / n_long * 1e3,
                    "note": "the same step run this many times after the timed region with all learning rates 0 (same kernels and bytes, the model stays "
                            "the metric's workload), no HIP events; not the metric"}
    # untimed: every stage, a few steps
    capi.check(L.gs_trainer_set_profiling(tr.handle, 1))
    for _ in range(min(args.steps, 10)):
        tr.train(proj, densify=False)
    capi.check(L.gs_trainer_stage_times(tr.handle, ms, launches))
    capi.check(L.gs_trainer_set_profiling(tr.handle, 0))
    st = tr.train(proj, densify=False, stats=True) if st is None else st
    list_totals = (C.c_longlong * 4)()
    capi.check(L.gs_trainer_debug_list_totals(tr.handle, list_totals))
    list_totals = [int(x) for x in list_totals]
    # untimed, reported separately: the same steps in the per-pass form (option "fuse_camera_passes" off: one backward per
    # PASS, `var` produced on every step like the reference's accumulateGradients does), so that the cost of the dead
    # value the default step does not compute is on record
    per_pass = None
    if not use_dist:
        tr.set_option("fuse_camera_passes", 0)
        try:
            for _ in range(3):
                tr.train(proj, densify=False)
            tr.synchronize()
            n_pp = max(10, min(args.steps, 30))
            t_pp = time.perf_counter()
            for _ in range(n_pp):
                tr.train(proj, densify=False)
            tr.synchronize()
            per_pass = {"value": n_pp / (time.perf_counter() - t_pp), "unit": "steps/s", "steps": n_pp,
                        "note": "gs_trainer_set_option('fuse_camera_passes', 0): one backward per pass and `var` on every step (what a densify step runs); "
                                "measured after the timed region"}
        finally:
            tr.set_option("fuse_camera_passes", 1)
    # untimed, reported separately: one step WITH densify/prune, as the driver loop runs every 200th iteration
    densify_ms = None
    if not use_dist:
        tr.synchronize()
        t_d = time.perf_counter()
        st_d = tr.train(proj, densify=True, stats=True)
        densify_ms = (time.perf_counter() - t_d) * 1e3
    replicas_identical = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0])
        # every rank applied the same update to the same reduced gradients: the replicas must agree bit for bit
        digests = replica_digests()
        replicas_identical = all(x == digests[0] for x in digests)

    if rank == 0 and args.no_stage_events:
        print(json.dumps({"diagnostic": "no stage events", "value": args.steps / elapsed, "unit": "steps/s", "ms_per_step": elapsed / args.steps * 1e3,
                          "n_gpus": world, "views_per_step": V_total}))
    elif rank == 0:
        V_local = len(mine)
        R_mean = st.num_rendered / max(st.views, 1)
        N = W * H
        stages = {}
        for i in range(capi.GS_STAGE_COUNT):
            name = L.gs_stage_name(i).decode()
            if launches[i]:
                stages[name] = {"ms_per_launch": ms[i] / launches[i], "launches": int(launches[i])}
        # dominant kernel = the stage with the largest device time
        kern = {k: v for k, v in stages.items() if k not in ("collective",)}
        dom = max(kern, key=lambda k: kern[k]["ms_per_launch"] * kern[k]["launches"])
        dom_ms, dom_src = kern[dom]["ms_per_launch"], "HIP events, untimed stage pass after the timed region (another stage outweighed render_backward)"
        if dom == L.gs_stage_name(DOM_STAGE).decode() and dom_timed[1]:
            dom_ms, dom_src = dom_timed[0] / dom_timed[1], f"HIP events on the trainer's stream around every launch of the timed region ({int(dom_timed[1])} launches)"
        dom_bytes = stage_bytes(dom, P, M, N, R_mean, V_local)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # the bytes the form that actually RAN must move: the fused backward walks one list and writes one row set per CAMERA
        # (its two passes share them) and reads truth + colour per pass — fewer than SURVEY's per-view figure x views
        n_groups = max(V_local // 2, 1)
        form_bytes = {"render_backward": n_groups * (40 * R_mean + 44 * P) + V_local * 24 * N,
                      "render_forward": n_groups * 40 * R_mean + V_local * 12 * N + n_groups * 8 * N}.get(dom, dom_bytes)
        cut_active = tr.list_cut_stats()[0] > 0 and list_totals[1] > 0
        if cut_active:   # the lists the kernels walked are the cut ones
            R_listed = list_totals[1] / max(n_groups, 1)
            form_bytes = {"render_backward": n_groups * (40 * R_listed + 44 * P) + V_local * 24 * N,
                          "render_forward": n_groups * 40 * R_listed + V_local * 12 * N + n_groups * 8 * N}.get(dom, form_bytes)
        step_bytes = sum(stage_bytes(k, P, M, N, R_mean, V_local) for k in kern) + (48 + 12 * M) * P
        # (a dense step that cut its tile lists moved the entries it LISTED, not the R of SURVEY's model: the form's bytes are priced with those)
        cut_active = tr.list_cut_stats()[0] > 0 and list_totals[1] > 0
        R_form = list_totals[1] / max(n_groups, 1) if cut_active else R_mean
        step_form_bytes = sum(stage_form_bytes(k, P, M, N, R_form, V_local, n_groups) for k in kern)
        update_fused = "update" not in kern and "splat_backward" in kern
        if update_fused:     # no update launch: the per-splat reduction applied it (its bytes are part of that stage's time)
            step_bytes += stage_bytes("update", P, M, N, R_mean, V_local)
            step_form_bytes += stage_form_bytes("update", P, M, N, R_mean, V_local, n_groups)
        # what this box's HBM sustains for a streaming copy ("state both"): 1 GiB, float4 per lane, the fastest launch of sixteen forms x 3
        peak_measured = C.c_double(0.0)
        if L.gs_debug_hbm_copy_rate(1 << 30, 3, C.byref(peak_measured)) != 0:
            peak_measured = C.c_double(0.0)
        peak_measured = float(peak_measured.value) or None
        ms_per_step = elapsed / args.steps * 1e3
        # Counters of the dominant kernel (rocprofv3 --pmc passes of the same command, tools/pmc_pass.sh, committed as
        # profiles/pmc_latest.json).  They are only reported when they were collected from the kernel sources this run
        # uses (source stamp) and for the metric's own configuration; otherwise null.
        traffic = valu = None
        pmc_note = "profiles/pmc_latest.json missing"
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from source_stamp import source_stamp
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            stamp_ok = pmc.get("_stamp", {}).get("source_sha256") == source_stamp()
            k = pmc.get(dom)
            if not stamp_ok:
                pmc_note = "profiles/pmc_latest.json was collected from other kernel sources (stamp mismatch): counters not reported"
            elif not (k and args.config == 3 and not args.views and world == 1):
                pmc_note = "counters are collected for the metric's configuration on one GPU only"
            else:
                pmc_note = "rocprofv3 --pmc, separate passes, same sources (stamp matches), largest dispatch of the kernel"
                if "fetch_bytes_x2" in k and "write_bytes" in k:
                    traffic = k["fetch_bytes_x2"] + k["write_bytes"]
                if "SQ_INSTS_VALU" in k:
                    # VALU issue roofline: every wave64 VALU instruction holds its SIMD-32 for 2 cycles
                    insts = k["SQ_INSTS_VALU"]
                    per_simd_cycle = insts / N_SIMD / (dom_ms * 1e-3 * CLOCK_GHZ * 1e9)
                    valu = {"wave_instructions": insts, "per_simd_per_cycle": per_simd_cycle, "peak_per_simd_per_cycle": 0.5,
                            "frac": per_simd_cycle / 0.5, "clock_ghz_assumed": CLOCK_GHZ,
                            "note": "SQ_INSTS_VALU of the launch / 1024 SIMDs / (measured launch time x max clock) against one instruction per 2 cycles"}
        except Exception as e:  # noqa: BLE001
            pmc_note = f"profiles/pmc_latest.json unusable: {e!r}"
        hbm_frac = achieved / HBM_PEAK_GBS
        bound = "valu" if (valu and valu["frac"] > hbm_frac) else "hbm"
        out = {
            "metric": "train steps/sec (fwd+bwd+Adam), 100k splats x 16 views @1024^2, 1->8 GPU" if args.config == 3 else
                      f"train steps/sec (fwd+bwd+update), BASELINE config {args.config}",
            "value": args.steps / elapsed,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[args.config] + (f" [DIAGNOSTIC: views overridden to {V_total}]" if args.views else ""), "splats": P, "sh_coeffs": M, "views_per_step": V_total,
                       "views_per_gpu": V_local, "width": W, "height": H, "update": args.update,
                       "sh_storage": "fp16 read copy (trainer option sh_fp16; fp32 master and gradients)" if args.sh_fp16 else "fp32",
                       "parallelism": f"view-parallel x{world}" if world > 1 else "single GPU",
                       "collective": ({"torch": "torch all-reduce of %d fp32" % ((12 + 3 * M) * P), "rccl": "rccl all-reduce of %d fp32" % ((12 + 3 * M) * P),
                                       "torch-sharded": "torch reduce-scatter + all-gather of %d fp32" % ((12 + 3 * M) * P),
                                       "rccl-sharded": "rccl reduce-scatter + all-gather of %d fp32" % ((12 + 3 * M) * P),
                                       "torch-compact": "torch all-gather of %d dL_dRGB records of %d fp32 + all-reduce of %d fp32 (twelve non-SH planes), side by side" % (n_cams, 3 * P, 12 * P),
                                       "rccl-compact": "rccl all-gather of %d dL_dRGB records of %d fp32 + all-reduce of %d fp32 (twelve non-SH planes), side by side" % (n_cams, 3 * P, 12 * P)}[collective]
                                      if use_dist else "none"),
                       "collective_note": collective_note,
                       "exchange_checked_against_unsharded_step": exchange_check,
                       "wire_bytes_received_per_rank_per_step": ({f: gsdist.exchange_wire_bytes(f, n_cams, world, P, M) for f in ("allreduce", "sharded", "compact")}
                                                                  if use_dist else None),
                       "replicas_identical_after_run": replicas_identical,
                       "mean_num_rendered_per_view": R_mean, "max_tile_list": st.max_tile_list,
                       "camera_pass_sharing": "on (default): the white/black passes of a camera share projection, tile lists and "
                                              "the forward blend; a step without densify runs ONE backward per camera on the sum of the "
                                              "two residual images (the backward is linear in dL/dpixel; only accumulateGradients' `var`, "
                                              "read by the densify block alone, needs per-pass gradients and is produced on densify steps)"},
            "roofline": {"bound": bound, "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": hbm_frac, "traffic": traffic, "valu": valu, "counters": pmc_note,
                         "bound_note": "achieved/peak/frac are the HBM roofline of the dominant kernel (algorithmic bytes / measured time); "
                                       "`valu` is its vector-issue roofline; `bound` names the larger fraction (no dense contraction here: MFMA is not used)",
                         "algorithmic_bytes_per_launch": dom_bytes, "ms_per_launch": dom_ms, "measured": dom_src,
                         "form_bytes_per_launch": form_bytes, "frac_form": form_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "form_note": "frac prices SURVEY 8d's per-view bytes x the views the launch serves; frac_form prices the compulsory bytes of the form "
                                      "that ran (camera passes share lists and rows): the smaller, stricter figure",
                         "step_algorithmic_GB": step_bytes / 1e9,
                         "step_achieved_GBs": step_bytes / (ms_per_step * 1e-3) / 1e9,
                         "step_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "step_form_GB": step_form_bytes / 1e9,
                         "step_frac_form": step_form_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "peak_measured": peak_measured,
                         "peak_measured_note": "GB/s (read + written) of a 1 GiB float4 streaming copy on this device right after the run (gs_debug_hbm_copy_rate: the fastest launch of "
                                               "sixteen forms — grid size, 1 or 4 loads in flight per lane, plain or non-temporal accesses; form %d won): the achievable HBM "
                                               "ceiling (a 256 MiB copy reads ~6.8 TB/s: the Infinity Cache helps it); the *_measured fractions divide by it instead of the 8 TB/s "
                                               "specification" % L.gs_debug_hbm_copy_form(),
                         "frac_measured": None if not peak_measured else achieved / peak_measured,
                         "frac_form_measured": None if not peak_measured else form_bytes / (dom_ms * 1e-3) / 1e9 / peak_measured,
                         "step_frac_measured": None if not peak_measured else step_bytes / (ms_per_step * 1e-3) / 1e9 / peak_measured,
                         "step_frac_form_measured": None if not peak_measured else step_form_bytes / (ms_per_step * 1e-3) / 1e9 / peak_measured,
                         "cut_note": None if not cut_active else "this run cut its tile lists by depth (list_cut): achieved / frac / step_frac price SURVEY 8d's bytes for ALL "
                                     "%d entries per camera although only %d were listed — they are algorithmic rates, not bandwidths, and may exceed the peak; "
                                     "the *_form figures price the entries listed" % (int(R_mean), int(list_totals[1] / max(n_groups, 1))),
                         "step_note": "step_frac prices SURVEY 8d's per-VIEW bytes x the views of the step although projection, lists, forward blend and "
                                      "the backward ran once per CAMERA; step_frac_form prices the bytes of the form that ran, every stage: the stricter figure"},
            "stages_ms_per_launch": {k: round(v["ms_per_launch"], 4) for k, v in stages.items()},
            "update_fused_into_splat_backward": update_fused,
            "list_totals_last_step": dict(zip(("coarse_candidates", "tile_list_entries", "tiles_with_a_depth_bound", "tiles"), list_totals)),
            "list_cut": dict(zip(("attempts_with_cut_lists", "replayed_uncut"), tr.list_cut_stats()),
                             note="depth cut of the tile lists (dense scenes only: from 384 entries per tile on average; the whole run incl. warm-up and untimed legs)"),
            "host": {"enqueue_us_per_step_mean": float(np.mean(host_us)), "enqueue_us_per_step_min": float(np.min(host_us)), "enqueue_us_per_step_max": float(np.max(host_us)),
                     "gpu_us_per_step": ms_per_step * 1e3,
                     "note": "wall time of the gs_trainer_step call on rank 0 (Python wrapper + C enqueue + the one early wait on the arena-overflow flags; a call that "
                             "finds the stream's queue full also waits there), timed region; the host must stay below gpu_us_per_step for the device never to idle"},
            "metric_definition": "v2 (rounds 4+): EXACTLY K timed steps after W warm-up steps, preceded by a ~1 s untimed pre-warm of the same workload and a model reset "
                                 "(sustained clocks); v1 (rounds 1-3) timed the same K steps right after setup — that figure is `cold_start` here",
            "stages_note": "all-stage table: HIP events over %d extra steps run after the timed region (timing every stage costs ~3 us of stream time per event)" % min(args.steps, 10),
            "cold_start": cold_start,
            "prewarm": prewarm,
            "long_run": long_run,
            "per_pass_form": per_pass,
            "densify_step": None if densify_ms is None else {"ms": round(densify_ms, 3), "count_before": st_d.count_before, "count_after": st_d.count_after,
                             "note": "one extra step with densify/prune after the timed region (the driver loop does this every 200th iteration); includes the step itself"},
            "setup_seconds": round(setup_s, 2),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(gs, s, cams, framesW, framesB, P, M, D, W, H, n_cams, args.cpu_views)
        print(json.dumps(out), flush=True)
    tr.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(gs, s, cams, framesW, framesB, P, M, D, W, H, n_cams, n_views):
    """The oracle (CPU restatement of the reference step; the reference has no CPU path of its own, SURVEY
    D2) timed on this box's host cores on a bounded sample: `n_views` of the workload's passes, scaled
    to the full pass count, plus the update timed in full."""
    from oracle import pyoracle as orc
    views = gs.camera.train_views(cams, W, H)
    V_total = 2 * n_cams
    pick = [0, n_cams][:n_views] if n_views <= 2 else list(range(min(n_views, V_total)))
    truths = np.concatenate([(framesW[v] if v < n_cams else framesB[v - n_cams]) for v in pick])
    # repeated until about ten seconds of CPU work are on the clock (at most five times); the FASTEST repetition is the baseline
    t_all, reps, t_views = time.perf_counter(), 0, float("inf")
    while reps < 5 and (reps == 0 or time.perf_counter() - t_all < 10.0):
        t0 = time.perf_counter()
        o = orc.train_views(P, D, M, W, H, s["loc"], s["sh"], s["scale"], s["opac"], s["rot"], views[pick], truths, float(V_total))
        t_views = min(t_views, time.perf_counter() - t0)
        reps += 1
    p = {k: s[k].copy() for k in ("loc", "sh", "scale", "opac", "rot")}
    m = np.zeros((11 + 3 * M) * P, np.float32)
    v = np.zeros_like(m)
    t0 = time.perf_counter()
    orc.apply_adam(p["loc"], p["sh"], p["scale"], p["opac"], p["rot"], o, m, v, 1, (5e-5, 1e-4, 2e-5, 1e-4, 2.5e-5), 0.3, 0.9, 0.999, 1e-15, M)
    t_upd = time.perf_counter() - t0
    step_s = t_views / len(pick) * V_total + t_upd
    return {"value": 1.0 / step_s, "unit": "steps/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{len(pick)} of {V_total} passes of the same workload timed ({t_views:.2f} s, fastest of {reps} repetitions) and scaled x{V_total / len(pick):g}, "
                      f"Adam update timed in full ({t_upd:.3f} s); CPU restatement of the reference semantics (oracle/), OpenMP with one thread per CPU "
                      f"this process may use ({orc.usable_cpus()} by affinity and cgroup quota; the host shows {os.cpu_count()} hardware threads — with one OpenMP "
                      f"thread for each of those, which rounds 1-3 ran, the same sample takes about three times as long)"}


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:  # noqa: BLE001
        if isinstance(e, SystemExit):
            raise
        # A failed step on one rank (e.g. GS_ERR_COLLECTIVE) leaves the peers blocked in the collective: end THIS process at
        # once and non-zero, without the process-group teardown that would wait for them; the launcher then ends the peers.
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
