#!/usr/bin/env python3
"""The LOCAL step of rank 0 of an N-GPU run, measured on one GPU: the trainer is sharded as rank 0 of `world` (it owns the cameras
c % world == 0), the exchange is installed with collective hooks that return at once — so every kernel of the data-parallel path runs
with its real shapes (the per-camera pack, the SH rebuild over ALL cameras' records, the chunked update of the sharded form), and
nothing of the wire.  1.337 ms (the single-GPU step) / this figure is the ceiling of the strong-scaling curve at that N for that
exchange; the collectives' own time comes on top.  The gradients are wrong by construction (the other ranks'
parts of the buffers are never filled): learning rates are 0.
    gpurun -- 'python tools/local_step_at_world.py'"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsplat_amd as gs  # noqa: E402
from gsplat_amd import capi  # noqa: E402


def measure(world, form, steps, config, want_stages=False):
    P, M, V, W, H = gs.synth.CONFIGS[config]
    n_cams = V // 2
    s = gs.synth.random_splats(P, M, gs.synth.seed_for(config))
    cams = gs.camera.get_cameras(n_cams)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = s["D"]
    rng = np.random.default_rng(1)
    frames = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) for _ in range(n_cams)]
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, frames, frames)
    tr.shard(0, world)
    noop = capi.ALLREDUCE_FN(lambda buf, n, stream, user: 0)
    L = capi.lib()
    if world > 1 or form != "none":
        if form == "compact":
            campos = np.ascontiguousarray([c.location for c in cams], np.float32).reshape(-1, 3)
            capi.check(L.gs_trainer_set_compact_exchange(tr.handle, C.cast(noop, C.c_void_p), C.cast(noop, C.c_void_p), None, 0, world, n_cams,
                                                         campos.ctypes.data_as(C.c_void_p)))
        elif form == "sharded":
            capi.check(L.gs_trainer_set_sharded_update(tr.handle, C.cast(noop, C.c_void_p), C.cast(noop, C.c_void_p), None, 0, world))
        elif form == "allreduce":
            capi.check(L.gs_trainer_set_allreduce(tr.handle, C.cast(noop, C.c_void_p), None))
    proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM, lrLocation=0.0, lrSh=0.0, lrScale=0.0, lrOpacity=0.0, lrRotation=0.0)
    for _ in range(400):      # warm clocks
        tr.train(proj, densify=False)
    tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train(proj, densify=False)
    tr.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    stages = None
    if want_stages:      # every stage timed with HIP events over a few more steps (the events cost ~3 us of stream time each)
        capi.check(L.gs_trainer_set_profiling(tr.handle, 1))
        for _ in range(20):
            tr.train(proj, densify=False)
        tr.synchronize()
        sums, launches = (C.c_double * capi.GS_STAGE_COUNT)(), (C.c_longlong * capi.GS_STAGE_COUNT)()
        capi.check(L.gs_trainer_stage_times(tr.handle, sums, launches))
        L.gs_stage_name.restype = C.c_char_p
        stages = {L.gs_stage_name(i).decode(): [round(sums[i] / 20, 4), int(launches[i]) // 20] for i in range(capi.GS_STAGE_COUNT) if launches[i]}
    tr.close()
    return (ms, stages) if want_stages else ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--config", type=int, default=3)
    a = ap.parse_args()
    single = measure(1, "none", max(a.steps // 3, 100), a.config)
    out = {"config": a.config, "single_gpu_ms": round(single, 4), "local_step_ms": {}, "ceiling_speedup": {}}
    for world in (2, 4, 8):
        for form in ("allreduce", "sharded", "compact"):
            ms = measure(world, form, a.steps, a.config)
            out["local_step_ms"][f"{form}@{world}"] = round(ms, 4)
            out["ceiling_speedup"][f"{form}@{world}"] = round(single / ms, 2)
    out["stages_ms_and_launches_per_step@8"] = {form: measure(8, form, 200, a.config, want_stages=True)[1] for form in ("allreduce", "sharded", "compact")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
