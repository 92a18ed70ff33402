#!/usr/bin/env python3
"""Soak run of the auto-train loop (src/ui/UiFrame.cpp:266-298) with the reference's default project values: the canonical grid start,
truth images re-captured from a target splat set with re-rotated camera spheres, densify every `intervalDensify` iterations — for
thousands of iterations, under both update rules.  Watches what a benchmark never sees: the model growing towards its capacity through
dozens of densify steps, non-finite parameters, device memory creeping.   gpurun -- 'python tools/soak.py --iterations 6000'"""
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


def free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=6000)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--adam", type=int, default=1)
    ap.add_argument("--list-cut-min-avg", type=int, default=-1, help="trainer option list_cut_min_avg (0: the depth cut of the tile lists wherever a tile's "
                    "pixels finish; -1: the library's default, dense scenes only).  The cut changes no bit: two soak runs that differ only in this must end "
                    "with the same `model_sha256`")
    ap.add_argument("--list-cut", type=int, default=1)
    a = ap.parse_args()
    hip = C.CDLL("libamdhip64.so")
    W = H = a.size
    target = gs.synth.random_splats(3000, 4, 5)
    target["scale"] *= 3.0
    thost = gs.ModelSplatsHost.fromVectors(target["loc"] * 0.5, target["sh"], target["scale"], target["opac"], target["rot"])
    painter = gs.Trainer(W, H)
    painter.model = gs.ModelSplatsDevice(thost)

    def capture(cameras):
        fw = [painter.render(W, H, 1.0, c, background=(1.0, 1.0, 1.0)).reshape(-1) for c in cameras]
        fb = [painter.render(W, H, 1.0, c, background=(0.0, 0.0, 0.0)).reshape(-1) for c in cameras]
        return fw, fb
    tr = gs.Trainer(W, H)
    tr.set_option("list_cut", a.list_cut)
    if a.list_cut_min_avg >= 0:
        tr.set_option("list_cut_min_avg", a.list_cut_min_avg)
    tr.model = gs.ModelSplatsDevice(gs.fields.initFieldGrid())
    p = gs.Project.initProject()            # the reference's defaults (src/Project.h): 25 + 25 cameras, capture / densify intervals
    if a.adam:
        p.updateRule = capi.GS_UPDATE_ADAM
    drv = gs.driver.AutoTrainer(tr, p, capture)
    log, t0 = [], time.time()
    free0 = None
    for it in range(a.iterations):
        cap, den = drv.step()
        if it == 50:
            tr.synchronize(); free0 = free_bytes(hip)
        if den or it == a.iterations - 1:
            n = tr.model.count
            host = gs.ModelSplatsHost.fromDevice(tr.model)
            M = host.shCoeffs
            finite = all(np.isfinite(x).all() for x in (host.locations[:3 * n], host.shs[:3 * M * n], host.scales[:3 * n], host.opacities[:n], host.rotations[:4 * n]))
            log.append(dict(iteration=it, count=n, finite=bool(finite), free_MB=round(free_bytes(hip) / 2 ** 20), scale_max=float(host.scales[:3 * n].max()) if n else 0.0,
                            opacity_range=[float(host.opacities[:n].min()), float(host.opacities[:n].max())] if n else None))
            if not finite or n == 0:
                break
    tr.synchronize()
    st = tr.train(p, densify=False, stats=True)
    free1 = free_bytes(hip)
    import hashlib
    hm = gs.ModelSplatsHost.fromDevice(tr.model)
    digest = hashlib.sha256(b"".join(np.ascontiguousarray(x[:k * hm.count]).tobytes() for x, k in
                                     ((hm.locations, 3), (hm.shs, 3 * hm.shCoeffs), (hm.scales, 3), (hm.opacities, 1), (hm.rotations, 4)))).hexdigest()
    print(json.dumps(dict(model_sha256=digest, list_cut=dict(zip(("attempts_with_cut_lists", "replayed_uncut"), tr.list_cut_stats())), iterations=p.iterations, seconds=round(time.time() - t0, 1), cameras=len(tr.truthCameras), update="adam" if a.adam else "sgd",
                          final_count=tr.model.count, capacity=tr.model.capacity, final_loss=float(st.loss), longest_tile_list=int(st.max_tile_list),
                          memory_grown_since_iteration_50_MB=round((free0 - free1) / 2 ** 20, 1) if free0 else None,
                          peak_count=max(x["count"] for x in log), iteration_of_peak=max(log, key=lambda x: x["count"])["iteration"],
                          free_MB_at_thirds=[log[len(log) // 3]["free_MB"], log[2 * len(log) // 3]["free_MB"], log[-1]["free_MB"]],
                          densify_log_first=log[:3], densify_log_last=log[-3:], all_finite=all(x["finite"] for x in log))))


if __name__ == "__main__":
    main()
